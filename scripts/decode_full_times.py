"""Per-layer time of the full sparse-layer attention (window + sampled part) at cfg 1, eager launches:
decode_full (append, window attention, sparse layer, merge_state) vs decode_full_fused (append + one
kernel)."""
import sys, torch
sys.path.insert(0, "/root/repo")
import magicpig_amd as mp
from bench import CONFIGS
cfg = CONFIGS["cfg1"]
B, H, Hkv, D, M, K, Lt, P = (cfg[k] for k in ("B", "H", "Hkv", "D", "M", "K", "L", "P"))
NL = 4
dev = torch.device("cuda:0")
server = mp.LSHSparseAttnServer(NL, H, Hkv, D, K=K, L=Lt, batch_size=B, max_length=M, dense_layers=(), device="cuda:0")
for li in range(NL):
    gen = torch.Generator(device=dev).manual_seed(li)
    kc = torch.randn((P, Hkv, D), device=dev, generator=gen).to(torch.bfloat16)
    vc = torch.randn((P, Hkv, D), device=dev, generator=gen).to(torch.bfloat16)
    server.fill(li, 0, kc, vc, P); server.build_table(li, 0, P)
server.collect_nnz = False
q = torch.randn((NL, B, H, 1, D), device=dev).to(torch.bfloat16)
k = torch.randn((NL, B, Hkv, 1, D), device=dev).to(torch.bfloat16)
v = torch.randn((NL, B, Hkv, 1, D), device=dev).to(torch.bfloat16)
def run(fn, reps=30):
    g = torch.cuda.CUDAGraph()
    server.kv_last_page_len.fill_(69)
    server.window_nnz.fill_(69)
    s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for li in range(NL): fn(q[li], k[li], v[li], li)
    torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for li in range(NL): fn(q[li], k[li], v[li], li)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * NL)
print(f"decode_full (4 launches + centring ops): {run(server.decode_full):.1f} us per layer")
print(f"decode_full_fused (append + 1 launch):    {run(server.decode_full_fused):.1f} us per layer")
