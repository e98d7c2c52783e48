"""A/B: a batch decoded as `chains` independent half-batch launch chains on separate streams (one hipGraph), with an
optional start skew of the second chain, against one chain.  Meant for an A/B library whose decode workgroups are
512 threads (two co-resident per CU): MP_LIB=magicpig_amd/lib/libmagicpig_hip_nt512.so.
usage: python scripts/two_chain_probe.py [cfg2] [chains] [skew_us] [layers] [B] [cluster]
cluster: workgroups per head of every chain (decode_cluster option; 1 = one 1024-thread workgroup per head, so that two
half-batch chains of 128 heads run on disjoint halves of the CUs)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import magicpig_amd._lib as L
if os.environ.get("MP_LIB"):
    L.LIB_PATH = os.path.abspath(os.environ["MP_LIB"])
import magicpig_amd as mp
from bench import CONFIGS

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
chains = int(sys.argv[2]) if len(sys.argv) > 2 else 2
skew_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
NL = int(sys.argv[4]) if len(sys.argv) > 4 else 6
cfg = CONFIGS[name]
B, H, Hkv, D, M, K, Lt, P = (cfg[k] for k in ("B", "H", "Hkv", "D", "M", "K", "L", "P"))
if len(sys.argv) > 5:
    B = int(sys.argv[5])                     # batch override (e.g. one half-batch chain alone)
if len(sys.argv) > 6:
    L.set_option("decode_cluster", int(sys.argv[6]))
assert B % chains == 0
Bc = B // chains
dev = torch.device("cuda:0")
servers = []
for c in range(chains):
    sv = mp.LSHSparseAttnServer(NL, H, Hkv, D, K=K, L=Lt, batch_size=Bc, max_length=M, dense_layers=(), device="cuda:0")
    for li in range(NL):
        for b in range(Bc):
            gen = torch.Generator(device=dev).manual_seed(1000 * c + 100 * li + b)
            kc = torch.randn((P, Hkv, D), device=dev, generator=gen).to(torch.bfloat16)
            vc = torch.randn((P, Hkv, D), device=dev, generator=gen).to(torch.bfloat16)
            sv.fill(li, b, kc, vc, P); sv.build_table(li, b, P)
    sv.collect_nnz = False
    servers.append(sv)
qs = [torch.randn((NL, Bc, H, 1, D), device=dev).to(torch.bfloat16) for _ in range(chains)]
print("R per chain:", servers[0].lsh_retriever.R, "lib", L.LIB_PATH)

def step():
    cur = torch.cuda.current_stream()
    if chains == 1:
        for li in range(NL):
            servers[0].decode(qs[0][li], li)
        return
    for c, s in enumerate(streams):
        s.wait_stream(cur)
        with torch.cuda.stream(s):
            if c > 0 and skew_us > 0:
                torch.cuda._sleep(int(skew_us * c * 2100))
            for li in range(NL):
                servers[c].decode(qs[c][li], li)
    for s in streams:
        cur.wait_stream(s)

streams = [torch.cuda.Stream() for _ in range(chains)]
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    step()
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    step()
for _ in range(5):
    graph.replay()
torch.cuda.synchronize()
reps = 40
t0 = time.perf_counter()
for _ in range(reps):
    graph.replay()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
for sv in servers:
    sv.attn_server.check()
print(f"{name}: chains {chains}, skew {skew_us} us, {NL} layers: {dt * 1e6 / NL:.2f} us per layer (whole batch), "
      f"{B / (dt / NL * 30):.0f} tokens/s at 30 layers")
