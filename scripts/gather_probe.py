"""Is the attention kernel bound by the random 512-byte gathers or by its own instruction stream?
Same kernel, same byte count, three index patterns at the cfg-2 shape: random (as LSH selects),
ascending-random, and contiguous rows (no DRAM page misses)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import magicpig_amd as mp
B, H, Hkv, D, M, K, L, NL = 8, 32, 8, 128, 32960, 10, 170, 12
nz = int(sys.argv[1]) if len(sys.argv) > 1 else 640
dev = "cuda:0"
srv = mp.SparseAttentionServer(); srv.alloc(NL, H, Hkv, D, B, M)
n = 32700
for l in range(NL):
    for b in range(B):
        k = torch.randn((Hkv, n, D), device=dev).to(torch.bfloat16); v = torch.randn((Hkv, n, D), device=dev).to(torch.bfloat16)
        srv.fill(l, b, k, v, k.float().norm(dim=-1))
BH = B * H
q = torch.randn((BH, D), device=dev).to(torch.bfloat16); qn = q.float().norm(dim=-1)
nnz = torch.full((BH,), nz, dtype=torch.int32, device=dev)
out = torch.zeros((BH, D), dtype=torch.bfloat16, device=dev); mve = torch.zeros((2, BH), device=dev)
def pattern(kind):
    ind = torch.zeros((BH, M), dtype=torch.int32, device=dev)
    for h in range(BH):
        if kind == "random": idx = torch.randperm(n, device=dev)[:nz]
        elif kind == "ascending": idx = torch.randperm(n, device=dev)[:nz].sort().values
        else: idx = torch.arange(nz, device=dev) + (h % 4) * 8000 + 37
        ind[h, :nz] = idx.int()
    return ind
for kind in ("random", "ascending", "contiguous"):
    ind = pattern(kind)
    for l in range(NL): srv.attention_wrapper(l, K, L, out, mve, q, qn, ind, nnz)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for rep in range(5):
        for l in range(NL): srv.attention_wrapper(l, K, L, out, mve, q, qn, ind, nnz)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / (5 * NL)
    print(f"{kind:11s} nnz={nz}: {us:7.2f} us/launch, {BH * nz * 520 / us / 1e3:7.1f} GB/s algorithmic")
