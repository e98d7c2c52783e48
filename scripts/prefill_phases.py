"""Prefill side of the path at cfg 1 (one layer, one request): key SimHash (all kv heads) and the
table build, timed with device events; the counting-sort tables are compared with torch.sort + fill."""
import sys, torch
sys.path.insert(0, "/root/repo")
import magicpig_amd as mp

n, M, D, K, Lt, Hkv, H = 97932, 98304, 128, 10, 150, 8, 32
W = torch.randn((D, K * Lt), device="cuda").to(torch.bfloat16)
sh = mp.SimHash(W, K, Lt)
keys = torch.randn((Hkv, n, D), device="cuda").to(torch.bfloat16)
lsh = mp.LSH(); lsh.alloc(K, Lt, 1, H, Hkv, 1, M)
ref = mp.LSH(); ref.alloc(K, Lt, 1, H, Hkv, 1, M)

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

codes = sh.keys(keys)
print(f"key hash, {Hkv} kv heads: {timed(lambda: sh.keys(keys)):.3f} ms")
print(f"table build (counting sort): {timed(lambda: lsh.fastfill(0, 0, codes)):.3f} ms")
def by_sort():
    sv, si = codes.sort(dim=-1, stable=True)
    ref.fill(0, 0, sv.contiguous(), si.int().contiguous())
print(f"table build (torch.sort + fill): {timed(by_sort, 2):.3f} ms")
tb, bb = lsh.get_tables(0), ref.get_tables(0)
same = torch.equal(tb[0], bb[0]) and torch.equal(tb[1][..., :n], bb[1][..., :n])
print("tables identical:", same)
