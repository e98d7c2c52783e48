"""Lengths of the (table, bucket, token range) pieces a decode step probes at a BASELINE config: how often a piece
is longer than a 31-id / 63-id direct slot.  usage: python scripts/piece_lengths.py [cfg1]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import magicpig_amd as mp
from bench import CONFIGS

name = sys.argv[1] if len(sys.argv) > 1 else "cfg1"
cfg = CONFIGS[name]
B, H, Hkv, D, M, K, Lt, P = (cfg[k] for k in ("B", "H", "Hkv", "D", "M", "K", "L", "P"))
dev = torch.device("cuda:0")
server = mp.LSHSparseAttnServer(1, H, Hkv, D, K=K, L=Lt, batch_size=B, max_length=M, dense_layers=(), device="cuda:0")
for b in range(B):
    gen = torch.Generator(device=dev).manual_seed(b)
    kc = torch.randn((P, Hkv, D), device=dev, generator=gen).to(torch.bfloat16)
    vc = torch.randn((P, Hkv, D), device=dev, generator=gen).to(torch.bfloat16)
    server.fill(0, b, kc, vc, P); server.build_table(0, b, P)
R = server.lsh_retriever.R
bounds, _ = server.lsh_retriever.get_tables(0)          # [groups, L, NB, R + 1]
bounds = bounds.cpu().numpy().astype(np.int64)
G = H // Hkv
tot = np.zeros(3, np.int64); wg_over = np.zeros(2, np.int64); wgs = 0
allp = np.diff(bounds, axis=-1)
print(f"{name}: R = {R}; ALL pieces: mean {allp.mean():.2f}, >31: {(allp > 31).mean():.2e}, >63: {(allp > 63).mean():.2e}; "
      f"bucket size mean {bounds[..., -1].mean() - bounds[..., 0].mean():.1f}, "
      f"p1 {np.percentile(bounds[..., -1] - bounds[..., 0], 1):.0f}, p99 {np.percentile(bounds[..., -1] - bounds[..., 0], 99):.0f}")
for it in range(8):
    q = torch.randn((B, H, 1, D), device=dev).to(torch.bfloat16)
    codes, _ = server.hasher.query(q.reshape(B * H, D))
    codes = codes.cpu().numpy()                          # [BH, L]
    for h in range(B * H):
        g = (h // H) * Hkv + (h % H) // G
        rec = bounds[g, np.arange(Lt), codes[h]]        # [L, R + 1]
        pl = np.diff(rec, axis=-1)                      # [L, R] probed piece lengths
        tot += [pl.size, (pl > 31).sum(), (pl > 63).sum()]
        wg_over += [(pl > 31).any(axis=0).sum(), (pl > 63).any(axis=0).sum()]
        wgs += R
print(f"probed pieces: mean over-31 fraction {tot[1] / tot[0]:.3e}, over-63 {tot[2] / tot[0]:.3e}; "
      f"workgroups with at least one piece over 31: {wg_over[0] / wgs:.3f}, over 63: {wg_over[1] / wgs:.3f}")
