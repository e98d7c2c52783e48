"""CPU-side tuning of the `clustered` workload (bench.py --data clustered, tests/synth.clustered_keys):
selected fraction, candidates and the (table, bucket, token-range) piece lengths a decode step probes, for one kv
head at a BASELINE shape.  numpy only -- no GPU, no library.

    python scripts/tune_clustered.py [n] [alpha] [a] [b] [clusters] [rank] [K] [L] [R]
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402


def main():
    av = sys.argv[1:]
    n = int(av[0]) if len(av) > 0 else 97932
    alpha = float(av[1]) if len(av) > 1 else 0.5
    a = float(av[2]) if len(av) > 2 else 1.0
    b = float(av[3]) if len(av) > 3 else 0.5
    NC = int(av[4]) if len(av) > 4 else 64
    rank = int(av[5]) if len(av) > 5 else 4
    K = int(av[6]) if len(av) > 6 else 10
    L = int(av[7]) if len(av) > 7 else 150
    R = int(av[8]) if len(av) > 8 else 8
    D, G = 128, 4
    kb, kn = synth.clustered_keys(5, 1, n, D, alpha=alpha, a=a, b=b, clusters=NC, rank=rank)
    k = synth.bf16_bits_to_f32(kb[0])                                 # [n, D] centred
    W = synth.bf16_bits_to_f32(synth.normal_bf16_bits(12, (D, K * L)))
    bits = np.empty((n, K * L), bool)
    for s in range(0, n, 8192):
        bits[s:s + 8192] = (k[s:s + 8192] @ W) > 0
    kc = (bits.reshape(n, L, K) * (1 << np.arange(K))).sum(-1).astype(np.int32)     # [n, L]
    NQ = 16
    q = synth.normal_f32(77, (NQ, D))
    tgt = synth.randint(78, 0, n, (NQ,))
    qh = 0.5 * q + 3.0 * k[tgt]
    qn = qh / np.linalg.norm(qh, axis=-1, keepdims=True)
    qc = (((qn @ W) > 0).reshape(NQ, L, K) * (1 << np.arange(K))).sum(-1).astype(np.int32)   # [NQ, L]
    range_len = (((n + 371 + R - 1) // R) + 31) & ~31            # M = n + 372 at cfg 1
    sel, cand, pieces = [], [], []
    for i in range(NQ):
        hit = kc == qc[i][None, :]                               # [n, L]
        cnt = hit.sum(1)
        sel.append((cnt >= 2).mean())
        cand.append(hit.sum())
        # piece lengths: per table, per range
        rr = (np.arange(n) // range_len)
        for r in range(R):
            pieces.append(hit[rr == r].sum(0))
    pieces = np.concatenate(pieces)
    cosv = (k @ qn.T) / np.maximum(np.linalg.norm(k, axis=-1, keepdims=True), 1e-9)
    print(f"n={n} alpha={alpha} a={a} b={b} clusters={NC} rank={rank} K={K} L={L} R={R}")
    print(f"selected fraction mean {np.mean(sel) * 100:.2f}%  min {np.min(sel) * 100:.2f}%  max {np.max(sel) * 100:.2f}%")
    print(f"candidates/head mean {np.mean(cand):.0f}  max {np.max(cand):.0f}")
    print(f"pieces: mean {pieces.mean():.1f} p50 {np.percentile(pieces, 50):.0f} p99 {np.percentile(pieces, 99):.0f} "
          f"max {pieces.max()}  share>30 {np.mean(pieces > 30) * 100:.2f}%  share>126 {np.mean(pieces > 126) * 100:.3f}%")
    bs = np.bincount(kc[:, 0], minlength=1 << K)
    print(f"bucket sizes table 0: mean {bs.mean():.0f} p1 {np.percentile(bs, 1):.0f} p50 {np.percentile(bs, 50):.0f} "
          f"p99 {np.percentile(bs, 99):.0f} max {bs.max()}  empty {np.mean(bs == 0) * 100:.1f}%")
    print(f"|cos(q,k)| p50 {np.percentile(np.abs(cosv), 50):.3f} p99 {np.percentile(np.abs(cosv), 99):.3f}  key norm mean "
          f"{np.linalg.norm(k, axis=-1).mean():.2f}")


if __name__ == "__main__":
    main()
