"""Shared builders for the parity tests: regenerate the synthetic inputs of a golden case
exactly as tests/golden/make_golden.py did (same seeds, tests/synth.py generator)."""
from __future__ import annotations

import os

import numpy as np

import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_golden(name: str) -> dict:
    with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def case_inputs(seed, B, H, Hkv, n, D, K, L):
    """Identical to make_golden.case_inputs (kept in sync by test_oracle_golden)."""
    keys, kns, vals = [], [], []
    for b in range(B):
        k, kn = synth.centred_keys(seed + 10 * b, Hkv, n, D)
        keys.append(k)
        kns.append(kn)
        vals.append(synth.normal_bf16_bits(seed + 10 * b + 1, (Hkv, n, D)))
    W = synth.normal_bf16_bits(seed + 7, (D, K * L))
    q = synth.normal_f32(seed + 3, (B * H, D))
    G = H // Hkv
    tgt = synth.randint(seed + 4, 0, n, (B * H,))
    for h in range(B * H):
        b, g = h // H, (h % H) // G
        q[h] = 0.5 * q[h] + 3.0 * synth.bf16_bits_to_f32(keys[b][g, tgt[h]])
    qb = synth.f32_to_bf16_bits(q)
    return np.stack(keys), np.stack(kns), np.stack(vals), W, qb


def stable_sort_codes(codes: np.ndarray):
    """torch.sort(stable=True) along the token axis -> (sorted int16 codes, int32 ids)."""
    order = np.argsort(codes, axis=-1, kind="stable").astype(np.int32)
    return np.take_along_axis(codes, order, axis=-1), order


def split_ragged(flat: np.ndarray, nnz: np.ndarray):
    off = np.concatenate([[0], np.cumsum(nnz)])
    return [flat[off[i]:off[i + 1]] for i in range(len(nnz))]


def dense_mask_counts(kcodes: np.ndarray, qcodes: np.ndarray, G: int) -> np.ndarray:
    """Second, independent statement of the retrieve math (library/lsh/test.py:41-43,
    evaluations/RULER/pred/attnserver_dist.py:777-886): per head, per token, the number of
    tables whose key code equals the query code.  kcodes [BHkv, L, n], qcodes [BH, L]."""
    BH = qcodes.shape[0]
    out = np.zeros((BH, kcodes.shape[-1]), np.int32)
    for h in range(BH):
        out[h] = (kcodes[h // G] == qcodes[h][:, None].astype(kcodes.dtype)).sum(0)
    return out
