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


DATA_KINDS = ("randn", "clustered", "skewed")     # fixture key `data` holds the index; absent = randn


def golden_data(g: dict) -> str:
    return DATA_KINDS[int(g["data"])] if "data" in g else "randn"


def case_keys(seed, Hkv, n, D, data="randn"):
    """Centred keys + norms of one request: isotropic (synth.centred_keys) or the clustered / skewed workload
    of SURVEY.md 8(d) (synth.clustered_keys with the CLUSTERED / SKEWED parameters)."""
    if data == "randn":
        return synth.centred_keys(seed, Hkv, n, D)
    return synth.clustered_keys(seed, Hkv, n, D, **(synth.CLUSTERED if data == "clustered" else synth.SKEWED))


def case_inputs(seed, B, H, Hkv, n, D, K, L, data="randn"):
    """Synthetic single-layer inputs shared by tests/golden/make_golden.py and the tests (same seeds, tests/synth.py
    generator): keys, norms, values, hyperplanes, heavy-hitter queries."""
    keys, kns, vals = [], [], []
    for b in range(B):
        k, kn = case_keys(seed + 10 * b, Hkv, n, D, data)
        keys.append(k)
        kns.append(kn)
        vals.append(synth.normal_bf16_bits(seed + 10 * b + 1, (Hkv, n, D)))
    W = synth.normal_bf16_bits(seed + 7, (D, K * L))
    q = synth.normal_f32(seed + 3, (B * H, D))
    # heavy hitters: pull each query toward one key of its kv group so a few
    # tokens have cos ~ 0.9 and the importance weights span several orders of magnitude
    G = H // Hkv
    tgt = synth.randint(seed + 4, 0, n, (B * H,))
    for h in range(B * H):
        b, g = h // H, (h % H) // G
        q[h] = 0.5 * q[h] + 3.0 * synth.bf16_bits_to_f32(keys[b][g, tgt[h]])
    qb = synth.f32_to_bf16_bits(q)
    return np.stack(keys), np.stack(kns), np.stack(vals), W, qb


def check_sign_ties(g: dict, kcodes: np.ndarray, K: int) -> None:
    """The key codes of a fixture are the sign bits of the EXACT dot products; `kcodes_ties` lists the
    (b, g, l, t, bit, torch_bit) where torch's bf16 GEMM -- f32 accumulation in the library's order --
    landed on the other side of zero (make_golden.exact_sign_ties): a rounding-level handful out of
    10^8, exact value zero or ~1e-7.  Checks the list is that small and that `kcodes` holds the exact sign
    there, i.e. the complement of what torch's summation order happened to give."""
    ties = g.get("kcodes_ties")
    if ties is None or len(ties) == 0:
        return
    assert len(ties) <= 1e-7 * kcodes.size * K + 1
    assert np.abs(g["kcodes_tie_dots"]).max() < 1e-5
    for (b, gg, l, t, bit, torch_bit), dot in zip(ties, g["kcodes_tie_dots"]):
        assert (int(kcodes[b, gg, l, t]) >> bit) & 1 == int(dot > 0) == 1 - torch_bit


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


# ---- dense fixture (tests/golden/full_dense.npz): (tag, B, H, Hkv, n, M, nnz values).  D = 128 only: the
# reference's wv_kernel_dim128_full hard-codes eight 16-lane chunks (sparse_attention.cc:400-402,
# 441-449) and group sizes 1, 4, 8 (:395-397) -- the grid of library/sparse_attention/test_dense.py:8-14.
FULL_DENSE_CASES = [
    ("g1", 1, 4, 4, 200, 264, [0, 1, 16, 63, 64, 65, 200]),
    ("g4", 2, 8, 2, 300, 320, [1, 48, 63, 64, 65, 257, 300]),
    ("g8", 1, 8, 1, 1024, 1153, [15, 64, 1000, 1024]),
]


def full_dense_seed(seed, tag, H):
    return seed + len(tag) * 100 + H


def full_dense_inputs(seed, B, H, Hkv, n, D):
    keys = np.stack([synth.normal_bf16_bits(seed + 10 * b, (Hkv, n, D)) for b in range(B)])
    vals = np.stack([synth.normal_bf16_bits(seed + 10 * b + 1, (Hkv, n, D)) for b in range(B)])
    q = synth.normal_f32(seed + 3, (B * H, D))
    return keys, vals, q


# ---- window + LSE merge fixture (tests/golden/window_merge.npz, SURVEY f-2)
WINDOW_MERGE = dict(seed=61, B=2, H=8, Hkv=2, D=128, K=8, L=60, n=1500, M=1600, win_rows=(75, 70), win_M=96)


def window_merge_inputs(c):
    """Offloaded part as case_inputs; a static window of win_rows[b] rows per request (sink + local +
    generated tokens, already centred), the last row of which is this step's own (k, v)."""
    seed, B, H, Hkv, D, K, L, n = (c[k] for k in ("seed", "B", "H", "Hkv", "D", "K", "L", "n"))
    keys, kns, vals, W, qb = case_inputs(seed, B, H, Hkv, n, D, K, L)
    wk = [synth.normal_bf16_bits(seed + 100 + b, (Hkv, c["win_rows"][b], D)) for b in range(B)]
    wv = [synth.normal_bf16_bits(seed + 200 + b, (Hkv, c["win_rows"][b], D)) for b in range(B)]
    return keys, kns, vals, W, qb, wk, wv


# ---- prefill fixture (tests/golden/fill_centre.npz, SURVEY f-1): key / value caches of one request
FILL_CENTRE = dict(seed=71, seq_len=3000, Hkv=4, D=128, num_sink=4, num_local=64)


def fill_centre_inputs(c):
    """Token-major bf16 caches [seq_len, Hkv, D] whose keys have a clear per-column mean (as RoPE'd keys do)."""
    T, Hkv, D = c["seq_len"], c["Hkv"], c["D"]
    mean = synth.normal_f32(c["seed"] + 2, (1, Hkv, D)) * np.float32(0.75)
    k = synth.f32_to_bf16_bits((synth.normal_f32(c["seed"], (T, Hkv, D)) + mean).astype(np.float32))
    v = synth.normal_bf16_bits(c["seed"] + 1, (T, Hkv, D))
    return k, v


# ---- model plumbing around the path (tests/golden/llama_ops.npz, SURVEY f-3): RoPE tables + apply_rotary_pos_emb
# (models/llama.py:114-126, models/utils.py:29-45) and RMSNorm (models/utils.py:47-56 -> flashinfer.rmsnorm)
LLAMA_OPS = dict(seed=81, B=2, H=4, Hkv=2, D=128, hidden=512, max_len=4096, theta=500000.0, eps=1e-5,
                 positions=(0, 1, 777, 4095))


def llama_ops_inputs(c):
    """bf16 bit patterns: hidden states [B, 1, hidden], norm weight [hidden] (around 1), q [B, H, 1, D], k [B, Hkv, 1, D]."""
    x = synth.normal_bf16_bits(c["seed"], (c["B"], 1, c["hidden"]))
    w = synth.f32_to_bf16_bits((1.0 + 0.25 * synth.normal_f32(c["seed"] + 1, (c["hidden"],))).astype(np.float32))
    q = synth.normal_bf16_bits(c["seed"] + 2, (c["B"], c["H"], 1, c["D"]))
    k = synth.normal_bf16_bits(c["seed"] + 3, (c["B"], c["Hkv"], 1, c["D"]))
    return x, w, q, k
