"""Independent cross-checks of the oracle restatement (CPU): the same functions stated a second time
from the paper's definition in a few lines of numpy / float64, on randomised small cases.  The golden
vectors (test_oracle_golden.py) pin the oracle to the reference's outputs; these pin it to the MATH."""
import numpy as np
import pytest

import cases
import oracle
import synth


@pytest.mark.parametrize("seed,K,L,H,Hkv,B,n", [(1, 4, 12, 4, 2, 2, 300), (2, 7, 33, 6, 3, 1, 500),
                                                (3, 10, 150, 8, 2, 1, 2000), (4, 1, 5, 2, 2, 1, 70)])
def test_retrieve_is_collision_count_at_least_two(seed, K, L, H, Hkv, B, n):
    """lsh.cc:243-288: a token is emitted iff its key code equals the query code in >= 2 of the L
    tables of the query head's kv group."""
    NB, M, G = 1 << K, n + 5, H // Hkv
    codes = synth.randint(seed, 0, NB, (B, Hkv, L, n)).astype(np.int16)
    q = synth.randint(seed + 1, 0, NB, (B * H, L)).astype(np.int32)
    q[0] = codes[0, 0, :, 7]                       # one head shares every code with token 7
    lsh = oracle.LSH()
    lsh.alloc(K, L, 1, H, Hkv, B, M)
    for b in range(B):
        sc, si = cases.stable_sort_codes(codes[b])
        lsh.fill(0, b, sc, si)
    results = np.zeros((B * H, M), np.int32)
    nnz = np.zeros((B * H,), np.int32)
    lsh.batch_retrieve(0, q, results, nnz)
    for h in range(B * H):
        b, g = h // H, (h % H) // G
        hits = (codes[b, g].astype(np.int32) == q[h][:, None]).sum(0)          # [n] collision counts
        want = np.nonzero(hits >= 2)[0]
        assert np.array_equal(np.sort(results[h, :nnz[h]]), want), h
    assert 7 in results[0, :nnz[0]]


@pytest.mark.parametrize("seed,K,L", [(11, 8, 40), (12, 10, 150), (13, 4, 6)])
def test_attention_is_importance_corrected_softmax(seed, K, L):
    """sparse_attention.cc:164-240: z_j = q.k_j / sqrt(D) - ln(w_j + 1e-4) with
    p = (1 - theta/pi)^K, w = 1 - (1-p)^L - L p (1-p)^(L-1); out = softmax(z) V; LSE in base 2.
    Stated in float64 from bf16 inputs; the oracle (exact exp, cancellation-free weight) must agree to
    f32 accuracy."""
    B, H, Hkv, n, D = 1, 4, 2, 400, 128
    M = n
    keys, kns, vals, W, qb = cases.case_inputs(seed, B, H, Hkv, n, D, K, L)
    qn = np.linalg.norm(synth.bf16_bits_to_f32(qb).astype(np.float64), axis=-1).astype(np.float32)
    nnz = np.array([0, 37, 128, 400], np.int32)
    ind = np.zeros((B * H, M), np.int32)
    for h, z in enumerate(nnz):
        ind[h, :z] = np.sort(np.argsort(synth.u64(seed + h, n), kind="stable")[:z]).astype(np.int32)
    srv = oracle.SparseAttentionServer(exp_mode=2, clamp_cos=1)
    srv.alloc(1, H, Hkv, D, B, M)
    srv.fill(0, 0, keys[0], vals[0], kns[0])
    out = np.zeros((B * H, D), np.uint16)
    mve = np.zeros((2, B * H), np.float32)
    srv.attention_wrapper(0, K, L, out, mve, qb, qn, ind, nnz)
    kf = synth.bf16_bits_to_f32(keys[0]).astype(np.float64)
    vf = synth.bf16_bits_to_f32(vals[0]).astype(np.float64)
    qf = synth.bf16_bits_to_f32(qb).astype(np.float64)
    G = H // Hkv
    for h, z in enumerate(nnz):
        if z == 0:
            assert not out[h].any() and mve[1, h] == -np.inf
            continue
        g = h // G
        ids = ind[h, :z]
        s = kf[g, ids] @ qf[h]
        cos = np.clip(s / (float(qn[h]) * kns[0][g, ids].astype(np.float64)), -1, 1)
        p = (1 - np.arccos(cos) / np.pi) ** K
        w = 1 - (1 - p) ** L - L * p * (1 - p) ** (L - 1)
        zz = s / np.sqrt(D) - np.log(w + 1e-4)
        m = zz.max()
        e = np.exp(zz - m)
        ref = (e / e.sum()) @ vf[g, ids]
        assert np.allclose(synth.bf16_bits_to_f32(out[h]), ref, rtol=2 ** -7, atol=2e-4), h
        assert abs(mve[1, h] - (m + np.log(e.sum())) / np.log(2)) < 1e-3, h


def test_merge_state_is_lse_weighted_average():
    rng = np.random.default_rng(5)
    R, D = 16, 64
    va = synth.f32_to_bf16_bits(rng.standard_normal((R, D)).astype(np.float32))
    vb = synth.f32_to_bf16_bits(rng.standard_normal((R, D)).astype(np.float32))
    sa = rng.standard_normal(R).astype(np.float32) * 4
    sb = rng.standard_normal(R).astype(np.float32) * 4
    sb[3] = -np.inf                                   # an empty part contributes nothing
    v, s = oracle.merge_state(va, sa, vb, sb)
    wa = np.exp2(sa.astype(np.float64))
    wb = np.exp2(sb.astype(np.float64))
    ref = (wa[:, None] * synth.bf16_bits_to_f32(va) + wb[:, None] * synth.bf16_bits_to_f32(vb)) / (wa + wb)[:, None]
    assert np.allclose(synth.bf16_bits_to_f32(v), ref, rtol=2 ** -7, atol=1e-6)
    assert np.allclose(s, np.log2(wa + wb), atol=1e-5)
    assert np.array_equal(v[3], va[3])
