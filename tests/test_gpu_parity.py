"""Parity of the HIP path (through the C ABI, via the host-side mirror classes) against the
CPU oracle and the reference's golden vectors.  Needs a real MI355X: `pytest -m gpu`.

Bars (BASELINE.json north_star): hash codes and selected token sets bit-exact; attention
outputs within the tolerance written at each assert (the reference's own test tolerance is
rtol = atol = 1e-2, library/sparse_attention/test_sparse.py:87-92; vs the oracle with exact exp
and the cancellation-free importance weight we hold the HIP path to <= 1 bf16 ulp on outputs,
1e-3 relative on probabilities, 1e-3 on the base-2 LSE)."""
import hashlib
import os

import numpy as np
import pytest
import torch

import cases
import oracle
import synth

pytestmark = pytest.mark.gpu

QHASH = ["qhash_r1_k10_l150", "qhash_r32_k10_l150", "qhash_r64_k11_l300", "qhash_r40_k8_l50",
         "qhash_r256_k10_l170"]
PIPE = ["lsh_small", "cfg0", "gqa_32h", "b2_k8_l60",
        "cfg2_small", "cfg3_small",   # BASELINE cfg 2 / cfg 3 head counts: B = 8 x H = 32 = 256 query heads, L = 170 / 150
        "cfg4_small", "g8_hkv2",      # cfg 4's per-GPU geometry (H = 8, Hkv = 1, K11 L300) and G = 8 with Hkv > 1
        "skew_small", "clustered_k10"]   # the non-isotropic workloads of SURVEY.md 8(d) (tests/synth.py)


@pytest.fixture(scope="module")
def mp():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import magicpig_amd

    return magicpig_amd


def bf16_t(bits, device="cpu"):
    return synth.to_torch_bf16(np.ascontiguousarray(bits)).to(device)


def bits_of(t):
    return t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)


# ------------------------------------------------------------------ a-1 SimHash

@pytest.mark.parametrize("name", QHASH)
@pytest.mark.parametrize("where", ["cuda", "cpu"])
def test_query_simhash_bit_exact_vs_reference(mp, name, where):
    g = cases.load_golden(name)
    seed, R, D, K, L = (int(x) for x in g["meta"])
    qb = synth.normal_bf16_bits(seed, (R, D))
    W = synth.normal_bf16_bits(seed + 7, (D, K * L))
    sh = mp.SimHash(bf16_t(W, "cuda"), K, L)
    codes, qn = sh.query(bf16_t(qb, where))
    assert codes.device.type == where
    assert np.array_equal(codes.cpu().numpy(), g["qcodes"])            # bit-exact vs torch reference
    oc, oqn = oracle.simhash_query(qb, W, K, L)
    assert np.array_equal(codes.cpu().numpy(), oc)                      # and vs the oracle
    assert np.allclose(qn.cpu().numpy(), oqn, rtol=1e-6)


def test_query_simhash_guard_band_covers_mfma_error(mp):
    """The exact-sign guard (|acc| <= 2^-16 ||x|| ||w||, simhash.hip SH_EPS) must sit far above the
    real error of the MFMA f32 accumulation: measure |acc - exact| / (||x|| ||w||) on 32 x 1500 x 8
    dot products and require an 8x margin."""
    D, K, L = 128, 10, 150
    worst = 0.0
    for seed in range(8):
        qb = synth.normal_bf16_bits(900 + seed, (32, D))
        W = synth.normal_bf16_bits(950 + seed, (D, K * L))
        sh = mp.SimHash(bf16_t(W, "cuda"), K, L)
        acc = torch.zeros((32, K * L), dtype=torch.float32, device="cuda")
        sh._debug_acc(acc)
        codes, _ = sh.query(bf16_t(qb, "cuda"))
        sh._debug_acc(None)
        torch.cuda.synchronize()
        # exact: normalised query (oracle definition) x planes in float64
        qf = synth.bf16_bits_to_f32(qb).astype(np.float64)
        nrm = np.sqrt((qf * qf).sum(-1)).astype(np.float32)
        nb = synth.bf16_bits_to_f32(synth.f32_to_bf16_bits(nrm)).astype(np.float32)
        nq = synth.bf16_bits_to_f32(synth.f32_to_bf16_bits(
            (synth.bf16_bits_to_f32(qb) / nb[:, None]).astype(np.float32))).astype(np.float64)
        Wf = synth.bf16_bits_to_f32(W).astype(np.float64)
        exact = nq @ Wf
        bound = np.linalg.norm(nq, axis=1)[:, None] * np.linalg.norm(Wf, axis=0)[None, :]
        err = np.abs(acc.cpu().numpy().astype(np.float64) - exact) / bound
        worst = max(worst, float(err.max()))
        bits = (exact > 0).reshape(32, L, K)
        ref_codes = (bits * (1 << np.arange(K))).sum(-1)
        assert np.array_equal(codes.cpu().numpy(), ref_codes)
    print(f"worst MFMA accumulation error / (||x|| ||w||) = {worst:.3e} = 2^{np.log2(worst):.1f}")
    assert worst < 2.0 ** -16 / 8, worst       # >= 8x margin under the guard band (measured: 2^-24.1)


@pytest.mark.parametrize("D", [128, 64])
def test_fast_query_normalisation_never_changes_a_code(mp, D):
    """The fused query hash normalises the row by a fast f32 form (f32 sum, v_sqrt_f32, reciprocal multiply) and takes
    the exact sequence (f64 sum and sqrt, IEEE divisions: the definition, pinned by the qhash_* fixtures) only where a
    value sits within a few ulps of a bf16 rounding boundary.  Over 64 K random rows at five scales -- ~0.1 % of them on
    the norm's fallback, ~3 % on the quotients' -- the codes equal those of the exact sequence forced for every row
    (`simhash_exact_norm`), and 4 K of them the oracle's; ||q|| agrees to 5e-7."""
    import magicpig_amd._lib as L_

    K, L = 10, 150
    gen = torch.Generator(device="cuda").manual_seed(900 + D)
    W = torch.randn((D, K * L), device="cuda", generator=gen).to(torch.bfloat16)
    sh = mp.SimHash(W, K, L)
    rows = 64 * 200
    try:
        for si, scale in enumerate((1.0, 1e-3, 37.0, 2.0 ** -20, 2.0 ** 12)):
            q = (torch.randn((rows, D), device="cuda", generator=gen) * scale).to(torch.bfloat16)
            fast_c, fast_n, ex_c, ex_n = [], [], [], []
            for r0 in range(0, rows, 64):                      # <= 64 rows per call: the fused (vector-pipe) hash
                L_.set_option("simhash_exact_norm", 0)
                c, n = sh.query(q[r0:r0 + 64])
                fast_c.append(c.clone()); fast_n.append(n.clone())
                L_.set_option("simhash_exact_norm", 1)
                c, n = sh.query(q[r0:r0 + 64])
                ex_c.append(c.clone()); ex_n.append(n.clone())
            fc, ec = torch.cat(fast_c), torch.cat(ex_c)
            assert torch.equal(fc, ec), (scale, int((fc != ec).any(-1).sum()))
            fn, en = torch.cat(fast_n), torch.cat(ex_n)
            assert torch.allclose(fn, en, rtol=5e-7, atol=0.0)
            if si < 2:
                qb = q[:2048].cpu().view(torch.int16).numpy().view(np.uint16)
                Wb = W.cpu().view(torch.int16).numpy().view(np.uint16)
                oc, oqn = oracle.simhash_query(qb, Wb, K, L)
                assert np.array_equal(fc[:2048].cpu().numpy(), oc)
                assert np.allclose(fn[:2048].cpu().numpy(), oqn, rtol=5e-7)
    finally:
        L_.set_option("simhash_exact_norm", 0)


def test_simhash_exact_sign_when_every_dot_product_is_tiny(mp):
    """Forces the rare branch (HIP guide rule 26): every hyperplane is built exactly or almost exactly
    orthogonal to one of the (normalised) queries, so a quarter of all dot products sit inside the
    guard band -- true ties and +-1e-7 -- and their bits come from the exact f64 recomputation -- in the MFMA kernel (codes compared directly) and in the hash fused
    into the retrieve kernel (nnz / selected ids compared through a decode)."""
    D, K, L, H, Hkv, n, M = 128, 10, 24, 4, 2, 1500, 1536
    qb = synth.normal_bf16_bits(321, (H, D))
    qf = synth.bf16_bits_to_f32(qb).astype(np.float64)
    nrm = np.sqrt((qf * qf).sum(-1)).astype(np.float32)
    nb = synth.bf16_bits_to_f32(synth.f32_to_bf16_bits(nrm))
    nq = synth.bf16_bits_to_f32(synth.f32_to_bf16_bits((synth.bf16_bits_to_f32(qb) / nb[:, None]).astype(np.float32))).astype(np.float64)
    # plane c is built against query r = c % H from three coordinates: (w_i, w_j) = (nq_j, -nq_i)
    # cancel EXACTLY (bf16 values, exact products), w_k = +-2^-20 leaves a dot product of ~1e-7
    # whose sign only exact arithmetic gets right; every third plane keeps w_k = 0 (a true tie -> bit 0)
    Wf = np.zeros((D, K * L), np.float64)
    for c in range(K * L):
        a = nq[c % H]
        i, j, k = (3 * c) % D, (3 * c + 1) % D, (3 * c + 2) % D
        Wf[i, c], Wf[j, c] = a[j], -a[i]
        Wf[k, c] = 0.0 if c % 3 == 0 else (2.0 ** -20) * (1 if (c // 3) % 2 else -1)
    W = synth.f32_to_bf16_bits(Wf.astype(np.float32))
    assert np.array_equal(synth.bf16_bits_to_f32(W).astype(np.float64), Wf)      # all exactly bf16
    exact = nq @ Wf                                                   # [H, K*L]
    inside = np.abs(exact) <= 2.0 ** -16 * np.linalg.norm(nq, axis=1)[:, None] * np.linalg.norm(Wf, axis=0)[None, :]
    assert inside.mean() > 0.2                                        # the branch really is exercised
    ref_codes = ((exact > 0).reshape(H, L, K) * (1 << np.arange(K))).sum(-1).astype(np.int32)
    sh = mp.SimHash(bf16_t(W, "cuda"), K, L)
    codes, _ = sh.query(bf16_t(qb, "cuda"))
    assert np.array_equal(codes.cpu().numpy(), ref_codes)
    oc, _ = oracle.simhash_query(qb, W, K, L)
    assert np.array_equal(oc, ref_codes)
    # fused path: same planes inside a decode; compare the retrieve result with the oracle's
    keys, kns = synth.centred_keys(323, Hkv, n, D)
    vals = synth.normal_bf16_bits(324, (Hkv, n, D))
    server = mp.LSHSparseAttnServer(1, H, Hkv, D, K=K, L=L, batch_size=1, num_sink_tokens=0,
                                    num_local_tokens=0, max_length=M, dense_layers=(), hash_func=bf16_t(W, "cuda"))
    server.hash_code_buffer = server.hasher.keys(bf16_t(keys, "cuda"))
    kcodes = server.hash_code_buffer.cpu().numpy()
    server.build_table(0, 0, n)
    server.attn_server.fill(0, 0, bf16_t(keys, "cuda"), bf16_t(vals, "cuda"), torch.from_numpy(kns).cuda())
    server.decode(bf16_t(qb, "cuda").view(1, H, 1, D), 0)
    torch.cuda.synchronize()
    cnt = cases.dense_mask_counts(kcodes, ref_codes, H // Hkv)
    assert np.array_equal(server.nnz.cpu().numpy(), (cnt > 1).sum(-1))


def test_key_simhash_bit_exact(mp):
    Hkv, n, D, K, L = 2, 777, 128, 10, 30
    keys, _ = synth.centred_keys(61, Hkv, n, D)
    W = synth.normal_bf16_bits(62, (D, K * L))
    sh = mp.SimHash(bf16_t(W, "cuda"), K, L)
    codes = sh.keys(bf16_t(keys, "cuda"))
    assert codes.dtype == torch.int16 and tuple(codes.shape) == (Hkv, L, n)
    assert np.array_equal(codes.cpu().numpy(), oracle.simhash_keys(keys, W, K, L))
    codes_h = sh.keys(bf16_t(keys, "cpu"))                  # host-staged path
    assert np.array_equal(codes_h.numpy(), codes.cpu().numpy())


# ------------------------------------------------------------------ tables + retrieve + attention

def _gpu_pipeline(mp, g, where, fused=False, table_build="fill"):
    seed, B, H, Hkv, n, M, D, K, L = (int(x) for x in g["meta"])
    keys, kns, vals, W, qb = cases.case_inputs(seed, B, H, Hkv, n, D, K, L, cases.golden_data(g))
    dev = "cuda"
    sh = mp.SimHash(bf16_t(W, dev), K, L)
    lsh = mp.LSH()
    lsh.alloc(K, L, 2, H, Hkv, B, M)
    srv = mp.SparseAttentionServer()
    srv.alloc(2, H, Hkv, D, B, M)
    layer = 1
    kcodes = []
    for b in range(B):
        kc = sh.keys(bf16_t(keys[b], dev))                                   # int16 [Hkv, L, n]
        kcodes.append(kc.cpu().numpy())
        if table_build == "fastfill":
            lsh.fastfill(layer, b, kc if where == "cuda" else kc.cpu())
        else:
            sc, si = kc.sort(dim=-1, stable=True)
            sc, si = sc.contiguous(), si.int().contiguous()
            if where == "cpu":
                sc, si = sc.cpu(), si.cpu()
            lsh.fill(layer, b, sc, si)
        k_t, v_t, kn_t = bf16_t(keys[b], where), bf16_t(vals[b], where), torch.from_numpy(kns[b]).to(where)
        srv.fill(layer, b, k_t, v_t, kn_t)
    q_t = bf16_t(qb, where)
    qcodes, qn = sh.query(q_t)
    results = torch.zeros((B * H, M), dtype=torch.int32, device=where)
    nnz = torch.zeros((B * H,), dtype=torch.int32, device=where)
    lsh.batch_retrieve(layer, qcodes, results, nnz)
    out = torch.zeros((B * H, D), dtype=torch.bfloat16, device=where)
    mve = torch.zeros((2, B * H), dtype=torch.float32, device=where)
    srv.attention_wrapper(layer, K, L, out, mve, q_t, qn, results, nnz)
    probs = srv.get_score().reshape(B * H, M)
    torch.cuda.synchronize()
    return dict(qcodes=qcodes.cpu().numpy(), kcodes=np.stack(kcodes), nnz=nnz.cpu().numpy(),
                results=results.cpu().numpy(), out=bits_of(out), mve=mve.cpu().numpy(),
                probs=probs.cpu().numpy(), qn=qn.cpu().numpy(), mask=lsh.get_mask().numpy(),
                dims=(B, H, Hkv, n, M, D, K, L), inputs=(keys, kns, vals, W, qb), srv=srv, lsh=lsh)


def _oracle_attention(r, ind, nnz, exp_mode=0):
    B, H, Hkv, n, M, D, K, L = r["dims"]
    keys, kns, vals, W, qb = r["inputs"]
    srv = oracle.SparseAttentionServer(exp_mode=exp_mode, clamp_cos=1)
    srv.alloc(1, H, Hkv, D, B, M)
    for b in range(B):
        srv.fill(0, b, keys[b], vals[b], kns[b])
    out = np.zeros((B * H, D), np.uint16)
    mve = np.zeros((2, B * H), np.float32)
    _, qn = oracle.simhash_query(qb, W, K, L)
    srv.attention_wrapper(0, K, L, out, mve, qb, qn, np.ascontiguousarray(ind), np.ascontiguousarray(nnz))
    return out, mve, srv.get_score().reshape(B * H, M)


def _check_attention(r, g=None):
    """(a) tight: against the oracle with the importance weight evaluated cancellation-free in
    f64 (exp_mode bit 1) -- what the HIP kernel computes; (b) against the oracle's literal
    restatement of the reference's f32 formula, whose own cancellation noise (~1e-3 relative in
    w + 1e-4) only supports the reference's test tolerance; (c) against the reference's outputs."""
    B, H, Hkv, n, M, D, K, L = r["dims"]
    nnz = r["nnz"]
    live = nnz > 0
    a = synth.bf16_bits_to_f32(r["out"])
    o_out, o_mve, o_probs = _oracle_attention(r, r["results"], nnz, exp_mode=2)
    for h in range(B * H):
        z = nnz[h]
        assert np.allclose(r["probs"][h, :z], o_probs[h, :z], rtol=1e-3, atol=1e-7), h
        assert abs(r["probs"][h, :z].sum() - 1) < 1e-4 or z == 0
    assert np.allclose(r["mve"][1], o_mve[1], atol=1e-3)                      # base-2 LSE
    assert np.allclose(r["mve"][0][live], o_mve[0][live], atol=1e-3)
    assert np.allclose(a, synth.bf16_bits_to_f32(o_out), rtol=2 ** -7, atol=2e-4)   # <= 1 bf16 ulp
    assert (r["out"] == o_out).mean() > 0.9
    f_out, f_mve, f_probs = _oracle_attention(r, r["results"], nnz, exp_mode=0)
    for h in range(B * H):
        z = nnz[h]
        assert np.allclose(r["probs"][h, :z], f_probs[h, :z], rtol=1e-2, atol=1e-2), h   # test_sparse.py:87
    assert np.allclose(a, synth.bf16_bits_to_f32(f_out), rtol=1e-2, atol=1e-2)           # test_sparse.py:92
    assert np.allclose(r["mve"][1], f_mve[1], atol=5e-3)
    if g is not None:   # the reference's own outputs at the reference's own tolerance
        assert np.allclose(a, synth.bf16_bits_to_f32(g["out_bits"]), rtol=1e-2, atol=1e-2)
        assert np.allclose(r["mve"][1], g["mve"][1], atol=0.03)


@pytest.mark.parametrize("name", PIPE)
@pytest.mark.parametrize("where", ["cuda", "cpu"])
def test_pipeline_vs_reference_and_oracle(mp, name, where):
    g = cases.load_golden(name)
    r = _gpu_pipeline(mp, g, where)
    B, H, Hkv, n, M, D, K, L = r["dims"]
    # integer work: bit-exact against the reference
    assert np.array_equal(r["qcodes"], g["qcodes"])
    assert np.array_equal(
        np.frombuffer(hashlib.sha256(r["kcodes"].tobytes()).digest(), np.uint8), g["kcodes_sha"])
    cases.check_sign_ties(g, r["kcodes"], K)
    assert np.array_equal(r["nnz"], g["nnz"])
    ref_lists = cases.split_ragged(g["results_ref_order"], g["nnz"])
    for h in range(B * H):
        got = r["results"][h, :r["nnz"][h]]
        assert np.array_equal(got, np.sort(ref_lists[h]))              # same set, ascending order
        hist = np.bincount(r["mask"].reshape(B * H, M)[h].astype(np.int64), minlength=3)
        assert np.array_equal(hist, g["mask_hist"][h])                 # get_mask counters
    _check_attention(r, g)


@pytest.mark.parametrize("pinned", [True, False])
@pytest.mark.parametrize("mode", ["zero_copy", "flag_wait", "staged"])
def test_host_buffer_modes_agree(mp, mode, pinned):
    """MP_MEM_HOST calls (the unchanged caller of models/attnserver.py:299-300) give what the device-buffer calls
    give, bit for bit, whichever way the buffers cross PCIe: kernels working on the caller's PINNED tensors in place
    (:61-66), on the handle's pinned mirror for PAGEABLE ones (`results_lsh_cpu`, `nnz`, :59-60; the default), with the
    completion word instead of a stream synchronisation (`host_flag_wait`), or through staged copies (`host_zero_copy = 0`); rows behind nnz stay untouched; a second call with other queries works as the first."""
    import magicpig_amd._lib as L_

    g = cases.load_golden("gqa_32h")
    seed, B, H, Hkv, n, M, D, K, L = (int(x) for x in g["meta"])
    keys, kns, vals, W, qb = cases.case_inputs(seed, B, H, Hkv, n, D, K, L)
    BH = B * H
    sh = mp.SimHash(bf16_t(W, "cuda"), K, L)
    lsh, srv = mp.LSH(), mp.SparseAttentionServer()
    lsh.alloc(K, L, 1, H, Hkv, B, M)
    srv.alloc(1, H, Hkv, D, B, M)
    lsh.fastfill(0, 0, sh.keys(bf16_t(keys[0], "cuda")))
    srv.fill(0, 0, bf16_t(keys[0], "cuda"), bf16_t(vals[0], "cuda"), torch.from_numpy(kns[0]).cuda())
    mk = (lambda t: t.pin_memory()) if pinned else (lambda t: t)
    L_.set_option("host_zero_copy", 0 if mode == "staged" else 1)
    L_.set_option("host_flag_wait", 1 if mode == "flag_wait" else 0)     # completion word in pinned memory instead of a sync
    keep = []
    try:
        for rep in range(2):
            q = bf16_t(qb if rep == 0 else np.roll(qb, 3, axis=0), "cuda")
            codes, qn = sh.query(q)
            d_res = torch.full((BH, M), -7, dtype=torch.int32, device="cuda")
            d_nnz = torch.zeros((BH,), dtype=torch.int32, device="cuda")
            lsh.batch_retrieve(0, codes, d_res, d_nnz)
            m_dev = lsh.get_mask()
            d_out = torch.zeros((BH, D), dtype=torch.bfloat16, device="cuda")
            d_mve = torch.zeros((2, BH), dtype=torch.float32, device="cuda")
            srv.attention_wrapper(0, K, L, d_out, d_mve, q, qn, d_res, d_nnz)
            d_probs = srv.get_score().reshape(BH, M).clone()
            h_codes, h_q, h_qn = mk(codes.cpu()), mk(q.cpu()), qn.cpu()
            h_res = mk(torch.full((BH, M), -7, dtype=torch.int32))
            keep.append(h_res)
            h_nnz = torch.zeros((BH,), dtype=torch.int32)
            lsh.batch_retrieve(0, h_codes, h_res, h_nnz)
            assert torch.equal(h_nnz, d_nnz.cpu()) and torch.equal(h_res, d_res.cpu())       # incl. the -7 behind nnz
            assert torch.equal(lsh.get_mask(), m_dev)                  # get_mask re-reads the staged / mapped codes
            h_out, h_mve = mk(torch.zeros((BH, D), dtype=torch.bfloat16)), mk(torch.zeros((2, BH), dtype=torch.float32))
            srv.attention_wrapper(0, K, L, h_out, h_mve, h_q, h_qn, h_res, h_nnz)
            assert torch.equal(h_out, d_out.cpu()) and torch.equal(h_mve, d_mve.cpu())
            assert torch.equal(srv.get_score().reshape(BH, M), d_probs)
            # The attention entry recognises the rows batch_retrieve has just handed out (same pointers, counts and row
            # checksums) and reads their HBM copy instead of uploading them.  A caller that EDITS `ind` in place between
            # the two calls must be served its edit: one id of the longest row is replaced by a token that was not selected.
            r = int(torch.argmax(h_nnz))
            z = int(h_nnz[r])
            assert z >= 2
            other = next(t for t in range(n) if t not in set(h_res[r, :z].tolist()))
            h_res[r, z // 2] = other
            e_res = d_res.clone()
            e_res[r, z // 2] = other
            srv.attention_wrapper(0, K, L, d_out, d_mve, q, qn, e_res, d_nnz)
            srv.attention_wrapper(0, K, L, h_out, h_mve, h_q, h_qn, h_res, h_nnz)
            assert torch.equal(h_out, d_out.cpu()) and torch.equal(h_mve, d_mve.cpu())
            assert not torch.equal(srv.get_score().reshape(BH, M)[r, :z], d_probs[r, :z])
        del lsh, srv
    finally:
        L_.set_option("host_zero_copy", 1)
        L_.set_option("host_flag_wait", 0)


@pytest.mark.parametrize("pinned", [True, False])
def test_host_mode_scores_outlive_the_next_retrieve(mp, pinned):
    """MP_MEM_HOST attention over the rows batch_retrieve has just handed out runs on the lsh handle's HBM copy of rows
    and counts (one launch, verified against the caller's rows while the kernel runs).  That copy is rewritten by the
    handle's next retrieve: get_score afterwards must still normalise with the counts of ITS call."""
    g = cases.load_golden("gqa_32h")
    seed, B, H, Hkv, n, M, D, K, L = (int(x) for x in g["meta"])
    keys, kns, vals, W, qb = cases.case_inputs(seed, B, H, Hkv, n, D, K, L)
    BH = B * H
    sh = mp.SimHash(bf16_t(W, "cuda"), K, L)
    lsh, srv = mp.LSH(), mp.SparseAttentionServer()
    lsh.alloc(K, L, 1, H, Hkv, B, M)
    srv.alloc(1, H, Hkv, D, B, M)
    lsh.fastfill(0, 0, sh.keys(bf16_t(keys[0], "cuda")))
    srv.fill(0, 0, bf16_t(keys[0], "cuda"), bf16_t(vals[0], "cuda"), torch.from_numpy(kns[0]).cuda())
    mk = (lambda t: t.pin_memory()) if pinned else (lambda t: t)
    q = bf16_t(qb, "cuda")
    codes, qn = sh.query(q)
    d_res = torch.zeros((BH, M), dtype=torch.int32, device="cuda")
    d_nnz = torch.zeros((BH,), dtype=torch.int32, device="cuda")
    lsh.batch_retrieve(0, codes, d_res, d_nnz)
    d_out = torch.zeros((BH, D), dtype=torch.bfloat16, device="cuda")
    d_mve = torch.zeros((2, BH), dtype=torch.float32, device="cuda")
    srv.attention_wrapper(0, K, L, d_out, d_mve, q, qn, d_res, d_nnz)
    d_probs = srv.get_score().reshape(BH, M).clone()
    h_res, h_nnz = mk(torch.zeros((BH, M), dtype=torch.int32)), torch.zeros((BH,), dtype=torch.int32)
    lsh.batch_retrieve(0, mk(codes.cpu()), h_res, h_nnz)
    h_out, h_mve = mk(torch.zeros((BH, D), dtype=torch.bfloat16)), mk(torch.zeros((2, BH), dtype=torch.float32))
    srv.attention_wrapper(0, K, L, h_out, h_mve, mk(q.cpu()), qn.cpu(), h_res, h_nnz)
    assert torch.equal(h_out, d_out.cpu())
    # another step's retrieve on the same handle (other codes, other buffers) before the scores are asked for
    q2 = bf16_t(np.roll(qb, 5, axis=0), "cuda")
    codes2, _ = sh.query(q2)
    h_res2, h_nnz2 = mk(torch.zeros((BH, M), dtype=torch.int32)), torch.zeros((BH,), dtype=torch.int32)
    lsh.batch_retrieve(0, mk(codes2.cpu()), h_res2, h_nnz2)
    assert not torch.equal(h_nnz2, h_nnz)
    assert torch.equal(srv.get_score().reshape(BH, M), d_probs)


@pytest.mark.parametrize("pinned", [True, False])
def test_host_fast_path_after_a_replayed_decode_and_checksum_neutral_edits(mp, pinned):
    """ADVICE r04 (capi.hip attn_entry).  (1) Between the caller's batch_retrieve and attention_wrapper on CPU tensors the
    SAME handles run one-launch decodes of OTHER queries -- eagerly and as a replayed hipGraph, which the host side of
    the library never sees: the attention call must still attend over the rows IT was handed (the rows' HBM copy lives
    in buffers only the host-mode retrieve writes), not over the decode's by-product rows.  (2) An in-place edit of
    `ind` that keeps BOTH linear sums of the round-4 checksum (ids +1, -2, +1 on three neighbouring positions: sum and
    position-weighted sum unchanged) must be served, pinned rows (non-linear checksum) and pageable rows (exact compare)."""
    g = cases.load_golden("gqa_32h")
    seed, B, H, Hkv, n, M, D, K, L = (int(x) for x in g["meta"])
    keys, kns, vals, W, qb = cases.case_inputs(seed, B, H, Hkv, n, D, K, L)
    BH = B * H
    sh = mp.SimHash(bf16_t(W, "cuda"), K, L)
    lsh, srv = mp.LSH(), mp.SparseAttentionServer()
    lsh.alloc(K, L, 1, H, Hkv, B, M)
    srv.alloc(1, H, Hkv, D, B, M)
    lsh.fastfill(0, 0, sh.keys(bf16_t(keys[0], "cuda")))
    srv.fill(0, 0, bf16_t(keys[0], "cuda"), bf16_t(vals[0], "cuda"), torch.from_numpy(kns[0]).cuda())
    mk = (lambda t: t.pin_memory()) if pinned else (lambda t: t)
    import magicpig_amd._lib as L_

    def decode(qdev, out, mve):
        L_.check(L_.lib().mp_decode_sparse_layer(sh._h, lsh._h, srv._h, 0, L_.ptr(qdev), L_.ptr(out), L_.ptr(mve), None,
                                                L_.current_stream(qdev)))

    q = bf16_t(qb, "cuda")
    q_other = bf16_t(np.roll(qb, 7, axis=0), "cuda")
    codes, qn = sh.query(q)
    d_res = torch.zeros((BH, M), dtype=torch.int32, device="cuda")
    d_nnz = torch.zeros((BH,), dtype=torch.int32, device="cuda")
    lsh.batch_retrieve(0, codes, d_res, d_nnz)
    d_out = torch.zeros((BH, D), dtype=torch.bfloat16, device="cuda")
    d_mve = torch.zeros((2, BH), dtype=torch.float32, device="cuda")
    srv.attention_wrapper(0, K, L, d_out, d_mve, q, qn, d_res, d_nnz)
    want_out, want_mve = d_out.cpu(), d_mve.cpu()
    want_probs = srv.get_score().reshape(BH, M).clone()
    # a captured decode step of the other queries
    g_out = torch.zeros((BH, D), dtype=torch.bfloat16, device="cuda")
    g_mve = torch.zeros((2, BH), dtype=torch.float32, device="cuda")
    decode(q_other, g_out, g_mve)                          # (packs / warms outside the capture)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        decode(q_other, g_out, g_mve)
    other_out = g_out.clone()
    assert not torch.equal(other_out.cpu(), want_out)
    for between in ("nothing", "eager", "replay"):
        h_res, h_nnz = mk(torch.zeros((BH, M), dtype=torch.int32)), torch.zeros((BH,), dtype=torch.int32)
        lsh.batch_retrieve(0, mk(codes.cpu()), h_res, h_nnz)
        assert torch.equal(h_nnz, d_nnz.cpu())
        if between == "eager":
            decode(q_other, g_out, g_mve)
        elif between == "replay":
            graph.replay()
        torch.cuda.synchronize()
        h_out, h_mve = mk(torch.zeros((BH, D), dtype=torch.bfloat16)), mk(torch.zeros((2, BH), dtype=torch.float32))
        srv.attention_wrapper(0, K, L, h_out, h_mve, mk(q.cpu()), qn.cpu(), h_res, h_nnz)
        assert torch.equal(h_out, want_out) and torch.equal(h_mve, want_mve), between
    # (2) the checksum-neutral edit: three neighbouring entries of the longest row, ids +1, -2, +1
    r = int(torch.argmax(h_nnz))
    z = int(h_nnz[r])
    row = h_res[r, :z].tolist()
    have = set(row)
    j = next(j for j in range(1, z - 2)
             if row[j] + 1 not in have and row[j + 1] - 2 not in have and row[j + 2] + 1 not in have
             and row[j + 1] - 2 >= 0 and row[j + 2] + 1 < n and len({row[j] + 1, row[j + 1] - 2, row[j + 2] + 1}) == 3)
    before = (sum(v + 1 for v in row), sum((v + 1) * (i + 1) for i, v in enumerate(row)))
    h_res[r, j] += 1
    h_res[r, j + 1] -= 2
    h_res[r, j + 2] += 1
    row2 = h_res[r, :z].tolist()
    assert before == (sum(v + 1 for v in row2), sum((v + 1) * (i + 1) for i, v in enumerate(row2)))
    e_res = d_res.clone()
    e_res[r, :z] = h_res[r, :z].cuda()
    srv.attention_wrapper(0, K, L, d_out, d_mve, q, qn, e_res, d_nnz)
    # (three tokens of a few hundred: the bf16 output may not move; the probabilities of the three positions do)
    e_probs = srv.get_score().reshape(BH, M)[r, :z].clone()
    assert not torch.equal(e_probs[j:j + 3], want_probs[r, j:j + 3])
    for name in ("host_fast_hits", "host_fast_edited", "host_fast_unpaired"):
        L_.set_option(name, 0)
    h_out, h_mve = mk(torch.zeros((BH, D), dtype=torch.bfloat16)), mk(torch.zeros((2, BH), dtype=torch.float32))
    srv.attention_wrapper(0, K, L, h_out, h_mve, mk(q.cpu()), qn.cpu(), h_res, h_nnz)
    assert torch.equal(h_out, d_out.cpu()) and torch.equal(h_mve, d_mve.cpu())
    assert torch.equal(srv.get_score().reshape(BH, M)[r, :z], e_probs)           # the EDITED rows were attended over
    assert (L_.get_option("host_fast_hits"), L_.get_option("host_fast_edited")) == (0, 1)
    # and the unedited rows of a fresh retrieve take the fast path again (pinned: the checksums agree; pageable: memcmp)
    lsh.batch_retrieve(0, mk(codes.cpu()), h_res, h_nnz)
    srv.attention_wrapper(0, K, L, h_out, h_mve, mk(q.cpu()), qn.cpu(), h_res, h_nnz)
    assert torch.equal(h_out, want_out) and L_.get_option("host_fast_hits") == 1
    del graph


@pytest.mark.parametrize("name", ["lsh_small", "gqa_32h"])
def test_device_table_build_equals_sorted_fill(mp, name):
    g = cases.load_golden(name)
    r = _gpu_pipeline(mp, g, "cuda", table_build="fastfill")
    assert np.array_equal(r["nnz"], g["nnz"])
    ref_lists = cases.split_ragged(g["results_ref_order"], g["nnz"])
    for h, lst in enumerate(ref_lists):
        assert np.array_equal(r["results"][h, :r["nnz"][h]], np.sort(lst))
    # counting sort is stable: ids ascend inside every bucket
    bounds, table = r["lsh"].get_tables(1)
    bounds, table = bounds.cpu().numpy(), table.cpu().numpy()
    for (gi, l) in [(0, 0), (bounds.shape[0] - 1, bounds.shape[1] - 1)]:
        for b in range(bounds.shape[2]):
            s, e = bounds[gi, l, b, 0], bounds[gi, l, b, -1]
            assert np.all(np.diff(table[gi, l, s:e]) > 0)


@pytest.mark.parametrize("K,L,H,Hkv,B,n,M", [(4, 1100, 4, 2, 1, 600, 640),      # more tables than threads in a workgroup
                                             (15, 12, 8, 8, 2, 3000, 3001),    # widest codes (int16), odd max_length
                                             (11, 300, 8, 1, 1, 5000, 5056)])  # cfg-4 hyper-parameters
def test_retrieve_and_fused_hash_unusual_shapes(mp, K, L, H, Hkv, B, n, M):
    """Sets + nnz bit-exact against the oracle for table counts / code widths / lengths away from the
    headline config, through both entry points (stand-alone calls and the fused decode)."""
    D = 128
    keys, kns, vals, W, qb = cases.case_inputs(500 + K + L, B, H, Hkv, n, D, K, L)
    server = mp.LSHSparseAttnServer(1, H, Hkv, D, K=K, L=L, batch_size=B, num_sink_tokens=0,
                                    num_local_tokens=0, max_length=M, dense_layers=(), hash_func=bf16_t(W, "cuda"))
    olsh = oracle.LSH()
    olsh.alloc(K, L, 1, H, Hkv, B, M)
    for b in range(B):
        server.hash_code_buffer = server.hasher.keys(bf16_t(keys[b], "cuda"))
        kc = server.hash_code_buffer.cpu().numpy()
        assert np.array_equal(kc, oracle.simhash_keys(keys[b], W, K, L))
        server.build_table(0, b, n)
        server.attn_server.fill(0, b, bf16_t(keys[b], "cuda"), bf16_t(vals[b], "cuda"), torch.from_numpy(kns[b]).cuda())
        sc, si = cases.stable_sort_codes(kc)
        olsh.fill(0, b, sc, si)
    qcodes, _ = oracle.simhash_query(qb, W, K, L)
    ores = np.zeros((B * H, M), np.int32)
    onnz = np.zeros((B * H,), np.int32)
    olsh.batch_retrieve(0, qcodes, ores, onnz)
    # stand-alone entry points
    codes, qn = server.hasher.query(bf16_t(qb, "cuda"))
    assert np.array_equal(codes.cpu().numpy(), qcodes)
    res = torch.zeros((B * H, M), dtype=torch.int32, device="cuda")
    nnz = torch.zeros((B * H,), dtype=torch.int32, device="cuda")
    server.lsh_retriever.batch_retrieve(0, codes, res, nnz)
    assert np.array_equal(nnz.cpu().numpy(), onnz)
    r = res.cpu().numpy()
    for h in range(B * H):
        assert np.array_equal(r[h, :onnz[h]], np.sort(ores[h, :onnz[h]]))
    # fused decode entry
    out, lse = server.decode(bf16_t(qb, "cuda").view(B, H, 1, D), 0)
    torch.cuda.synchronize()
    assert np.array_equal(server.nnz.cpu().numpy(), onnz)
    rr = dict(dims=(B, H, Hkv, n, M, D, K, L), inputs=(keys, kns, vals, W, qb), nnz=onnz, results=r,
              out=bits_of(out.reshape(B * H, D)), mve=server.max_value_expsum.cpu().numpy(),
              probs=server.attn_server.get_score().reshape(B * H, M).cpu().numpy())
    _check_attention(rr)


def test_lsh_edge_cases(mp):
    g = cases.load_golden("lsh_edge")
    seed, K, L, H, Hkv, B, n, M = (int(x) for x in g["meta"])
    NB = 1 << K
    codes = synth.randint(seed, 0, NB, (B, Hkv, L, n)).astype(np.int16)
    lsh = mp.LSH()
    lsh.alloc(K, L, 2, H, Hkv, B, M)
    for b in range(B):
        sc, si = cases.stable_sort_codes(codes[b])
        lsh.fill(1, b, torch.from_numpy(sc), torch.from_numpy(si))
    q = g["q"]
    for rep, (nk, sk) in enumerate((("nnz0", "sorted0"), ("nnz1", "sorted1"))):
        qq = q if rep == 0 else ((q + 3) % NB).astype(np.int32)
        results = torch.zeros((B * H, M), dtype=torch.int32)
        nnz = torch.zeros((B * H,), dtype=torch.int32)
        lsh.batch_retrieve(1, torch.from_numpy(np.ascontiguousarray(qq)), results, nnz)
        assert np.array_equal(nnz.numpy(), g[nk])
        got = np.concatenate([results[h, :nnz[h]].numpy() for h in range(B * H)])
        assert np.array_equal(got, g[sk])
    # error behaviour (an addition over the reference): unsorted codes are rejected
    bad = torch.from_numpy(codes[0].copy())
    with pytest.raises(mp.MagicPigError):
        lsh.fill(0, 0, bad, torch.zeros_like(bad, dtype=torch.int32))


def test_attention_edge_cases(mp):
    g = cases.load_golden("attn_edge")
    seed, K, L, H, Hkv, B, n, M, D = (int(x) for x in g["meta"])
    keys, kns, vals, W, qb = cases.case_inputs(seed, B, H, Hkv, n, D, K, L)
    nnz = g["nnz"].astype(np.int32)
    ind = np.zeros((B * H, M), np.int32)
    for h, z in enumerate(nnz):
        perm = np.argsort(synth.u64(seed + 50 + h, n), kind="stable").astype(np.int32)
        ind[h, :z] = perm[:z]
    for where in ("cuda", "cpu"):
        for qdtype in (torch.bfloat16, torch.float32):
            srv = mp.SparseAttentionServer()
            srv.alloc(1, H, Hkv, D, B, M)
            srv.fill(0, 0, bf16_t(keys[0], where), bf16_t(vals[0], where), torch.from_numpy(kns[0]).to(where))
            out = torch.zeros((B * H, D), dtype=torch.bfloat16, device=where)
            mve = torch.zeros((2, B * H), dtype=torch.float32, device=where)
            srv.attention_wrapper(0, K, L, out, mve, bf16_t(qb, where).to(qdtype),
                                  torch.from_numpy(g["qnorm"]).to(where),
                                  torch.from_numpy(ind).to(where), torch.from_numpy(nnz).to(where))
            probs = srv.get_score().reshape(B * H, M).cpu().numpy()
            r = dict(dims=(B, H, Hkv, n, M, D, K, L), inputs=(keys, kns, vals, W, qb), nnz=nnz,
                     results=ind, out=bits_of(out), mve=mve.cpu().numpy(), probs=probs)
            _check_attention(r, g)
            assert not r["out"][0].any() and r["mve"][1, 0] == -np.inf        # nnz == 0 head


# ------------------------------------------------------------------ the reference's own test logic

# the reference's grid, library/lsh/test.py:5-12 (192 combinations; num_layers only picks which layer is used: here
# always 2 layers, layer 1), plus its trailing call (K = 2, L = 4, one head)
_LSH_GRID = [(K, L, s, d, G, b, 32) for K in (4, 8) for L in (50, 100) for s in (1024, 4096, 8192)
             for d in (128, 1024) for G in (4, 8) for b in (1, 4)] + [(2, 4, 128, 16, 1, 1, 1)]


@pytest.mark.parametrize("K,L,seq_len,delta,G,bsz,H", _LSH_GRID)
def test_batch_retrieve_like_reference_test(mp, K, L, seq_len, delta, G, bsz, H):
    """library/lsh/test.py:5-76 restated over the reference's whole grid: random codes, nnz == ((codes == q).sum(L) > 1)
    .sum(), every returned index inside that mask; a second query re-checks the state reset."""
    Hkv = H // G
    NB = 1 << K
    gen = torch.Generator().manual_seed(K * 1000 + L + seq_len + G + bsz)
    lsh = mp.LSH()
    lsh.alloc(K, L, 2, H, Hkv, bsz, seq_len + delta)
    hash_code = torch.randint(0, NB, (bsz, Hkv, L, seq_len), dtype=torch.int16, generator=gen)
    sorted_code, sorted_indices = hash_code.sort()
    for i in range(bsz):
        lsh.fill(1, i, sorted_code[i].contiguous(), sorted_indices[i].int().contiguous())
    hc = hash_code[:, :, None, :, :].repeat(1, 1, G, 1, 1).reshape(bsz * H, L, seq_len).cuda()
    for _ in range(2):
        query = torch.randint(0, NB, (bsz * H, L), dtype=torch.int32, generator=gen)
        results = torch.zeros((bsz * H, seq_len + delta), dtype=torch.int32)
        nnz = torch.zeros((bsz * H,), dtype=torch.int32)
        lsh.batch_retrieve(1, query, results, nnz)
        ref_mask = ((hc == query.cuda()[:, :, None]).int().sum(dim=1) > 1).cpu()
        assert torch.equal(nnz, ref_mask.int().sum(dim=-1).int())
        for i in range(bsz * H):
            ri = results[i][:nnz[i]].long()
            assert ref_mask[i][ri].int().sum() == nnz[i]
            assert torch.all(ri[1:] > ri[:-1])            # ascending, hence duplicate-free


# the reference's grid, library/sparse_attention/test.py:6-14 (288 combinations; num_layers only picks the layer)
_ATTN_GRID = [(G, b, H, D, d, s) for s in (1024, 4096, 8192) for d in (128, 1024) for G in (4, 8) for b in (1, 4)
              for H in (32, 64) for D in (64, 128)]


@pytest.mark.parametrize("G,batch_size,H,D,delta,seq_len", _ATTN_GRID)
def test_sparse_attention_like_reference_test(mp, G, batch_size, H, D, delta, seq_len):
    """library/sparse_attention/test_sparse.py:6-92 / test.py:6-91 restated over the reference's whole grid (f32 and
    bf16 queries alternate) with the reference's torch formula and its rtol = atol = 1e-2."""
    import math

    K, L = 10, 150
    M = seq_len + delta
    Hkv = H // G
    gen = torch.Generator().manual_seed(G * 100 + batch_size * 10 + H + D)
    key = torch.randn((batch_size, Hkv, seq_len, D), generator=gen).to(torch.bfloat16)
    value = torch.randn((batch_size, Hkv, seq_len, D), generator=gen).to(torch.bfloat16)
    key_norm = key.norm(p=2, dim=-1).float()
    srv = mp.SparseAttentionServer()
    srv.alloc(2, H, Hkv, D, batch_size, M)
    for i in range(batch_size):
        srv.fill(1, i, key[i].contiguous(), value[i].contiguous(), key_norm[i].contiguous())
    query = torch.randn((batch_size, H, 1, D), generator=gen)
    if (G + batch_size + H // 32 + D // 64 + seq_len // 1024) % 2 == 0:
        query = query.to(torch.bfloat16)                      # attention_wrapper_bf16's query (models/attnserver.py:300)
    query_norm = query.norm(p=2, dim=-1).float()              # else f32 as library/sparse_attention/test.py:44
    BH = batch_size * H
    nnz = torch.randint(1, seq_len, (BH,), generator=gen).int()
    ind = torch.zeros((BH, M)).int()
    for i in range(BH):
        ind[i][:nnz[i]] = torch.randperm(seq_len, generator=gen)[:nnz[i]].int()
    output = torch.zeros((BH, D), dtype=torch.bfloat16)
    mve = torch.zeros((2, BH), dtype=torch.float32)
    srv.attention_wrapper(1, K, L, output, mve, query.reshape(BH, D).contiguous(),
                          query_norm.reshape(BH).contiguous(), ind, nnz)
    score = srv.get_score().view(BH, M).cpu()
    keyr = key[:, :, None].repeat(1, 1, G, 1, 1).reshape(BH, seq_len, D).cuda().float()
    valr = value[:, :, None].repeat(1, 1, G, 1, 1).reshape(BH, seq_len, D).cuda().float()
    knr = key_norm[:, :, None].repeat(1, 1, G, 1).reshape(BH, seq_len).cuda()
    qf = query.reshape(BH, D).cuda().float()
    for i in range(0, BH, max(1, BH // 48)):                  # <= ~48 heads per combination checked densely
        idx = ind[i][:nnz[i]].long().cuda()
        ref = keyr[i][idx] @ qf[i]
        cs = ref / (query_norm.reshape(BH)[i].cuda() * knr[i][idx])
        w = 1 - torch.arccos(cs) / torch.pi
        w = 1 - (1 - w ** K) ** L - L * ((1 - w ** K) ** (L - 1)) * (w ** K)
        ref = ref / math.sqrt(D) - torch.log(w + 1e-4)
        lse2 = torch.logsumexp(ref, 0) / math.log(2)
        p = torch.softmax(ref, dim=-1)
        assert torch.allclose(score[i][:nnz[i]], p.cpu(), rtol=1e-2, atol=1e-2)
        assert torch.abs(score[i][:nnz[i]].sum() - 1) <= 1e-2
        o = (p.unsqueeze(0) @ valr[i][idx]).cpu()
        assert torch.allclose(output[i].float(), o, rtol=1e-2, atol=1e-2)
        assert abs(float(mve[1][i]) - float(lse2)) < 1e-2          # the reference never asserts its LSE


@pytest.mark.parametrize("head_kernel", [0, 1])
def test_in_launch_merge_is_deterministic_under_load(mp, head_kernel):
    """head_kernel = 0: the split-KV attention kernel merges a head's slice partials inside the launch
    (last-arriver ticket, write-through partials).  Any stale read shows up as run-to-run differences: 60
    launches over 256 heads with ragged list lengths (1 .. 6000 entries, i.e. 1 .. 94 slices per head)
    must be bit-identical, and equal to the oracle.  At 256 heads the handle would pick the
    one-workgroup-per-head kernel by itself (head_kernel = 1, the same load without tickets), so the
    choice is forced through mp_debug_set_option for both."""
    import magicpig_amd._lib as L_

    L_.set_option("attn_head_kernel", head_kernel)
    try:
        _in_launch_merge_under_load(mp)
    finally:
        L_.set_option("attn_head_kernel", -1)


def _in_launch_merge_under_load(mp):
    H, Hkv, B, D, n, M, K, L = 32, 8, 8, 128, 8000, 8192, 10, 150
    gen = torch.Generator().manual_seed(99)
    key = torch.randn((B, Hkv, n, D), generator=gen).to(torch.bfloat16)
    val = torch.randn((B, Hkv, n, D), generator=gen).to(torch.bfloat16)
    kn = key.norm(p=2, dim=-1).float()
    srv = mp.SparseAttentionServer()
    srv.alloc(1, H, Hkv, D, B, M)
    for b in range(B):
        srv.fill(0, b, key[b].cuda(), val[b].cuda(), kn[b].cuda())
    BH = B * H
    q = torch.randn((BH, D), generator=gen).to(torch.bfloat16)
    qn = q.float().norm(p=2, dim=-1)
    nnz = torch.randint(1, 6000, (BH,), generator=gen).int()
    nnz[5] = 0
    nnz[17] = 1
    ind = torch.zeros((BH, M), dtype=torch.int32)
    for i in range(BH):
        ind[i, :nnz[i]] = torch.randperm(n, generator=gen)[:nnz[i]].int()
    qd, qnd, indd, nnzd = q.cuda(), qn.cuda(), ind.cuda(), nnz.cuda()
    first = None
    for rep in range(60):
        out = torch.zeros((BH, D), dtype=torch.bfloat16, device="cuda")
        mve = torch.zeros((2, BH), dtype=torch.float32, device="cuda")
        srv.attention_wrapper(0, K, L, out, mve, qd, qnd, indd, nnzd)
        cur = (out.view(torch.int16).cpu(), mve.cpu())
        if first is None:
            first = cur
        else:
            assert torch.equal(cur[0], first[0]) and torch.equal(cur[1], first[1]), rep
    got = synth.bf16_bits_to_f32(first[0].numpy().view(np.uint16))
    # (a) against the cancellation-free f64 evaluation of the importance weight: <= 1 bf16 ulp;
    # (b) against the reference's literal f32 formula: its own ~1e-3 noise in w + 1e-4 on these
    #     low-collision-probability tokens allows only the reference's tolerance (1e-2).
    for mode, rt, at in ((2, 2 ** -7, 2e-4), (0, 1e-2, 2e-3)):
        osrv = oracle.SparseAttentionServer(exp_mode=mode, clamp_cos=1)
        osrv.alloc(1, H, Hkv, D, B, M)
        for b in range(B):
            osrv.fill(0, b, key[b], val[b], kn[b])
        oout = np.zeros((BH, D), np.uint16)
        omve = np.zeros((2, BH), np.float32)
        osrv.attention_wrapper(0, K, L, oout, omve, q, qn, ind, nnz)
        ref = synth.bf16_bits_to_f32(oout)
        bad = np.argwhere(~np.isclose(got, ref, rtol=rt, atol=at))
        assert len(bad) == 0, (mode, [(int(h), int(d), int(nnz[h]), float(got[h, d]), float(ref[h, d]))
                                      for h, d in bad[:8]])
        assert np.allclose(first[1].numpy()[1], omve[1], atol=1e-3 if mode == 2 else 5e-3)


def test_full_attention_vs_oracle(mp):
    """library/sparse_attention/test_dense.py restated against the oracle."""
    H, Hkv, B, D, n, M = 8, 2, 2, 128, 700, 768
    keys, kns, vals, W, qb = cases.case_inputs(71, B, H, Hkv, n, D, 10, 20)
    srv = mp.SparseAttentionServer()
    srv.alloc(1, H, Hkv, D, B, M)
    osrv = oracle.SparseAttentionServer()
    osrv.alloc(1, H, Hkv, D, B, M)
    for b in range(B):
        srv.fill(0, b, bf16_t(keys[b], "cuda"), bf16_t(vals[b], "cuda"), torch.from_numpy(kns[b]).cuda())
        osrv.fill(0, b, keys[b], vals[b], kns[b])
    nnz = np.array([n, 1, 300, 0, 257, 64, 699, 5] * B, np.int32)
    qf = synth.bf16_bits_to_f32(qb)
    out = torch.zeros((B * H, D), dtype=torch.bfloat16, device="cuda")
    mve = torch.zeros((2, B * H), dtype=torch.float32, device="cuda")
    srv.full_attention(0, out, mve, torch.from_numpy(qf).cuda(), torch.from_numpy(nnz).cuda())
    oout = np.zeros((B * H, D), np.uint16)
    omve = np.zeros((2, B * H), np.float32)
    osrv.full_attention(0, oout, omve, qf, nnz)
    assert np.allclose(mve.cpu().numpy()[1], omve[1], atol=1e-3)
    assert np.allclose(synth.bf16_bits_to_f32(bits_of(out)), synth.bf16_bits_to_f32(oout),
                       rtol=2 ** -7, atol=2e-4)
    probs = srv.get_score().reshape(B * H, M).cpu().numpy()
    oprobs = osrv.get_score().reshape(B * H, M)
    for h in range(B * H):
        assert np.allclose(probs[h, :nnz[h]], oprobs[h, :nnz[h]], rtol=2e-3, atol=1e-8)


def test_merge_state_vs_oracle(mp):
    R, D = 37, 128
    va = synth.normal_bf16_bits(81, (R, D))
    vb = synth.normal_bf16_bits(82, (R, D))
    sa = synth.normal_f32(83, (R,)) * 5
    sb = synth.normal_f32(84, (R,)) * 5
    sb[3] = -np.inf
    sa[5] = -np.inf
    sa[7] = sb[7] = -np.inf
    v, s = mp.LSHSparseAttnServer.merge(bf16_t(va, "cuda"), torch.from_numpy(sa).cuda(),
                                        bf16_t(vb, "cuda"), torch.from_numpy(sb).cuda())
    ov, os_ = oracle.merge_state(va, sa, vb, sb)
    assert np.allclose(s.cpu().numpy(), os_, atol=1e-5, equal_nan=True)
    assert np.allclose(synth.bf16_bits_to_f32(bits_of(v)), synth.bf16_bits_to_f32(ov), rtol=2 ** -7, atol=1e-6)


# ------------------------------------------------------------------ fused decode step (the hot path as benchmarked)

@pytest.mark.parametrize("name", ["gqa_32h", "b2_k8_l60", "cfg2_small", "cfg3_small", "cfg4_small", "g8_hkv2",
                                  "skew_small", "clustered_k10"])
@pytest.mark.parametrize("table_build", ["sort", "counting"])
def test_fused_decode_layer(mp, name, table_build):
    """The one-launch decode entry against the reference's vectors and the oracle.  gqa_32h / b2_k8_l60
    run it with workgroup clusters (B*H = 32 / 16), cfg2_small / cfg3_small with B*H = 256 heads: one
    workgroup per head, no cluster -- the regime of BASELINE cfg 2 and cfg 3."""
    g = cases.load_golden(name)
    seed, B, H, Hkv, n, M, D, K, L = (int(x) for x in g["meta"])
    keys, kns, vals, W, qb = cases.case_inputs(seed, B, H, Hkv, n, D, K, L, cases.golden_data(g))
    server = mp.LSHSparseAttnServer(3, H, Hkv, D, K=K, L=L, batch_size=B, num_sink_tokens=0,
                                    num_local_tokens=0, max_length=M, dense_layers=(0,),
                                    hash_func=bf16_t(W, "cuda"), table_build=table_build)
    # feed already-centred keys through the hot-path stores directly (the centring of
    # LSHSparseAttnServer.fill is exercised in test_server_fill_centres_keys)
    for b in range(B):
        server.hash_code_buffer = server.hasher.keys(bf16_t(keys[b], "cuda"))
        server.build_table(2, b, n)
        server.attn_server.fill(2, b, bf16_t(keys[b], "cuda"), bf16_t(vals[b], "cuda"),
                                torch.from_numpy(kns[b]).cuda())
    out, lse = server.decode(bf16_t(qb, "cuda").view(B, H, 1, D), 2)
    torch.cuda.synchronize()
    assert np.array_equal(server.nnz.cpu().numpy(), g["nnz"])
    r = dict(dims=(B, H, Hkv, n, M, D, K, L), inputs=(keys, kns, vals, W, qb), nnz=g["nnz"],
             results=server.lsh_retriever  # placeholder, replaced below
             )
    ref_lists = cases.split_ragged(g["results_ref_order"], g["nnz"])
    ind = np.zeros((B * H, M), np.int32)
    for h, lst in enumerate(ref_lists):
        ind[h, :len(lst)] = np.sort(lst)
    r["results"] = ind
    r["out"] = bits_of(out.reshape(B * H, D))
    r["mve"] = server.max_value_expsum.cpu().numpy()
    r["probs"] = server.attn_server.get_score().reshape(B * H, M).cpu().numpy()
    _check_attention(r, g)
    assert np.allclose(lse.cpu().numpy().reshape(-1), r["mve"][1])


def _fused_server(mp, B, H, Hkv, n, M, D, K, L, seed):
    keys, kns, vals, W, qb = cases.case_inputs(seed, B, H, Hkv, n, D, K, L)
    server = mp.LSHSparseAttnServer(1, H, Hkv, D, K=K, L=L, batch_size=B, num_sink_tokens=0,
                                    num_local_tokens=0, max_length=M, dense_layers=(),
                                    hash_func=bf16_t(W, "cuda"))
    for b in range(B):
        server.hash_code_buffer = server.hasher.keys(bf16_t(keys[b], "cuda"))
        server.build_table(0, b, n)
        server.attn_server.fill(0, b, bf16_t(keys[b], "cuda"), bf16_t(vals[b], "cuda"),
                                torch.from_numpy(kns[b]).cuda())
    return server, (keys, kns, vals, W, qb)


@pytest.mark.parametrize("B,H,Hkv", [(1, 32, 8), (1, 8, 2), (2, 6, 3)])
def test_fused_decode_cluster_handoff_is_deterministic(mp, B, H, Hkv):
    """The one-launch decode entry spreads a head over a cluster of workgroups whose states meet
    through L2 (the grid pads B*H to a multiple of 8 so that the members share an XCD): 40 launches
    on changing queries, each repeated, must be bit-identical run to run and agree with the
    two-kernel path (same ids, same math, different summation order)."""
    n, M, D, K, L = 6000, 6144, 128, 8, 75
    server, (keys, kns, vals, W, qb) = _fused_server(mp, B, H, Hkv, n, M, D, K, L, 4242)
    BH = B * H
    gen = torch.Generator(device="cuda").manual_seed(7)
    for it in range(20):
        q = torch.randn((B, H, 1, D), device="cuda", generator=gen).to(torch.bfloat16)
        out1, lse1 = server.decode(q, 0)
        o1, l1, z1 = out1.clone(), lse1.clone(), server.nnz.clone()
        out2, lse2 = server.decode(q, 0)
        assert torch.equal(o1, out2) and torch.equal(l1, lse2)
        # two-kernel reference of the same step: standalone hash, retrieve, attention_wrapper
        codes, qn = server.hasher.query(q.reshape(BH, D))
        res = torch.zeros((BH, M), dtype=torch.int32, device="cuda")
        nz = torch.zeros((BH,), dtype=torch.int32, device="cuda")
        server.lsh_retriever.batch_retrieve(0, codes, res, nz)
        assert torch.equal(nz, z1)
        o_ref = torch.zeros((BH, D), dtype=torch.bfloat16, device="cuda")
        mve = torch.zeros((2, BH), dtype=torch.float32, device="cuda")
        server.attn_server.attention_wrapper(0, K, L, o_ref, mve, q.reshape(BH, D), qn, res, nz)
        live = (nz > 0).cpu().numpy()
        a = o1.reshape(BH, D).float().cpu().numpy()[live]
        b_ = o_ref.float().cpu().numpy()[live]
        assert np.allclose(a, b_, rtol=2 ** -6, atol=2e-3)
        assert np.allclose(l1.reshape(-1).cpu().numpy()[live], mve[1].cpu().numpy()[live], atol=2e-3)


@pytest.mark.parametrize("B,H,Hkv", [(1, 32, 8), (1, 8, 2)])
def test_fused_decode_under_graph_replay_keeps_the_cluster_checks_quiet(mp, B, H, Hkv):
    """A captured decode step replayed back to back: the block -> XCD round robin does not start where it did
    in the eager probe launches, which a check against the probe's absolute map reported as a misplaced cluster
    on every replay.  The members still share an XCD: outputs must equal the eager launch bit for bit, the
    per-launch placement check (mp_attn_check) must stay quiet, and get_score's compaction must still read the
    per-member counts (their top byte now carries the XCC_ID)."""  # get_score itself is a host-tracked view: eager
    n, M, D, K, L = 6000, 6144, 128, 8, 75
    server, _ = _fused_server(mp, B, H, Hkv, n, M, D, K, L, 777)
    BH = B * H
    assert server.lsh_retriever.R > 1
    gen = torch.Generator(device="cuda").manual_seed(11)
    qs = torch.randn((6, B, H, 1, D), device="cuda", generator=gen).to(torch.bfloat16)
    q_static = qs[0].clone()
    server.collect_nnz = False
    eager = []
    for i in range(6):
        o, l = server.decode(qs[i], 0)
        eager.append((o.clone(), l.clone()))
    server.attn_server.check()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        server.decode(q_static, 0)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(4):                      # four back-to-back launches per replay
            o_g, l_g = server.decode(q_static, 0)
    for rep in range(5):
        for i in range(6):
            q_static.copy_(qs[i])
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(o_g, eager[i][0]) and torch.equal(l_g, eager[i][1])
    server.attn_server.check()                  # raises MP_ERR_STATE if a cluster was seen on two XCDs
    # the compaction of the per-member segments (get_score) reads the same count words, top byte masked
    server.collect_nnz = True
    server.decode(qs[0], 0)
    sc = server.attn_server.get_score().reshape(BH, -1)
    nnz = server.nnz.cpu().numpy()
    assert nnz.max() > 0
    for h in range(BH):
        assert nnz[h] == 0 or float(sc[h, :nnz[h]].sum()) == pytest.approx(1.0, abs=1e-3)


@pytest.mark.parametrize("B,H,Hkv,D,K,L,n,M", [
    (1, 4, 2, 64, 4, 1100, 600, 640),        # more tables than threads in a workgroup, head_dim 64
    (2, 8, 8, 128, 15, 12, 3000, 3001),      # widest codes, odd max_length, no GQA
    (3, 6, 2, 64, 9, 33, 2000, 2048),        # B*H not a multiple of 8: padded grid, head_dim 64
    (1, 1, 1, 128, 6, 50, 70, 64 * 3),       # one head, a list shorter than the cluster
])
def test_fused_decode_unusual_shapes_equal_two_kernel_path(mp, B, H, Hkv, D, K, L, n, M):
    """The one-launch decode entry against hash -> batch_retrieve -> attention_wrapper on the same
    stores: identical codes, nnz and ids; outputs equal up to summation order."""
    server, _ = _fused_server(mp, B, H, Hkv, n, M, D, K, L, 1000 + K)
    BH = B * H
    gen = torch.Generator(device="cuda").manual_seed(K * L)
    for it in range(3):
        q = torch.randn((B, H, 1, D), device="cuda", generator=gen).to(torch.bfloat16)
        out, lse = server.decode(q, 0)
        nz1 = server.nnz.clone()
        codes, qn = server.hasher.query(q.reshape(BH, D))
        res = torch.zeros((BH, M), dtype=torch.int32, device="cuda")
        nz = torch.zeros((BH,), dtype=torch.int32, device="cuda")
        server.lsh_retriever.batch_retrieve(0, codes, res, nz)
        assert torch.equal(nz, nz1)
        o_ref = torch.zeros((BH, D), dtype=torch.bfloat16, device="cuda")
        mve = torch.zeros((2, BH), dtype=torch.float32, device="cuda")
        server.attn_server.attention_wrapper(0, K, L, o_ref, mve, q.reshape(BH, D), qn, res, nz)
        live = (nz > 0).cpu().numpy()
        dead = ~live
        a = out.reshape(BH, D).float().cpu().numpy()
        assert np.allclose(a[live], o_ref.float().cpu().numpy()[live], rtol=2 ** -6, atol=2e-3)
        assert np.all(a[dead] == 0) and np.all(np.isneginf(lse.reshape(-1).cpu().numpy()[dead]))
        assert np.allclose(lse.reshape(-1).cpu().numpy()[live], mve[1].cpu().numpy()[live], atol=2e-3)


@pytest.mark.parametrize("direct", [1, 0])
@pytest.mark.parametrize("K,L,n,M,cluster", [
    (4, 30, 6000, 6144, 8),        # 16 buckets: every piece overflows its 30-id slot
    (6, 75, 6000, 6144, 8),        # mean piece 12 ids: a mix of both
    (10, 150, 20000, 20480, 8),    # mean piece 2.5 ids: 32-byte slots (6 ids)
    (6, 75, 6000, 6144, 16),       # 16 workgroups per head: mean piece 6 ids in 128-byte slots
    (7, 300, 6000, 6144, 16),      # mean piece 3: 64-byte slots (14 ids), L = 300 in one round of five loads per wave
    (4, 30, 6000, 6144, 32),       # 32 workgroups per head, 16 buckets: 12-id pieces overflow the chunk pool's way in
    (8, 150, 20000, 20480, 32),    # mean piece 2.5 ids: 32-byte slots, 32 members
    (11, 300, 20000, 20480, 32)])  # cfg 4's K and L: 52 hash units over 32 members, mean piece 0.3 ids
def test_fused_decode_direct_slots_and_overflow(mp, K, L, n, M, cluster, direct):
    """R = 8 / 16 / 32 workgroups per head with the direct piece slots (128, 64 or 32 bytes by the mean piece length)
    forced on (also where pieces are far longer than a slot: the rest of such a piece comes through the sub-bounds +
    chunk pool) and forced off (sub-bounds only), against hash -> batch_retrieve -> attention_wrapper on the same
    stores: same nnz, same ids, outputs equal up to summation order."""
    import magicpig_amd._lib as L_

    B, H, Hkv, D = 1, 8, 2, 128
    L_.set_option("decode_direct", direct)
    L_.set_option("decode_cluster", cluster)
    try:
        server, _ = _fused_server(mp, B, H, Hkv, n, M, D, K, L, 3000 + K)
    finally:
        L_.set_option("decode_direct", -1)
        L_.set_option("decode_cluster", 0)
    assert server.lsh_retriever.R == cluster
    BH = B * H
    gen = torch.Generator(device="cuda").manual_seed(K * L + direct)
    for it in range(3):
        q = torch.randn((B, H, 1, D), device="cuda", generator=gen).to(torch.bfloat16)
        out, lse = server.decode(q, 0)
        nz1 = server.nnz.clone()
        probs1 = server.attn_server.get_score().reshape(BH, M).clone()
        codes, qn = server.hasher.query(q.reshape(BH, D))
        res = torch.zeros((BH, M), dtype=torch.int32, device="cuda")
        nz = torch.zeros((BH,), dtype=torch.int32, device="cuda")
        server.lsh_retriever.batch_retrieve(0, codes, res, nz)
        assert torch.equal(nz, nz1)
        o_ref = torch.zeros((BH, D), dtype=torch.bfloat16, device="cuda")
        mve = torch.zeros((2, BH), dtype=torch.float32, device="cuda")
        server.attn_server.attention_wrapper(0, K, L, o_ref, mve, q.reshape(BH, D), qn, res, nz)
        probs2 = server.attn_server.get_score().reshape(BH, M)
        live = (nz > 0).cpu().numpy()
        a = out.reshape(BH, D).float().cpu().numpy()
        assert np.allclose(a[live], o_ref.float().cpu().numpy()[live], rtol=2 ** -6, atol=2e-3)
        assert np.allclose(lse.reshape(-1).cpu().numpy()[live], mve[1].cpu().numpy()[live], atol=2e-3)
        for h in range(BH):      # get_score after the decode entry: per-member segments compacted into `ind` order
            z = int(nz[h])
            assert np.allclose(probs1[h, :z].cpu().numpy(), probs2[h, :z].cpu().numpy(), rtol=2e-3, atol=1e-7)


@pytest.mark.parametrize("B,H,Hkv,D,K,L", [(1, 32, 8, 128, 10, 150), (1, 8, 1, 128, 11, 300), (1, 16, 4, 64, 9, 40),
                                           (2, 8, 2, 128, 7, 26)])
def test_decode_with_planes_split_over_the_cluster(mp, B, H, Hkv, D, K, L):
    """decode_split_hash: every member of a head's cluster evaluates 1/R of the hyperplanes and the sign bits are
    exchanged through the XCD's L2 (64-bit words tagged with the launch's sequence number).  Modes 1 (split), 2
    (split, but nobody publishes: every member times out and hashes alone) and 0 (off) must give bit-identical
    codes, nnz, outputs and LSE, launch after launch (the sequence number advances)."""
    import magicpig_amd._lib as L_

    n, M = 6000, 6144
    server, _ = _fused_server(mp, B, H, Hkv, n, M, D, K, L, 600 + K)
    assert server.lsh_retriever.R > 1
    BH = B * H
    gen = torch.Generator(device="cuda").manual_seed(K + L)
    qs = torch.randn((5, B, H, 1, D), device="cuda", generator=gen).to(torch.bfloat16)
    ref = []
    try:
        for mode in (0, 1, 2, 1):
            L_.set_option("decode_split_hash", mode)
            for i in range(5):
                out, lse = server.decode(qs[i], 0)
                got = (out.clone(), lse.clone(), server.nnz.clone())
                if mode == 0:
                    ref.append(got)
                else:
                    assert torch.equal(got[2], ref[i][2]), (mode, i)
                    assert torch.equal(got[0], ref[i][0]) and torch.equal(got[1], ref[i][1]), (mode, i)
        server.attn_server.check()
    finally:
        L_.set_option("decode_split_hash", -1)
    assert int(torch.stack([r[2] for r in ref]).sum()) > 0


@pytest.mark.parametrize("B,H,Hkv,D", [(1, 32, 8, 128), (8, 32, 8, 128), (2, 6, 3, 64)])
def test_decode_with_mfma_hash_launch_equals_fused_hash(mp, B, H, Hkv, D):
    """The decode entry with the query SimHash computed by the MFMA kernel in a launch of its own
    (decode_mfma_hash option: the A/B variant north_star's "the projection uses MFMA" asks about) against the
    default, where the hash is the decode kernel's prologue: same codes, nnz, ids.  ||q|| (a factor of every cosine) is
    the exact f32 norm in the MFMA kernel and, since round 4, within 5e-7 of it in the fused prologue (the fast
    normalisation keeps the bf16 ROW exact, not the last ulps of the norm): outputs agree to a bf16 ulp, not bit for bit."""
    import magicpig_amd._lib as L_

    n, M, K, L = 5000, 5120, 9, 40
    server, _ = _fused_server(mp, B, H, Hkv, n, M, D, K, L, 555)
    gen = torch.Generator(device="cuda").manual_seed(9)
    for it in range(3):
        q = torch.randn((B, H, 1, D), device="cuda", generator=gen).to(torch.bfloat16)
        out, lse = server.decode(q, 0)
        o1, l1, z1 = out.clone(), lse.clone(), server.nnz.clone()
        m1 = server.lsh_retriever.get_mask().clone()
        L_.set_option("decode_mfma_hash", 1)
        try:
            out2, lse2 = server.decode(q, 0)
            torch.cuda.synchronize()
        finally:
            L_.set_option("decode_mfma_hash", 0)
        assert torch.equal(server.nnz, z1)
        assert torch.allclose(out2.float(), o1.float(), rtol=2 ** -7, atol=1e-4) and torch.allclose(lse2, l1, atol=1e-5)
        assert torch.equal(server.lsh_retriever.get_mask(), m1)        # recomputed from the codes each variant wrote


def test_cfg1_shaped_fused_decode_properties(mp):
    """BASELINE cfg 1 shape (B=1, H=32, Hkv=8, n=97 932, M=98 304, K=10, L=150), one layer, through
    size-independent properties: (1) the one-launch entry equals hash -> batch_retrieve ->
    attention_wrapper on the same stores (codes, nnz, ids bit for bit; outputs up to summation order);
    (2) attention_wrapper is a function of the SET of selected ids: a shuffled id list gives the same
    output; (3) scaling V by 2 scales the output by 2 exactly and leaves the LSE unchanged."""
    B, H, Hkv, D, K, L, n, M = 1, 32, 8, 128, 10, 150, 97932, 98304
    BH = B * H
    dev = torch.device("cuda")
    gen = torch.Generator(device=dev).manual_seed(5)
    W = torch.randn((D, K * L), device=dev, generator=gen).to(torch.bfloat16)
    server = mp.LSHSparseAttnServer(1, H, Hkv, D, K=K, L=L, batch_size=B, num_sink_tokens=0,
                                    num_local_tokens=0, max_length=M, dense_layers=(), hash_func=W)
    kc = torch.randn((n, Hkv, D), device=dev, generator=gen).to(torch.bfloat16)
    vc = torch.randn((n, Hkv, D), device=dev, generator=gen).to(torch.bfloat16)
    server.fill(0, 0, kc, vc, n)
    server.build_table(0, 0, n)
    q = torch.randn((B, H, 1, D), device=dev, generator=gen).to(torch.bfloat16)
    out, lse = server.decode(q, 0)
    out, lse, nz1 = out.clone(), lse.clone(), server.nnz.clone()
    codes, qn = server.hasher.query(q.reshape(BH, D))
    res = torch.zeros((BH, M), dtype=torch.int32, device=dev)
    nz = torch.zeros((BH,), dtype=torch.int32, device=dev)
    server.lsh_retriever.batch_retrieve(0, codes, res, nz)
    assert torch.equal(nz, nz1) and int(nz.min()) > 500
    o_ref = torch.zeros((BH, D), dtype=torch.bfloat16, device=dev)
    mve = torch.zeros((2, BH), dtype=torch.float32, device=dev)
    aw = server.attn_server.attention_wrapper
    aw(0, K, L, o_ref, mve, q.reshape(BH, D), qn, res, nz)
    assert np.allclose(out.reshape(BH, D).float().cpu().numpy(), o_ref.float().cpu().numpy(), rtol=2 ** -6, atol=2e-3)
    assert np.allclose(lse.reshape(-1).cpu().numpy(), mve[1].cpu().numpy(), atol=2e-3)
    # (2) permutation of the list
    res2 = res.clone()
    for h in range(BH):
        k_ = int(nz[h])
        perm = torch.randperm(k_, device=dev, generator=gen)
        res2[h, :k_] = res[h, :k_][perm]
    o2 = torch.zeros_like(o_ref)
    mve2 = torch.zeros_like(mve)
    aw(0, K, L, o2, mve2, q.reshape(BH, D), qn, res2, nz)
    assert np.allclose(o2.float().cpu().numpy(), o_ref.float().cpu().numpy(), rtol=2 ** -6, atol=2e-3)
    assert np.allclose(mve2[1].cpu().numpy(), mve[1].cpu().numpy(), atol=2e-3)
    # (3) V -> 2 V (exact in bf16): out -> 2 out bit for bit in f32 before rounding, so bf16(out) doubles exactly
    server.attn_server.fill(0, 0, (kc[:n].transpose(0, 1) - server.avg_k[0][0]).contiguous(),
                            (vc[:n].transpose(0, 1) * 2).contiguous(),
                            (kc[:n].transpose(0, 1) - server.avg_k[0][0]).norm(p=2, dim=-1).float())
    o3 = torch.zeros_like(o_ref)
    mve3 = torch.zeros_like(mve)
    aw(0, K, L, o3, mve3, q.reshape(BH, D), qn, res, nz)
    assert torch.equal(o3.float(), o_ref.float() * 2)
    assert torch.equal(mve3, mve)


def test_fused_decode_long_lists_spill_through_hbm(mp):
    """K = 1 selects almost every token: the id list of a head (~n entries) is far longer than the
    fused kernel's LDS stage, so every cluster member takes the spill path (list through HBM)."""
    B, H, Hkv, n, M, D, K, L = 1, 2, 1, 40000, 40960, 128, 1, 40
    server, _ = _fused_server(mp, B, H, Hkv, n, M, D, K, L, 99)
    BH = B * H
    q = torch.randn((B, H, 1, D), device="cuda", generator=torch.Generator(device="cuda").manual_seed(3)).to(torch.bfloat16)
    out, lse = server.decode(q, 0)
    nz1 = server.nnz.clone()
    assert int(nz1.min()) > 4096 * 8
    codes, qn = server.hasher.query(q.reshape(BH, D))
    res = torch.zeros((BH, M), dtype=torch.int32, device="cuda")
    nz = torch.zeros((BH,), dtype=torch.int32, device="cuda")
    server.lsh_retriever.batch_retrieve(0, codes, res, nz)
    assert torch.equal(nz, nz1)
    o_ref = torch.zeros((BH, D), dtype=torch.bfloat16, device="cuda")
    mve = torch.zeros((2, BH), dtype=torch.float32, device="cuda")
    server.attn_server.attention_wrapper(0, K, L, o_ref, mve, q.reshape(BH, D), qn, res, nz)
    assert np.allclose(out.reshape(BH, D).float().cpu().numpy(), o_ref.float().cpu().numpy(), rtol=2 ** -6, atol=2e-3)
    assert np.allclose(lse.reshape(-1).cpu().numpy(), mve[1].cpu().numpy(), atol=2e-3)


def test_clear_and_refill(mp):
    """clear() x2 (lsh.cc:293-306, sparse_attention.cc:586-598): after a clear every bucket is empty
    (nnz == 0, out == 0, LSE == -inf), get_mask is all zero, the stores read back as zeros, and a
    refill with other data behaves like a fresh object."""
    g = cases.load_golden("b2_k8_l60")
    seed, B, H, Hkv, n, M, D, K, L = (int(x) for x in g["meta"])
    keys, kns, vals, W, qb = cases.case_inputs(seed, B, H, Hkv, n, D, K, L)
    server = mp.LSHSparseAttnServer(1, H, Hkv, D, K=K, L=L, batch_size=B, num_sink_tokens=0,
                                    num_local_tokens=0, max_length=M, dense_layers=(), hash_func=bf16_t(W, "cuda"))

    def load(kk, vv, nn):
        for b in range(B):
            server.hash_code_buffer = server.hasher.keys(bf16_t(kk[b], "cuda"))
            server.build_table(0, b, n)
            server.attn_server.fill(0, b, bf16_t(kk[b], "cuda"), bf16_t(vv[b], "cuda"), torch.from_numpy(nn[b]).cuda())

    load(keys, vals, kns)
    q = bf16_t(qb, "cuda").view(B, H, 1, D)
    server.decode(q, 0)
    assert np.array_equal(server.nnz.cpu().numpy(), g["nnz"])
    assert np.array_equal(np.stack([np.bincount(m.astype(np.int64), minlength=3) for m in
                                    server.lsh_retriever.get_mask().numpy().reshape(B * H, M)]), g["mask_hist"])
    server.clear()
    out, lse = server.decode(q, 0)
    torch.cuda.synchronize()
    assert int(server.nnz.abs().sum()) == 0 and not out.float().abs().sum().item()
    assert torch.isinf(lse).all() and (lse < 0).all()
    assert not server.lsh_retriever.get_mask().any()
    assert not server.attn_server.get_key_cache(0).float().abs().sum().item()
    assert not server.attn_server.get_key_norm(0).abs().sum().item()
    # refill with DIFFERENT data: results equal a fresh object's
    keys2, kns2, vals2, _, _ = cases.case_inputs(seed + 99, B, H, Hkv, n, D, K, L)
    load(keys2, vals2, kns2)
    out2, lse2 = server.decode(q, 0)
    fresh = mp.LSHSparseAttnServer(1, H, Hkv, D, K=K, L=L, batch_size=B, num_sink_tokens=0,
                                   num_local_tokens=0, max_length=M, dense_layers=(), hash_func=bf16_t(W, "cuda"))
    server, keep = fresh, server
    load(keys2, vals2, kns2)
    out3, lse3 = fresh.decode(q, 0)
    torch.cuda.synchronize()
    assert torch.equal(keep.nnz, fresh.nnz) and int(fresh.nnz.sum()) > 0
    assert torch.equal(out2, out3) and torch.equal(lse2, lse3)


def test_server_fill_centres_keys(mp):
    """LSHSparseAttnServer.fill (models/attnserver.py:112-175): sink/local split, centring, norms."""
    H, Hkv, D, K, L, seq = 8, 2, 128, 8, 20, 600
    server = mp.LSHSparseAttnServer(1, H, Hkv, D, K=K, L=L, batch_size=1, max_length=1024,
                                    dense_layers=())
    gen = torch.Generator().manual_seed(5)
    kc = (torch.randn((seq, Hkv, D), generator=gen) + 0.5).to(torch.bfloat16).cuda()
    vc = torch.randn((seq, Hkv, D), generator=gen).to(torch.bfloat16).cuda()
    server.fill(0, 0, kc, vc, seq)
    server.build_table(0, 0, seq)
    n = seq - 68
    # the oracle's exactly-summed statement of attnserver.py:136-146 (pinned to torch by tests/golden/fill_centre.npz)
    e_avg, e_keys, e_vals, e_kn = oracle.centre_keys(kc.cpu(), vc.cpu(), seq, 4, 64)
    assert np.array_equal(bits_of(server.avg_k[0][0, :, 0]), e_avg)
    assert np.array_equal(bits_of(server.attn_server.get_key_cache(0)[0, :, :n]), e_keys)
    assert np.array_equal(bits_of(server.attn_server.get_value_cache(0)[0, :, :n]), e_vals)
    assert np.array_equal(server.attn_server.get_key_norm(0)[0, :, :n].cpu().numpy(), e_kn)
    # tables hold every offloaded token exactly once per (kv head, table)
    bounds, table = server.lsh_retriever.get_tables(0)
    assert torch.equal(table[0, 0, :n].sort().values.cpu(), torch.arange(n, dtype=torch.int32))
    assert int((bounds[..., -1] - bounds[..., 0]).sum()) == Hkv * L * n
    q = torch.randn((1, H, 1, D), generator=gen).to(torch.bfloat16).cuda()
    out, lse = server.decode(q, 0)
    assert torch.isfinite(out.float()).all()


def test_decode_full_window_plus_sparse_merge(mp):
    """LSHSparseAttnServer.decode_full (models/attnserver.py:261-312): static-window exact attention
    (sink + local + generated tokens, centred keys) merged by base-2 LSE with the LSH-sampled part.
    Oracle: torch f32 restatement of the window part (attnserver_dist.py:832-851) + the CPU oracle for
    the sampled part + oracle.merge_state.  FlashInfer is not in the reference tree: parity of this
    row is pinned only by that in-tree torch statement of the math."""
    import math

    H, Hkv, D, K, L, seq, B = 8, 2, 128, 8, 40, 700, 2
    G = H // Hkv
    gen = torch.Generator().manual_seed(11)
    W = synth.normal_bf16_bits(91, (D, K * L))
    server = mp.LSHSparseAttnServer(1, H, Hkv, D, K=K, L=L, batch_size=B, max_length=1024,
                                    dense_layers=(), hash_func=bf16_t(W, "cuda"), generation_buffer=8)
    kcs, vcs = [], []
    for b in range(B):
        kc = (torch.randn((seq, Hkv, D), generator=gen) * 0.5 + 0.3).to(torch.bfloat16)
        vc = torch.randn((seq, Hkv, D), generator=gen).to(torch.bfloat16)
        server.fill(0, b, kc.cuda(), vc.cuda(), seq)
        server.build_table(0, b, seq)
        kcs.append(kc)
        vcs.append(vc)
    n = seq - 68
    new_k, new_v = [], []
    for step in range(3):
        server.plan()
        q = torch.randn((B, H, 1, D), generator=gen).to(torch.bfloat16)
        # pull some queries toward a key so the sampled part carries weight
        for b in range(B):
            for h in range(0, H, 2):
                q[b, h, 0] = (0.5 * q[b, h, 0].float() + 2.0 * (kcs[b][100 + 7 * h + step, h // G].float() - 0.3)).to(torch.bfloat16)
        k_new = torch.randn((B, Hkv, 1, D), generator=gen).to(torch.bfloat16)
        v_new = torch.randn((B, Hkv, 1, D), generator=gen).to(torch.bfloat16)
        new_k.append(k_new)
        new_v.append(v_new)
        hidden = server.decode_full(q.cuda(), k_new.cuda(), v_new.cuda(), 0)
        server.window_server.check()
        torch.cuda.synchronize()
        # ---- oracle
        s_out = bits_of(server.output)                         # sampled part, checked elsewhere;
        s_lse = server.max_value_expsum[1].cpu().numpy()       # here it is an input of the merge
        w_out = np.zeros((B * H, D), np.float32)
        w_lse = np.zeros((B * H,), np.float32)
        for b in range(B):
            off = kcs[b][4:seq - 64].transpose(0, 1).contiguous()          # same ops as fill()
            avg = off.mean(dim=1, keepdim=True)                            # bf16 [Hkv,1,D]
            wk = torch.cat([kcs[b][:4], kcs[b][seq - 64:seq]] + [x[b].transpose(0, 1) for x in new_k], 0)
            wv = torch.cat([vcs[b][:4], vcs[b][seq - 64:seq]] + [x[b].transpose(0, 1) for x in new_v], 0)
            wk = (wk.transpose(0, 1) - avg).float()                        # [Hkv, W, D] centred (bf16 op, then f32)
            wv = wv.transpose(0, 1).float()
            for h in range(H):
                sc = (wk[h // G] @ q[b, h, 0].float()) / math.sqrt(D)
                p = torch.softmax(sc, 0)
                w_out[b * H + h] = (p @ wv[h // G]).numpy()
                w_lse[b * H + h] = float(torch.logsumexp(sc, 0)) / math.log(2)
        assert np.allclose(server.window_mve[1].cpu().numpy(), w_lse, atol=2e-3)
        assert np.allclose(server.window_out.float().cpu().numpy(), w_out, rtol=2 ** -7, atol=2e-3)
        mo, ms = oracle.merge_state(bits_of(server.window_out), server.window_mve[1].cpu().numpy(), s_out, s_lse)
        got = bits_of(hidden.reshape(B * H, D))
        assert np.allclose(synth.bf16_bits_to_f32(got), synth.bf16_bits_to_f32(mo), rtol=2 ** -7, atol=1e-5)
        # and end to end against an all-f32 merge of the two oracle parts
        wa = np.exp2(w_lse - np.maximum(w_lse, s_lse)); sa = np.exp2(s_lse - np.maximum(w_lse, s_lse))
        ref = (wa[:, None] * w_out + sa[:, None] * synth.bf16_bits_to_f32(s_out)) / (wa + sa)[:, None]
        assert np.allclose(synth.bf16_bits_to_f32(got), ref, rtol=1e-2, atol=3e-3)
    assert server.kv_last_page_len.tolist() == [71, 71]
    # the window is 4 + 64 + 8 rows: five more steps fill it, the sixth is refused by plan() on the host
    for _ in range(5):
        server.plan()
        server.decode_full(q.cuda(), k_new.cuda(), v_new.cuda(), 0)
    server.window_server.check()
    with pytest.raises(mp.MagicPigError):
        server.plan()
    # an append past the store's last row (bypassing plan) is reported by the device flag, never written
    server.window_server.append(0, k_new.reshape(B, Hkv, D).cuda().contiguous(), v_new.reshape(B, Hkv, D).cuda().contiguous(),
                                torch.full((B,), server.length, dtype=torch.int32, device="cuda"))
    with pytest.raises(mp.MagicPigError):
        server.window_server.check()


def test_decode_full_fused_equals_four_launch_path(mp):
    """decode_full_fused (append + ONE kernel: the static window joins the softmax of the sampled
    tokens) against decode_full (append, window attention, sparse attention, merge_state): same
    hidden states up to the bf16 rounding of the two partial outputs that the four-launch path merges,
    and against an all-f32 merge of the four-launch path's own parts.  Shapes with and without
    same-XCD clusters; a request whose sampled list is empty (K = 12, tiny context) still gets its
    window."""
    for (H, Hkv, B, K, L, seq) in [(8, 2, 2, 8, 40, 700), (32, 8, 1, 8, 40, 3000), (4, 4, 1, 12, 6, 90)]:
        D = 128
        gen = torch.Generator().manual_seed(23)
        W = synth.normal_bf16_bits(92, (D, K * L))
        mk = lambda: mp.LSHSparseAttnServer(1, H, Hkv, D, K=K, L=L, batch_size=B, max_length=4096,   # noqa: E731
                                            dense_layers=(), hash_func=bf16_t(W, "cuda"), generation_buffer=8)
        a, b_ = mk(), mk()
        for r in range(B):
            kc = (torch.randn((seq, Hkv, D), generator=gen) * 0.5 + 0.3).to(torch.bfloat16)
            vc = torch.randn((seq, Hkv, D), generator=gen).to(torch.bfloat16)
            for srv in (a, b_):
                srv.fill(0, r, kc.cuda(), vc.cuda(), seq)
                srv.build_table(0, r, seq)
        for step in range(3):
            q = (torch.randn((B, H, 1, D), generator=gen) * 2).to(torch.bfloat16).cuda()
            k_new = torch.randn((B, Hkv, 1, D), generator=gen).to(torch.bfloat16).cuda()
            v_new = torch.randn((B, Hkv, 1, D), generator=gen).to(torch.bfloat16).cuda()
            a.plan(); b_.plan()
            ref = a.decode_full(q, k_new, v_new, 0).float().cpu().numpy().reshape(B * H, D)
            got = b_.decode_full_fused(q, k_new, v_new, 0).float().cpu().numpy().reshape(B * H, D)
            assert torch.equal(a.nnz, b_.nnz)
            # all-f32 merge of the reference path's parts
            w_lse = a.window_mve[1].cpu().numpy(); s_lse = a.max_value_expsum[1].cpu().numpy()
            mx = np.maximum(w_lse, s_lse)
            wa, sa = np.exp2(w_lse - mx), np.exp2(s_lse - mx)
            f32 = (wa[:, None] * a.window_out.float().cpu().numpy() + sa[:, None] * a.output.float().cpu().numpy()) / (wa + sa)[:, None]
            assert np.allclose(got, f32, rtol=2 ** -6, atol=4e-3)
            assert np.allclose(got, ref, rtol=2 ** -6, atol=4e-3)
            lse = b_.max_value_expsum[1].cpu().numpy()
            assert np.allclose(lse, mx + np.log2(wa + sa), atol=2e-3)
        if K == 12:
            pass   # the sampled list is usually empty here: the window alone carries the output


# ------------------------------------------------------------------ device table build

@pytest.mark.parametrize("K,n", [(4, 20011), (10, 20011), (10, 8192), (10, 63), (11, 20011),
                                 (12, 40000), (13, 9000), (15, 5000)])
def test_table_build_is_a_stable_sort(mp, K, n):
    """LSH.fastfill == stable sort + LSH.fill, bit for bit (bounds and ids), for every staged
    geometry (NB <= 1024 / 2048 / 4096 / 8192) and the direct variant (K >= 14); n spans several
    LDS tiles and is not a multiple of anything."""
    Hkv, L, H, B, M = 2, 5, 4, 1, n + 3
    NB = 1 << K
    codes_np = synth.randint(700 + K, 0, NB, (Hkv, L, n)).astype(np.int16)
    codes_np[0, 0, :] = codes_np[0, 0, 0]          # one row with a single bucket
    codes_np[1, 1, :] = np.arange(n) % min(NB, 7)  # a few heavy buckets
    codes = torch.from_numpy(codes_np).cuda()
    a, b = mp.LSH(), mp.LSH()
    for x in (a, b):
        x.alloc(K, L, 1, H, Hkv, B, M)
    a.fastfill(0, 0, codes)
    sv, si = codes.sort(dim=-1, stable=True)
    b.fill(0, 0, sv.contiguous(), si.int().contiguous())
    (ba, ta), (bb, tb) = a.get_tables(0), b.get_tables(0)
    assert torch.equal(ba, bb)
    assert torch.equal(ta[..., :n], tb[..., :n])


@pytest.mark.parametrize("K", [4, 10, 11])
def test_table_build_ranking_fast_equals_exact_and_never_falls_back(mp, K):
    """Round 5: the build ranks a bucket's tokens by the order in which the LDS serves the lanes of one returning atomic (lane order
    on gfx950: observed, verified per written run, exact rebuild otherwise).  The same tables, bit for bit, as the exact ranking
    (`build_rank_exact = 1`: match-any ballots) -- with R = 8 token ranges, so the sub-bounds the sort writes are compared too --
    and NO fallback, on codes that put many lanes of one instruction on one counter (K = 4: 16 buckets; one row with a single
    bucket; a few heavy buckets)."""
    import magicpig_amd._lib as L_

    Hkv, L, H, B, n = 2, 6, 8, 1, 21000
    M = n + 40
    NB = 1 << K
    codes_np = synth.randint(900 + K, 0, NB, (Hkv, L, n)).astype(np.int16)
    codes_np[0, 0, :] = codes_np[0, 0, 0]
    codes_np[1, 1, :] = np.arange(n) % min(NB, 5)
    codes_np[1, 2, :] = (np.arange(n) // 3) % NB          # runs of three equal codes inside every 64-token group
    codes = torch.from_numpy(codes_np).cuda()
    tabs = {}
    try:
        for exact in (0, 1):
            L_.set_option("build_rank_exact", exact)
            L_.set_option("build_rank_fallbacks", 0)
            x = mp.LSH()
            x.alloc(K, L, 1, H, Hkv, B, M)
            assert x.R == 8
            x.fastfill(0, 0, codes)
            torch.cuda.synchronize()
            assert L_.get_option("build_rank_fallbacks") == 0
            bnd, tab = x.get_tables(0, raw=True)
            tabs[exact] = (bnd.clone(), tab[..., :n].clone())
            del x
    finally:
        L_.set_option("build_rank_exact", 0)
    assert torch.equal(tabs[0][0], tabs[1][0]) and torch.equal(tabs[0][1], tabs[1][1])
    sv, si = codes.sort(dim=-1, stable=True)
    assert torch.equal(tabs[0][1], si.int())             # the stable order: ids ascend inside every bucket


@pytest.mark.parametrize("H,Hkv,B", [(32, 8, 1), (8, 2, 2), (32, 8, 8)])
def test_token_range_sub_bounds_and_unstable_fill(mp, H, Hkv, B):
    """The tables of a decode cluster: every bucket's ids ascend and its R + 1 sub-bounds cut it at the
    token-range borders (R = workgroups per head, 8 / 8 / 1 for these head counts).  LSH.fill on codes
    sorted by an UNSTABLE sort (ids shuffled inside every bucket, models/attnserver.py:187) must give the
    same tables as the device counting sort: it detects the order and re-sorts on device."""
    K, L, n, M = 6, 7, 5000, 5120
    NB = 1 << K
    codes_np = synth.randint(800 + H, 0, NB, (Hkv, L, n)).astype(np.int16)
    codes_np[0, 0, :] = 3                                   # one row with a single bucket
    codes = torch.from_numpy(codes_np).cuda()
    a, b = mp.LSH(), mp.LSH()
    for x in (a, b):
        x.alloc(K, L, 1, H, Hkv, B, M)
    R, rl = a.R, a.range_len
    assert R == (1 if B * H >= 256 else 8) and rl % 32 == 0 and rl * R >= M
    a.fastfill(0, B - 1, codes)
    # unstable order: sort by (code, random key)
    rnd = torch.from_numpy(synth.randint(9, 0, 1 << 30, (Hkv, L, n))).cuda()
    order = torch.argsort(codes.long() * (1 << 31) + rnd, dim=-1)
    b.fill(0, B - 1, torch.gather(codes, -1, order).contiguous(), order.int().contiguous())
    (ba, ta), (bb, tb) = a.get_tables(0), b.get_tables(0)
    assert torch.equal(ba, bb)
    if R > 1:
        assert torch.equal(ta[..., :n], tb[..., :n])        # re-sorted on device: ascending ids, like the build
    else:                                                   # one workgroup per head needs no order: kept as given
        assert torch.equal(tb[(B - 1) * Hkv:, :, :n].cpu(), order.int().cpu())
    ba, ta = ba.cpu().numpy()[(B - 1) * Hkv:], ta.cpu().numpy()[(B - 1) * Hkv:]
    assert ba.shape == (Hkv, L, NB, R + 1)
    assert int((ba[..., -1] - ba[..., 0]).sum()) == Hkv * L * n
    for (g_, l) in [(0, 0), (Hkv - 1, L - 1), (0, 1)]:
        for bk in range(NB):
            cuts = ba[g_, l, bk]
            assert np.all(np.diff(cuts) >= 0)
            ids = ta[g_, l, cuts[0]:cuts[-1]]
            assert np.all(np.diff(ids) > 0) and np.all(codes_np[g_, l, ids] == bk)
            for r in range(R):
                piece = ta[g_, l, cuts[r]:cuts[r + 1]]
                assert np.all((piece >= r * rl) & (piece < (r + 1) * rl))


def test_unstable_fill_needs_a_permutation_and_says_so(mp):
    """with R > 1 a bucket whose ids do not ascend sends LSH.fill down the re-sort path, which needs the
    ids of a row to be a permutation of [0, n).  An id in [n, max_length) -- accepted by the sorted path -- is reported
    there with a message of its own instead of being dropped silently; fill_offload checks its device and lengths
    before it allocates."""
    import magicpig_amd._lib as L_

    K, L, H, Hkv, n, M = 6, 5, 8, 2, 3000, 3200
    codes = torch.from_numpy(synth.randint(77, 0, 1 << K, (Hkv, L, n)).astype(np.int16)).cuda()
    lsh = mp.LSH()
    lsh.alloc(K, L, 1, H, Hkv, 1, M)
    assert lsh.R == 8
    rnd = torch.from_numpy(synth.randint(10, 0, 1 << 30, (Hkv, L, n))).cuda()
    order = torch.argsort(codes.long() * (1 << 31) + rnd, dim=-1)              # unstable: ids shuffled inside buckets
    sc, ids = torch.gather(codes, -1, order).contiguous(), order.int().contiguous()
    lsh.fill(0, 0, sc, ids)                                                     # fine: re-sorted on device
    bad = ids.clone()
    bad[0, 0, 5] = n + 7                                                        # < max_length, but names no token
    with pytest.raises(L_.MagicPigError) as e:
        lsh.fill(0, 0, sc, bad)
    assert e.value.code == 6 and "not a permutation" in str(e.value)
    lsh.fill(0, 0, sc, ids)                                                     # the handle stays usable
    srv = mp.SparseAttentionServer()
    srv.alloc(1, H, Hkv, 128, 1, M)
    kc = torch.zeros((100, Hkv, 128), dtype=torch.bfloat16, device="cuda")
    with pytest.raises(ValueError):
        srv.fill_offload(0, 0, kc, kc, 60, 4, 64)                               # nothing to offload
    with pytest.raises(ValueError):
        srv.fill_offload(0, 0, kc.cpu(), kc.cpu(), 100, 4, 64)                  # not on the store's device


def test_table_build_rejects_codes_out_of_range(mp):
    K, Hkv, L, n = 6, 1, 3, 500
    codes = torch.from_numpy(synth.randint(5, 0, 1 << K, (Hkv, L, n)).astype(np.int16)).cuda()
    codes[0, 1, 17] = 1 << K
    lsh = mp.LSH()
    lsh.alloc(K, L, 1, 2, Hkv, 1, n)
    with pytest.raises(Exception):
        lsh.fastfill(0, 0, codes)


# ------------------------------------------------------------------ BASELINE cfg-1 shape

def test_cfg1_shaped_retrieve_sha(mp):
    """B=1, H=32, Hkv=8, n=97932, M=98304, K10 L150: SHA-256 of nnz + ascending selected ids
    equals the compiled reference's (tests/golden/cfg1_retrieve_sha.npz)."""
    g = cases.load_golden("cfg1_retrieve_sha")
    seed, K, L, H, Hkv, B, n, M = (int(x) for x in g["meta"])
    NB = 1 << K
    codes = torch.from_numpy(synth.randint(seed, 0, NB, (Hkv, L, n)).astype(np.int16)).cuda()
    lsh = mp.LSH()
    lsh.alloc(K, L, 1, H, Hkv, B, M)
    sc, si = codes.sort(dim=-1)            # unstable device sort, as models/attnserver.py:187
    lsh.fill(0, 0, sc.contiguous(), si.int().contiguous())
    q = torch.from_numpy(synth.randint(seed + 1, 0, NB, (B * H, L)).astype(np.int32)).cuda()
    results = torch.zeros((B * H, M), dtype=torch.int32, device="cuda")
    nnz = torch.zeros((B * H,), dtype=torch.int32, device="cuda")
    lsh.batch_retrieve(0, q, results, nnz)
    nz = nnz.cpu().numpy()
    assert np.array_equal(nz, g["nnz"])
    res = results.cpu().numpy()
    hsh = hashlib.sha256()
    hsh.update(nz.tobytes())
    for h in range(B * H):
        hsh.update(res[h, :nz[h]].tobytes())          # already ascending
    assert np.array_equal(np.frombuffer(hsh.digest(), np.uint8), g["sha256"])
    # same through the device counting-sort build
    lsh2 = mp.LSH()
    lsh2.alloc(K, L, 1, H, Hkv, B, M)
    lsh2.fastfill(0, 0, codes)
    results2 = torch.zeros_like(results)
    nnz2 = torch.zeros_like(nnz)
    lsh2.batch_retrieve(0, q, results2, nnz2)
    assert torch.equal(nnz, nnz2)
    for h in range(B * H):
        assert torch.equal(results[h, :nz[h]], results2[h, :nz[h]])


@pytest.mark.parametrize("n,shift", [(1, 0), (7, 1), (8, 0), (9, 3), (1023, 1), (4099, 5), (8193, 7)])
def test_table_build_histogram_on_unaligned_and_odd_rows(mp, n, shift):
    """Round 6 (EXPERIMENTS.md R6-8): the row histogram of the device table build reads eight codes per 16-byte load; a row
    starts wherever (kv head, table) x n puts it -- any 2-byte boundary -- so the codes in front of the first 16-byte boundary
    and behind the last whole vector are counted one by one.  Odd lengths, a code buffer that itself starts off a 16-byte
    boundary, rows shorter than a vector: the tables must be those of torch.sort + fill."""
    K, L, H, Hkv, B, M = 6, 5, 4, 2, 1, 8448
    gen = torch.Generator().manual_seed(1000 + n)
    codes_cpu = torch.randint(0, 1 << K, (Hkv, L, n), generator=gen, dtype=torch.int32).to(torch.int16)
    flat = torch.zeros((Hkv * L * n + 8,), dtype=torch.int16, device="cuda")
    view = flat[shift:shift + Hkv * L * n].view(Hkv, L, n)                 # contiguous, its first byte 2 * shift off the allocation
    view.copy_(codes_cpu)
    a, b = mp.LSH(), mp.LSH()
    a.alloc(K, L, 1, H, Hkv, B, M)
    b.alloc(K, L, 1, H, Hkv, B, M)
    a.fastfill(0, 0, view)
    sv, si = codes_cpu.cuda().sort(dim=-1, stable=True)
    b.fill(0, 0, sv.contiguous(), si.int().contiguous())
    ta, tb = a.get_tables(0), b.get_tables(0)
    assert torch.equal(ta[0], tb[0])                                       # bounds (bucket starts / ends, sub-bounds)
    assert torch.equal(ta[1][..., :n], tb[1][..., :n])                     # the table rows
