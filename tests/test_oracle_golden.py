"""Pins the CPU oracle (oracle/mp_oracle.c) against golden vectors produced by the reference
itself (tests/golden/make_golden.py: compiled library/lsh + library/sparse_attention, and
the torch-CPU restatement of models/attnserver.py:264-270 / :159-168).  CPU only."""
import hashlib

import numpy as np
import pytest

import cases
import oracle
import synth

QHASH = ["qhash_r1_k10_l150", "qhash_r32_k10_l150", "qhash_r64_k11_l300", "qhash_r40_k8_l50",
         "qhash_r256_k10_l170"]
PIPE = ["lsh_small", "cfg0", "gqa_32h", "b2_k8_l60",
        "cfg2_small", "cfg3_small",   # BASELINE cfg 2 / cfg 3 head counts (B = 8, 256 query heads; L = 170 / 150)
        "cfg4_small", "g8_hkv2",      # cfg 4's per-GPU geometry (H = 8, Hkv = 1, K11 L300) and G = 8 with Hkv > 1
        "skew_small", "clustered_k10"]   # the non-isotropic workloads of SURVEY.md 8(d) (tests/synth.py)


@pytest.mark.parametrize("name", QHASH)
def test_query_simhash_bit_exact(name):
    g = cases.load_golden(name)
    seed, R, D, K, L = (int(x) for x in g["meta"])
    qb = synth.normal_bf16_bits(seed, (R, D))
    W = synth.normal_bf16_bits(seed + 7, (D, K * L))
    codes, qn = oracle.simhash_query(qb, W, K, L)
    assert codes.dtype == np.int32 and codes.shape == (R, L)
    assert np.array_equal(codes, g["qcodes"])          # bit-exact hash codes
    qf = synth.bf16_bits_to_f32(qb).astype(np.float64)
    assert np.allclose(qn, np.sqrt((qf * qf).sum(-1)), rtol=1e-6)


def _run_oracle_pipeline(g, exp_mode):
    seed, B, H, Hkv, n, M, D, K, L = (int(x) for x in g["meta"])
    keys, kns, vals, W, qb = cases.case_inputs(seed, B, H, Hkv, n, D, K, L, cases.golden_data(g))
    qcodes, qn = oracle.simhash_query(qb, W, K, L)
    kcodes = np.stack([oracle.simhash_keys(keys[b], W, K, L) for b in range(B)])
    lsh = oracle.LSH()
    lsh.alloc(K, L, 1, H, Hkv, B, M)
    for b in range(B):
        sc, si = cases.stable_sort_codes(kcodes[b])
        lsh.fill(0, b, sc, si)
    results = np.zeros((B * H, M), np.int32)
    nnz = np.zeros((B * H,), np.int32)
    lsh.batch_retrieve(0, qcodes, results, nnz)
    srv = oracle.SparseAttentionServer(exp_mode=exp_mode)
    srv.alloc(1, H, Hkv, D, B, M)
    for b in range(B):
        srv.fill(0, b, keys[b], vals[b], kns[b])
    out = np.zeros((B * H, D), np.uint16)
    mve = np.zeros((2, B * H), np.float32)
    srv.attention_wrapper(0, K, L, out, mve, qb, qn, results, nnz)
    return dict(qcodes=qcodes, kcodes=kcodes, nnz=nnz, results=results, mask=lsh.get_mask(),
                out=out, mve=mve, probs=srv.get_score().reshape(B * H, M), qn=qn,
                dims=(B, H, Hkv, n, M, D, K, L))


@pytest.mark.parametrize("name", PIPE)
def test_pipeline_matches_reference(name):
    g = cases.load_golden(name)
    r = _run_oracle_pipeline(g, exp_mode=1)
    B, H, Hkv, n, M, D, K, L = r["dims"]
    # ---- integer work: bit-exact
    assert np.array_equal(r["qcodes"], g["qcodes"])
    assert np.array_equal(r["kcodes"][0, 0, 0], g["kcodes_head0_table0"])
    assert np.array_equal(
        np.frombuffer(hashlib.sha256(r["kcodes"].tobytes()).digest(), np.uint8), g["kcodes_sha"])
    cases.check_sign_ties(g, r["kcodes"], K)
    assert np.array_equal(r["nnz"], g["nnz"])
    ref_lists = cases.split_ragged(g["results_ref_order"], g["nnz"])
    for h in range(B * H):
        # same tables, same scan => even the reference's second-hit ORDER is reproduced
        assert np.array_equal(r["results"][h, :r["nnz"][h]], ref_lists[h])
        hist = np.bincount(r["mask"].reshape(B * H, M)[h].astype(np.int64), minlength=3)
        assert np.array_equal(hist, g["mask_hist"][h])
    # independent dense statement of the same math (library/lsh/test.py:41-47)
    cnt = cases.dense_mask_counts(r["kcodes"].reshape(B * Hkv, L, n), r["qcodes"], H // Hkv)
    assert np.array_equal((cnt > 1).sum(-1), g["nnz"])
    # ---- floating point, reference-polynomial exp emulated => tight agreement
    assert np.allclose(r["qn"], g["qnorm"], rtol=1e-6)
    ref_probs = cases.split_ragged(g["probs"], g["nnz"])
    for h in range(B * H):
        z = r["nnz"][h]
        assert np.allclose(r["probs"][h, :z], ref_probs[h], rtol=2e-4, atol=1e-7), h
    assert np.allclose(r["mve"], g["mve"], rtol=1e-5, atol=1e-4)
    o = synth.bf16_bits_to_f32(r["out"])
    og = synth.bf16_bits_to_f32(g["out_bits"])
    assert np.allclose(o, og, rtol=2 ** -7, atol=1e-5)   # <= 1 bf16 ulp
    assert (r["out"] == g["out_bits"]).mean() > 0.97


@pytest.mark.parametrize("name", PIPE)
def test_pipeline_exact_exp_within_reference_tolerance(name):
    """The oracle proper (exact expf) against the reference at the reference's own test
    tolerance rtol=atol=1e-2 (library/sparse_attention/test_sparse.py:87-92)."""
    g = cases.load_golden(name)
    r = _run_oracle_pipeline(g, exp_mode=0)
    B, H = r["dims"][:2]
    ref_probs = cases.split_ragged(g["probs"], g["nnz"])
    for h in range(B * H):
        z = r["nnz"][h]
        assert np.allclose(r["probs"][h, :z], ref_probs[h], rtol=2e-2, atol=1e-4)
        assert abs(r["probs"][h, :z].sum() - 1) <= 1e-4
    assert np.allclose(synth.bf16_bits_to_f32(r["out"]), synth.bf16_bits_to_f32(g["out_bits"]),
                       rtol=1e-2, atol=1e-2)
    # base-2 LSE: the polynomial exp is <= 1.7 % low => LSE differs by <= log2(1.017)
    assert np.allclose(r["mve"][1], g["mve"][1], atol=0.03)


def test_lsh_edge_cases():
    g = cases.load_golden("lsh_edge")
    seed, K, L, H, Hkv, B, n, M = (int(x) for x in g["meta"])
    NB = 1 << K
    codes = synth.randint(seed, 0, NB, (B, Hkv, L, n)).astype(np.int16)
    lsh = oracle.LSH()
    lsh.alloc(K, L, 2, H, Hkv, B, M)
    for b in range(B):
        sc, si = cases.stable_sort_codes(codes[b])
        lsh.fill(1, b, sc, si)
    q = g["q"]
    for rep, (nk, sk) in enumerate((("nnz0", "sorted0"), ("nnz1", "sorted1"))):
        qq = q if rep == 0 else ((q + 3) % NB).astype(np.int32)
        results = np.zeros((B * H, M), np.int32)
        nnz = np.zeros((B * H,), np.int32)
        lsh.batch_retrieve(1, np.ascontiguousarray(qq), results, nnz)   # second call: mask reset
        assert np.array_equal(nnz, g[nk])
        got = np.concatenate([np.sort(results[h, :nnz[h]]) for h in range(B * H)])
        assert np.array_equal(got, g[sk])
    assert g["nnz0"][3] == 0                      # the empty-result head
    first = cases.split_ragged(g["sorted0"], g["nnz0"])
    assert 5 in first[1]                          # the head that copies token 5's codes


def test_attention_edge_cases():
    g = cases.load_golden("attn_edge")
    seed, K, L, H, Hkv, B, n, M, D = (int(x) for x in g["meta"])
    keys, kns, vals, W, qb = cases.case_inputs(seed, B, H, Hkv, n, D, K, L)
    nnz = g["nnz"].astype(np.int32)
    ind = np.zeros((B * H, M), np.int32)
    for h, z in enumerate(nnz):
        perm = np.argsort(synth.u64(seed + 50 + h, n), kind="stable").astype(np.int32)
        ind[h, :z] = perm[:z]
    for exp_mode, rt, at in ((1, 2e-4, 1e-7), (0, 2e-2, 1e-4)):
        srv = oracle.SparseAttentionServer(exp_mode=exp_mode)
        srv.alloc(1, H, Hkv, D, B, M)
        srv.fill(0, 0, keys[0], vals[0], kns[0])
        out = np.zeros((B * H, D), np.uint16)
        mve = np.zeros((2, B * H), np.float32)
        srv.attention_wrapper(0, K, L, out, mve, qb, g["qnorm"], ind, nnz)
        probs = srv.get_score().reshape(B * H, M)
        ref = cases.split_ragged(g["probs"], nnz)
        for h, z in enumerate(nnz):
            assert np.allclose(probs[h, :z], ref[h], rtol=rt, atol=at), (exp_mode, h)
        # nnz == 0: out = 0, LSE = -inf (row 0 of max_value_expsum is stale garbage in the
        # reference for an empty head: std::max_element of an empty range)
        assert nnz[0] == 0 and not out[0].any() and mve[1, 0] == -np.inf and g["mve"][1, 0] == -np.inf
        assert np.allclose(mve[1, 1:], g["mve"][1, 1:], atol=0.03 if exp_mode == 0 else 1e-4)
        assert np.allclose(synth.bf16_bits_to_f32(out), synth.bf16_bits_to_f32(g["out_bits"]),
                           rtol=1e-2, atol=1e-2)


def test_cfg1_shaped_retrieve_sha():
    """BASELINE cfg-1-shaped layer (H=32, Hkv=8, n=97932, M=98304, K10 L150): SHA-256 of
    nnz + sorted selected ids equals the compiled reference's."""
    g = cases.load_golden("cfg1_retrieve_sha")
    seed, K, L, H, Hkv, B, n, M = (int(x) for x in g["meta"])
    NB = 1 << K
    codes = synth.randint(seed, 0, NB, (Hkv, L, n)).astype(np.int16)
    sc, si = cases.stable_sort_codes(codes)
    lsh = oracle.LSH()
    lsh.alloc(K, L, 1, H, Hkv, B, M)
    lsh.fill(0, 0, sc, si)
    q = synth.randint(seed + 1, 0, NB, (B * H, L)).astype(np.int32)
    results = np.zeros((B * H, M), np.int32)
    nnz = np.zeros((B * H,), np.int32)
    lsh.batch_retrieve(0, q, results, nnz)
    assert np.array_equal(nnz, g["nnz"])
    hsh = hashlib.sha256()
    hsh.update(nnz.tobytes())
    for h in range(B * H):
        hsh.update(np.sort(results[h, :nnz[h]]).tobytes())
    assert np.array_equal(np.frombuffer(hsh.digest(), np.uint8), g["sha256"])


# ------------------------------------------------------------------ a-15 / f-4: full_attention pinned

@pytest.mark.parametrize("case", cases.FULL_DENSE_CASES, ids=[c[0] for c in cases.FULL_DENSE_CASES])
def test_full_attention_matches_reference(case):
    """mpo_full_attention against the compiled reference's full_attention (sparse_attention.cc:988-1037),
    group sizes 1 / 4 / 8 as library/sparse_attention/test_dense.py:8-14, list lengths around the 16- and
    64-row block edges and 0.  quirks = 3 reproduces the reference (polynomial exp, 16-slot softmax tail):
    tight; quirks = 0 is the definition: equal to the reference within its own test tolerance wherever
    the tail quirk is silent (nnz % 16 == 0)."""
    g = cases.load_golden("full_dense")
    seed, D = (int(x) for x in g["meta"])
    tag, B, H, Hkv, n, M, nnz_list = case
    keys, vals, q = cases.full_dense_inputs(cases.full_dense_seed(seed, tag, H), B, H, Hkv, n, D)
    kn = np.zeros((Hkv, n), np.float32)
    BH = B * H
    for z in nnz_list:
        ref_out = synth.bf16_bits_to_f32(g[f"{tag}_z{z}_out"])
        ref_mve = g[f"{tag}_z{z}_mve"]
        ref_probs = g[f"{tag}_z{z}_probs"]
        nnz = np.full((BH,), z, np.int32)
        for quirks in (3, 0):
            srv = oracle.SparseAttentionServer()          # fresh server: the score buffer is zero
            srv.alloc(1, H, Hkv, D, B, M)
            for b in range(B):
                srv.fill(0, b, keys[b], vals[b], kn)
            out = np.zeros((BH, D), np.uint16)
            mve = np.zeros((2, BH), np.float32)
            srv.full_attention(0, out, mve, q, nnz, quirks=quirks)
            probs = srv.get_score().reshape(BH, M)
            o = synth.bf16_bits_to_f32(out)
            if z == 0:        # empty list: out = 0, LSE = -inf (row 0 of max_value_expsum is -inf too)
                assert not out.any() and not g[f"{tag}_z{z}_out"].any()
                assert np.all(np.isneginf(mve[1])) and np.all(np.isneginf(ref_mve[1]))
                continue
            if quirks == 3:
                z16 = min((z + 15) & ~15, M)
                assert np.allclose(probs[:, :z16], ref_probs[:, :z16], rtol=2e-4, atol=1e-7), (tag, z)
                assert np.allclose(mve, ref_mve, rtol=1e-5, atol=1e-4), (tag, z)
                assert np.allclose(o, ref_out, rtol=2 ** -7, atol=1e-5), (tag, z)          # <= 1 bf16 ulp
                assert (out == g[f"{tag}_z{z}_out"]).mean() > 0.97
            elif z % 16 == 0:
                assert np.allclose(probs[:, :z], ref_probs[:, :z], rtol=2e-2, atol=1e-4), (tag, z)
                assert np.allclose(o, ref_out, rtol=1e-2, atol=1e-2), (tag, z)             # test_dense.py:66
                assert np.allclose(mve[1], ref_mve[1], atol=0.03), (tag, z)                # poly exp <= 1.7 % low
                assert np.all(np.abs(probs[:, :z].sum(-1) - 1) <= 1e-4)
            else:
                # the reference's softmax also counted zero-score slots behind the list: its sum is
                # larger by (z16 - z) * exp(0 - m), nothing else differs
                assert np.all(np.abs(probs[:, :z].sum(-1) - 1) <= 1e-4)
                assert np.all(mve[1] <= ref_mve[1] + 0.03)


# ------------------------------------------------------------------ a-13 / f-2: window + LSE merge pinned

def _window_merge_oracle_parts(c, g):
    seed, B, H, Hkv, D, K, L, n, M = (c[k] for k in ("seed", "B", "H", "Hkv", "D", "K", "L", "n", "M"))
    keys, kns, vals, W, qb, wk, wv = cases.window_merge_inputs(c)
    BH = B * H
    qcodes, qn = oracle.simhash_query(qb, W, K, L)
    kcodes = np.stack([oracle.simhash_keys(keys[b], W, K, L) for b in range(B)])
    lsh = oracle.LSH()
    lsh.alloc(K, L, 1, H, Hkv, B, M)
    for b in range(B):
        sc, si = cases.stable_sort_codes(kcodes[b])
        lsh.fill(0, b, sc, si)
    results = np.zeros((BH, M), np.int32)
    nnz = np.zeros((BH,), np.int32)
    lsh.batch_retrieve(0, qcodes, results, nnz)
    return keys, kns, vals, W, qb, wk, wv, qn, results, nnz


def test_window_merge_matches_torch_statement():
    """The three oracle functions of the sparse-layer decode against the committed torch-CPU statement of
    the reference's call site (tests/golden/make_golden.py: run_window_merge): sampled half
    (attnserver_dist.py:813-851), exact window attention with a base-2 LSE, flashinfer.merge_state."""
    c = cases.WINDOW_MERGE
    g = cases.load_golden("window_merge")
    B, H, Hkv, D, K, L, n, M, win_M = (c[k] for k in ("B", "H", "Hkv", "D", "K", "L", "n", "M", "win_M"))
    BH = B * H
    keys, kns, vals, W, qb, wk, wv, qn, results, nnz = _window_merge_oracle_parts(c, g)
    assert np.array_equal(nnz, g["nnz"])                 # the torch collision mask selects the same tokens
    # -- sampled half: the literal f32 formula (exp_mode 0) within the reference's tolerance
    srv = oracle.SparseAttentionServer(exp_mode=0)
    srv.alloc(1, H, Hkv, D, B, M)
    for b in range(B):
        srv.fill(0, b, keys[b], vals[b], kns[b])
    sp_out = np.zeros((BH, D), np.uint16)
    sp_mve = np.zeros((2, BH), np.float32)
    srv.attention_wrapper(0, K, L, sp_out, sp_mve, qb, qn, results, nnz)
    assert np.allclose(sp_mve[1], g["sparse_lse"], atol=5e-3)
    assert np.allclose(synth.bf16_bits_to_f32(sp_out), synth.bf16_bits_to_f32(g["sparse_out"]), rtol=1e-2, atol=1e-2)
    # -- window: full_attention over the first win_rows[b] rows of a second store
    wsrv = oracle.SparseAttentionServer()
    wsrv.alloc(1, H, Hkv, D, B, win_M)
    for b in range(B):
        wsrv.fill(0, b, wk[b], wv[b], np.zeros((Hkv, wk[b].shape[1]), np.float32))
    w_out = np.zeros((BH, D), np.uint16)
    w_mve = np.zeros((2, BH), np.float32)
    wnnz = np.repeat(np.array(c["win_rows"], np.int32), H)
    wsrv.full_attention(0, w_out, w_mve, synth.bf16_bits_to_f32(qb), wnnz)
    assert np.allclose(w_mve[1], g["window_lse"], atol=1e-4)
    assert np.allclose(synth.bf16_bits_to_f32(w_out), synth.bf16_bits_to_f32(g["window_out"]), rtol=2 ** -7, atol=1e-5)
    # -- merge_state on the statement's own partials: bit-level agreement with its f64 evaluation
    v, s = oracle.merge_state(g["window_out"], g["window_lse"], g["sparse_out"], g["sparse_lse"])
    assert np.allclose(s, g["merged_lse"], atol=1e-5)
    assert np.allclose(synth.bf16_bits_to_f32(v), synth.bf16_bits_to_f32(g["merged_out"]), rtol=2 ** -7, atol=1e-6)
    assert (v == g["merged_out"]).mean() > 0.99
    # -- the defining property: merged == one softmax over the union (up to the bf16 rounding of the partials)
    assert np.allclose(s, g["joint_lse"], atol=1e-4)
    assert np.allclose(synth.bf16_bits_to_f32(v), g["joint_out"], rtol=2 ** -6, atol=2e-3)
    # -- and end to end from the oracle's own partials
    v2, s2 = oracle.merge_state(w_out, w_mve[1], sp_out, sp_mve[1])
    assert np.allclose(s2, g["joint_lse"], atol=5e-3)
    assert np.allclose(synth.bf16_bits_to_f32(v2), g["joint_out"], rtol=1e-2, atol=1e-2)


# ------------------------------------------------------------------ f-1: key centring / norms of the prefill

def test_fill_centre_matches_torch():
    """oracle.centre_keys (exact sums; the definition the HIP fill kernels implement) against the torch-CPU
    execution of models/attnserver.py:133-146 (tests/golden/make_golden.py: run_fill_centre): equal bit for bit,
    except at the listed summation-order ties (torch sums in f32), where the fixture holds both values."""
    c = cases.FILL_CENTRE
    g = cases.load_golden("fill_centre")
    k, v = cases.fill_centre_inputs(c)
    avg, keys, vals, kn = oracle.centre_keys(k, v, c["seq_len"], c["num_sink"], c["num_local"])
    n = c["seq_len"] - c["num_sink"] - c["num_local"]
    assert keys.shape == (c["Hkv"], n, c["D"]) and kn.shape == (c["Hkv"], n)
    t_avg = g["avg_k"].copy()
    assert len(g["avg_ties"]) <= 2 and len(g["kn_ties"]) <= 4            # rounding-boundary cases only
    for (i, j), e in zip(g["avg_ties"], g["avg_exact_at_ties"]):
        assert abs(int(t_avg[i, j]) - int(e)) == 1                       # one bf16 ulp apart
        t_avg[i, j] = e
    assert np.array_equal(avg, t_avg)
    t_kn = g["kn"].copy()
    for (i, j), e in zip(g["kn_ties"], g["kn_exact_at_ties"]):
        t_kn[i, j] = e
    if len(g["avg_ties"]) == 0:      # same avg_k => same centred keys
        assert np.array_equal(np.frombuffer(hashlib.sha256(keys.tobytes()).digest(), np.uint8), g["key_sha"])
        assert np.array_equal(keys[0, 0], g["key_head0_tok0"])
        assert np.array_equal(kn, t_kn)
    assert np.array_equal(vals, np.ascontiguousarray(v[c["num_sink"]:c["seq_len"] - c["num_local"]].transpose(1, 0, 2)))
