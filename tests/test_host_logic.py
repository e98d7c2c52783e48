"""CPU-only tests of the host side of the boundary: argument checking of the mirror classes and the
error conventions of the C ABI (status codes + mp_last_error) that do not need a device."""
import ctypes as C

import pytest
import torch

import magicpig_amd
from magicpig_amd import _lib as L


def test_status_codes_without_device():
    lib = L.lib()
    h = C.c_void_p()
    assert lib.mp_lsh_create(C.byref(h)) == 0
    # not allocated -> MP_ERR_STATE, text available, no crash (the reference would dereference null)
    q = torch.zeros((4, 8), dtype=torch.int32)
    rc = lib.mp_lsh_batch_retrieve(h, 0, L.ptr(q), L.ptr(q), L.ptr(q), L.MEM_HOST, None)
    assert rc == 2 and b"not allocated" in lib.mp_last_error()
    assert lib.mp_lsh_clear(h, None) == 2
    # bad arguments are rejected before any HIP call
    assert lib.mp_lsh_alloc(h, 0, 8, 1, 4, 2, 1, 128) == 1          # K < 1
    assert lib.mp_lsh_alloc(h, 16, 8, 1, 4, 2, 1, 128) == 1         # K > 15 (int16 codes)
    assert lib.mp_lsh_alloc(h, 4, 8, 1, 6, 4, 1, 128) == 1          # H % Hkv != 0
    assert lib.mp_lsh_alloc(h, 4, 8, 1, 4, 2, 1, (1 << 22) + 1) == 1
    assert lib.mp_lsh_alloc(h, 4, 8, 1, 4, 2, 1, 2_000_000) == 5    # bitmaps exceed the 160 KiB LDS
    assert lib.mp_lsh_destroy(h) == 0
    a = C.c_void_p()
    assert lib.mp_attn_create(C.byref(a)) == 0
    assert lib.mp_attn_alloc(a, 1, 4, 2, 96, 1, 128) == 5           # head_dim must be 64 or 128
    assert b"head_dim" in lib.mp_last_error()
    assert lib.mp_attn_clear(a, None) == 2
    # ONE bound on max_length: 32-bit row offsets of the gather, max_length x 4 head_dim <= 2^32 -- rejected
    # before any allocation; a dense / window store may be longer than an LSH handle's 2^22
    assert lib.mp_attn_alloc(a, 1, 4, 2, 128, 1, (1 << 23) + 1) == 1 and b"32-bit row offsets" in lib.mp_last_error()
    assert lib.mp_attn_alloc(a, 1, 4, 2, 64, 1, (1 << 24) + 1) == 1
    assert lib.mp_attn_destroy(a) == 0
    s = C.c_void_p()
    assert lib.mp_simhash_create(C.byref(s)) == 0
    w = torch.zeros((48, 80), dtype=torch.bfloat16)
    assert lib.mp_simhash_set_planes(s, 48, 10, 8, L.ptr(w), L.MEM_HOST, None) == 5   # head_dim 48
    assert lib.mp_simhash_query(s, L.ptr(w), 1, L.ptr(w), None, L.MEM_HOST, None) == 2  # planes not set
    assert lib.mp_simhash_destroy(s) == 0
    assert lib.mp_merge_state(None, None, None, None, 1, 1, None, None, None) == 1


def test_mirror_classes_check_tensors():
    """The reference casts raw data_ptr(); the mirror classes check dtype / shape /
    contiguity / device side before anything reaches the library."""
    lsh = magicpig_amd.LSH()
    lsh.K, lsh.L, lsh.H, lsh.Hkv, lsh.B, lsh.M, lsh.NB = 4, 8, 4, 2, 1, 128, 16   # as after alloc()
    good_q = torch.zeros((4, 8), dtype=torch.int32)
    res = torch.zeros((4, 128), dtype=torch.int32)
    nnz = torch.zeros((4,), dtype=torch.int32)
    with pytest.raises(TypeError):
        lsh.batch_retrieve(0, good_q.long(), res, nnz)                # wrong dtype
    with pytest.raises(ValueError):
        lsh.batch_retrieve(0, good_q[:, :4], res, nnz)                # wrong shape
    with pytest.raises(ValueError):
        lsh.batch_retrieve(0, good_q.t().contiguous().t(), res, nnz)  # not contiguous
    with pytest.raises(TypeError):
        lsh.fill(0, 0, torch.zeros((2, 8, 16), dtype=torch.int32), torch.zeros((2, 8, 16), dtype=torch.int32))
    # no memo of "already validated" tensors (round 4's ArgCache skipped the checks for a tensor OBJECT it had
    # seen): the same object after an in-place metadata change, or after the handle was re-dimensioned, is checked again
    same = torch.zeros((4, 8), dtype=torch.int32)
    L.expect(same, torch.int32, (4, 8), "query")
    same.resize_(2, 8)
    with pytest.raises(ValueError):
        lsh.batch_retrieve(0, same, res, nnz)
    same.resize_(8, 4).t_()
    with pytest.raises(ValueError):
        lsh.batch_retrieve(0, same, res, nnz)                         # right shape, transposed in place
    lsh.M = 256                                                       # "re-alloc" with another max_length
    with pytest.raises(ValueError):
        lsh.batch_retrieve(0, good_q, res, nnz)
    lsh.M = 128
    srv = magicpig_amd.SparseAttentionServer()
    srv.H, srv.Hkv, srv.D, srv.B, srv.M = 4, 2, 128, 1, 128
    out = torch.zeros((4, 128), dtype=torch.bfloat16)
    mve = torch.zeros((2, 4), dtype=torch.float32)
    q = torch.zeros((4, 128), dtype=torch.bfloat16)
    qn = torch.zeros((4,), dtype=torch.float32)
    with pytest.raises(TypeError):
        srv.attention_wrapper(0, 10, 150, out.float(), mve, q, qn, res, nnz)
    with pytest.raises(ValueError):
        srv.attention_wrapper(0, 10, 150, out, mve[:1], q, qn, res, nnz)
    with pytest.raises(TypeError):
        srv.fill(0, 0, torch.zeros((2, 16, 128)), torch.zeros((2, 16, 128)), torch.zeros((2, 16)))


def test_bench_configs_match_baseline_json():
    """bench.py's workloads are the BASELINE.json configs (model shape constants of SURVEY.md 8)."""
    import bench

    c1 = bench.CONFIGS["cfg1"]
    assert (c1["B"], c1["P"], c1["K"], c1["L"], c1["H"], c1["Hkv"], c1["D"]) == (1, 98000, 10, 150, 32, 8, 128)
    assert len([i for i in range(c1["layers"]) if i not in c1["dense"]]) == 30
    c2 = bench.CONFIGS["cfg2"]
    assert (c2["B"], c2["P"], c2["K"], c2["L"]) == (8, 32768, 10, 170)
    c4 = bench.CONFIGS["cfg4"]
    assert (c4["H"], c4["Hkv"], c4["P"], c4["K"], c4["L"]) == (8, 1, 131072, 11, 300)
    assert len([i for i in range(c4["layers"]) if i not in c4["dense"]]) == 75
    for c in bench.CONFIGS.values():
        assert c["M"] >= c["P"] - 68


def test_harness_rope_and_rmsnorm_match_the_reference_fixture():
    """SURVEY f-3: the two pieces of model plumbing the decode-step harness restates (magicpig_amd/decode_harness.py:
    rope_tables / apply_rotary_pos_emb / rms_norm) against tests/golden/llama_ops.npz -- the torch-CPU execution of
    models/llama.py:114-126 and models/utils.py:29-45 verbatim, and flashinfer.rmsnorm's published definition.  Bit
    for bit on the CPU (same torch, same arithmetic); the tables may differ from the fixture in the last bf16 bit only
    where torch.outer and the reference's batched matmul round a product differently (none observed)."""
    import numpy as np
    import torch

    import cases
    import synth
    from magicpig_amd import decode_harness as dh

    g = cases.load_golden("llama_ops")
    c = cases.LLAMA_OPS
    x, w, q, k = cases.llama_ops_inputs(c)
    bits = lambda t: t.contiguous().view(torch.int16).numpy().view(np.uint16)       # noqa: E731
    cos, sin = dh.rope_tables(c["D"], c["max_len"], c["theta"], "cpu")
    rows = list(c["positions"])
    assert np.array_equal(bits(cos[rows]), g["cos_rows"]) and np.array_equal(bits(sin[rows]), g["sin_rows"])
    qt, kt = synth.to_torch_bf16(q), synth.to_torch_bf16(k)
    for p in c["positions"]:
        pos = torch.full((c["B"], 1), p, dtype=torch.long)
        assert np.array_equal(bits(dh.apply_rotary_pos_emb(qt, cos, sin, pos)), g[f"q_rope_{p}"])
        assert np.array_equal(bits(dh.apply_rotary_pos_emb(kt, cos, sin, pos)), g[f"k_rope_{p}"])
    got = dh.rms_norm(synth.to_torch_bf16(x), synth.to_torch_bf16(w), c["eps"])
    assert np.array_equal(bits(got), g["rmsnorm"])
