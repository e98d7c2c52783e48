"""mp_lsh_alloc_ex (round 6: the footprint policy as API -- how much HBM a handle's accelerator structures may take and how
many token ranges a table row is cut into are the CALLER's to state, lsh.cc:44-91 allocates exactly what its arguments say),
and the hooks that make two round-5 advisories testable: the table build's exact-ranking rebuild and the empty request.
Needs a real MI355X: `pytest -m gpu`."""
import numpy as np
import pytest
import torch

from test_gpu_parity import _fused_server, bf16_t, mp  # noqa: F401  (mp: fixture)

pytestmark = pytest.mark.gpu


def test_two_handles_with_different_budgets_in_one_process(mp):
    """One process, two LSH handles of the same shape: one allowed the direct piece slots, one not, one with a forced number
    of ranges -- each keeps what IT was given (no process-wide switch involved), both decode the same layer to the same
    counts and outputs (summation order differs with the number of ranges)."""
    B, H, Hkv, D, K, L, n, M = 1, 8, 2, 128, 8, 75, 6000, 6144
    import cases
    keys, kns, vals, W, qb = cases.case_inputs(77, B, H, Hkv, n, D, K, L)
    servers = []
    for budget, ranges in ((0, 0), (1 << 30, 0), (1 << 30, 4), (None, 0)):
        s = mp.LSHSparseAttnServer(1, H, Hkv, D, K=K, L=L, batch_size=B, num_sink_tokens=0, num_local_tokens=0,
                                   max_length=M, dense_layers=(), hash_func=bf16_t(W, "cuda"),
                                   accel_budget_bytes=budget, ranges=ranges)
        for b in range(B):
            s.hash_code_buffer = s.hasher.keys(bf16_t(keys[b], "cuda"))
            s.build_table(0, b, n)
            s.attn_server.fill(0, b, bf16_t(keys[b], "cuda"), bf16_t(vals[b], "cuda"), torch.from_numpy(kns[b]).cuda())
        servers.append(s)
    f = [s.lsh_retriever.footprint() for s in servers]
    assert f[0]["slots"] == 0 and f[0]["accel_budget"] == 0 and f[0]["accel_in_use"] == 0
    assert f[1]["slots"] > 0 and f[1]["accel_budget"] == 1 << 30 and f[1]["accel_in_use"] == f[1]["slots"]
    assert servers[2].lsh_retriever.R == 4 and servers[1].lsh_retriever.R == 8
    assert f[3]["accel_budget"] == -1 and f[3]["slots"] == f[1]["slots"]          # the library's rule: the same choice here
    q = torch.randn((B, H, 1, D), device="cuda", generator=torch.Generator(device="cuda").manual_seed(5)).to(torch.bfloat16)
    ref = None
    for s in servers:
        o, l = s.decode(q, 0)
        got = (o.float().cpu().numpy().copy(), l.cpu().numpy().copy(), s.nnz.cpu().numpy().copy())
        if ref is None:
            ref = got
            assert got[2].sum() > 0
        assert np.array_equal(got[2], ref[2])
        assert np.allclose(got[0], ref[0], rtol=2 ** -6, atol=2e-3) and np.allclose(got[1], ref[1], atol=2e-3)
    # the budget also covers the host-buffer mode's HBM copy of the handed-out rows: refused at budget 0 (the staged path
    # serves the call), taken inside 1 GiB
    for s, want in ((servers[0], 0), (servers[1], B * H * M * 4 + B * H * 4)):
        codes, _ = s.hasher.query(q.reshape(B * H, D))
        res, nz = torch.zeros((B * H, M), dtype=torch.int32), torch.zeros((B * H,), dtype=torch.int32)
        s.lsh_retriever.batch_retrieve(0, codes.cpu(), res, nz)
        assert np.array_equal(nz.numpy(), ref[2])
        assert s.lsh_retriever.footprint()["host_mode_row_copy"] == want
    with pytest.raises(Exception):
        mp.LSH().alloc(K, L, 1, H, Hkv, B, M, ranges=3)


def test_table_build_rebuilds_with_the_exact_ranking_when_its_check_fails(mp):
    """The default build ranks by the order in which the LDS serves the lanes of one atomic and verifies every bucket run; a
    failed check means: rebuild the request with the exact ranking.  gfx950 never fails the check, so the hook
    `build_rank_inject` makes the next build behave as if it had: identical tables, the counter says one rebuild."""
    import magicpig_amd._lib as L_

    Hkv, L, K, n, M = 2, 20, 10, 20011, 20480
    gen = torch.Generator(device="cuda").manual_seed(1)
    codes = torch.randint(0, 1 << K, (Hkv, L, n), device="cuda", generator=gen, dtype=torch.int32).to(torch.int16)
    tabs = []
    for inject in (0, 1):
        lsh = mp.LSH()
        lsh.alloc(K, L, 1, 8, Hkv, 1, M)
        L_.set_option("build_rank_fallbacks", 0)
        L_.set_option("build_rank_inject", inject)
        lsh.fastfill(0, 0, codes)
        assert L_.get_option("build_rank_fallbacks") == inject
        assert L_.get_option("build_rank_inject") == 0
        b, t = lsh.get_tables(0)
        tabs.append((b.clone(), t[:, :, :n].clone()))
    assert torch.equal(tabs[0][0], tabs[1][0]) and torch.equal(tabs[0][1], tabs[1][1])


def test_an_empty_request_builds_empty_tables(mp):
    """n = 0 through mp_lsh_build*: every bucket empty, no kernel reads codes[-1]; a decode over it selects nothing."""
    Hkv, H, L, K, M, D = 2, 8, 20, 10, 2048, 128
    lsh = mp.LSH()
    lsh.alloc(K, L, 1, H, Hkv, 1, M)
    lsh.fastfill(0, 0, torch.zeros((Hkv, L, 0), dtype=torch.int16, device="cuda"))
    b, _ = lsh.get_tables(0)
    assert int(b.abs().sum()) == 0
    q = torch.randint(0, 1 << K, (H, L), dtype=torch.int32, device="cuda")
    res, nz = torch.zeros((H, M), dtype=torch.int32, device="cuda"), torch.ones((H,), dtype=torch.int32, device="cuda")
    lsh.batch_retrieve(0, q, res, nz)
    assert int(nz.sum()) == 0


def test_slot_width_option_is_read_at_alloc_only(mp):
    """`decode_slot_log2` can be read back, and a handle keeps the width it was allocated with when the option changes
    afterwards (round 5: the builder and the reader re-read the global: out-of-bounds slot writes)."""
    import magicpig_amd._lib as L_

    L_.set_option("decode_slot_log2", 3)
    try:
        assert L_.get_option("decode_slot_log2") == 3
        server, _ = _fused_server(mp, 1, 8, 2, 6000, 6144, 128, 6, 75, 31)
        assert server.lsh_retriever.footprint()["slot_bytes"] == 32
        L_.set_option("decode_slot_log2", 5)           # would be 128-byte slots for a NEW handle
        q = torch.randn((1, 8, 1, 128), device="cuda", generator=torch.Generator(device="cuda").manual_seed(2)).to(torch.bfloat16)
        o1, _ = server.decode(q, 0)
        o1, z1 = o1.clone(), server.nnz.clone()
        server.hash_code_buffer = None
        L_.set_option("decode_direct", 0)
        ref, _ = _fused_server(mp, 1, 8, 2, 6000, 6144, 128, 6, 75, 31)
        o2, _ = ref.decode(q, 0)
        assert torch.equal(ref.nnz, z1) and int(z1.sum()) > 0
        assert np.allclose(o1.float().cpu().numpy(), o2.float().cpu().numpy(), rtol=2 ** -6, atol=2e-3)
    finally:
        L_.set_option("decode_slot_log2", 0)
        L_.set_option("decode_direct", -1)
