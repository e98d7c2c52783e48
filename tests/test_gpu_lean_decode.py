"""mp_decode_*_ex with MP_DECODE_NO_BYPRODUCTS (round 6): the one-launch decode that leaves no by-products -- no query
codes, no result rows, no logits -- and hands the selected ids to the gather through an UNORDERED on-chip list
(csrc/lsh.hip: LEAN).  The selected SET and the counts are the reference's (library/lsh/lsh.cc:266-283), the arithmetic
per token is sparse_attention.cc:164-240; only the order in which the tokens enter the online softmax is the order in
which they were found, which differs run to run.  So the bars are: counts bit-exact; outputs <= 1 bf16 ulp and the base-2
LSE within 1e-3 of the oracle (the parity tolerance of tests/test_gpu_parity.py), and the same against the by-products-on
call on the same stores.  Needs a real MI355X: `pytest -m gpu`."""
import numpy as np
import pytest
import torch

import cases
import synth
from test_gpu_parity import _fused_server, _oracle_attention, bf16_t, bits_of, mp  # noqa: F401  (mp: fixture)

pytestmark = pytest.mark.gpu


def _ulp_close(a, b):
    """bf16 outputs as f32 arrays: within one bf16 ulp of each other (rtol 2^-7 on a bf16 number + the absolute floor the
    parity tests use)."""
    return np.allclose(a, b, rtol=2 ** -7, atol=2e-4)


@pytest.mark.parametrize("name", ["cfg0", "gqa_32h", "b2_k8_l60", "cfg2_small", "cfg3_small", "cfg4_small", "g8_hkv2",
                                  "skew_small", "clustered_k10"])
def test_lean_decode_vs_reference_and_oracle(mp, name):
    """The lean launch on the reference-generated fixtures: nnz bit-exact against the reference, outputs / LSE against
    the oracle evaluated on the reference's selected sets (ascending), and at the reference's own tolerance against its
    outputs; get_mask / get_score say that their inputs were not written."""
    import magicpig_amd._lib as L_

    g = cases.load_golden(name)
    seed, B, H, Hkv, n, M, D, K, L = (int(x) for x in g["meta"])
    keys, kns, vals, W, qb = cases.case_inputs(seed, B, H, Hkv, n, D, K, L, cases.golden_data(g))
    server = mp.LSHSparseAttnServer(2, H, Hkv, D, K=K, L=L, batch_size=B, num_sink_tokens=0,
                                    num_local_tokens=0, max_length=M, dense_layers=(),
                                    hash_func=bf16_t(W, "cuda"))
    for b in range(B):
        server.hash_code_buffer = server.hasher.keys(bf16_t(keys[b], "cuda"))
        server.build_table(1, b, n)
        server.attn_server.fill(1, b, bf16_t(keys[b], "cuda"), bf16_t(vals[b], "cuda"),
                                torch.from_numpy(kns[b]).cuda())
    q = bf16_t(qb, "cuda").view(B, H, 1, D)
    server.by_products = False
    out, lse = server.decode(q, 1)
    torch.cuda.synchronize()
    assert np.array_equal(server.nnz.cpu().numpy(), g["nnz"])                     # integer work: bit-exact
    if server.lsh_retriever.R > 1:      # (one workgroup per head: the flag has no effect, the by-products are there)
        with pytest.raises(L_.MagicPigError):
            server.attn_server.get_score()
        with pytest.raises(L_.MagicPigError):
            server.lsh_retriever.get_mask()
    ref_lists = cases.split_ragged(g["results_ref_order"], g["nnz"])
    ind = np.zeros((B * H, M), np.int32)
    for h, lst in enumerate(ref_lists):
        ind[h, :len(lst)] = np.sort(lst)
    r = dict(dims=(B, H, Hkv, n, M, D, K, L), inputs=(keys, kns, vals, W, qb))
    o_out, o_mve, _ = _oracle_attention(r, ind, g["nnz"], exp_mode=2)
    a = synth.bf16_bits_to_f32(bits_of(out.reshape(B * H, D)))
    mve = server.max_value_expsum.cpu().numpy()
    live = g["nnz"] > 0
    assert _ulp_close(a, synth.bf16_bits_to_f32(o_out))                           # <= 1 bf16 ulp
    assert np.allclose(mve[1], o_mve[1], atol=1e-3)                               # base-2 LSE
    assert np.allclose(mve[0][live], o_mve[0][live], atol=1e-3)
    assert np.allclose(a, synth.bf16_bits_to_f32(g["out_bits"]), rtol=1e-2, atol=1e-2)   # the reference's outputs
    assert np.allclose(mve[1], g["mve"][1], atol=0.03)
    # the same stores through the by-products-on call: same counts, outputs within the same bar; the views work again
    o_lean, l_lean, z_lean = out.clone(), lse.clone(), server.nnz.clone()
    server.by_products = True
    out2, lse2 = server.decode(q, 1)
    assert torch.equal(server.nnz, z_lean)
    assert _ulp_close(o_lean.float().cpu().numpy(), out2.float().cpu().numpy())
    assert np.allclose(l_lean.cpu().numpy()[live.reshape(B, H)], lse2.cpu().numpy()[live.reshape(B, H)], atol=1e-3)
    probs = server.attn_server.get_score().reshape(B * H, M)
    for h in range(B * H):
        z = int(g["nnz"][h])
        assert z == 0 or float(probs[h, :z].sum()) == pytest.approx(1.0, abs=1e-3)
    assert server.lsh_retriever.get_mask().shape[-1] == M


@pytest.mark.parametrize("direct", [1, 0])
@pytest.mark.parametrize("K,L,n,M,cluster", [
    (4, 30, 6000, 6144, 8),        # 16 buckets: every piece overflows its slot (follow-up loads + the chunk pool)
    (6, 75, 6000, 6144, 8),
    (10, 150, 20000, 20480, 8),
    (6, 75, 6000, 6144, 16),
    (7, 300, 6000, 6144, 16),
    (4, 30, 6000, 6144, 32),
    (11, 300, 20000, 20480, 32),
    (6, 75, 6000, 6144, 1),        # one workgroup per head: the sub-bounds stream (cfg 2 / 3's regime)
    (4, 30, 6000, 6144, 1)])       # ... with pieces of hundreds of ids: both chunks and the pool
def test_lean_decode_equals_by_products_decode(mp, K, L, n, M, cluster, direct):
    """Every counting tier of the kernel (slot ids, follow-up loads, pooled chunks, the R = 1 stream) in its LEAN form
    against the by-products-on launch on the same stores, on changing queries: counts identical, outputs / LSE within
    the parity bar."""
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
    gen = torch.Generator(device="cuda").manual_seed(K * L + direct + cluster)
    for it in range(4):
        q = torch.randn((B, H, 1, D), device="cuda", generator=gen).to(torch.bfloat16)
        server.by_products = True
        o1, l1 = (t.clone() for t in server.decode(q, 0))
        z1 = server.nnz.clone()
        server.by_products = False
        o2, l2 = server.decode(q, 0)
        assert torch.equal(server.nnz, z1) and int(z1.sum()) > 0
        live = (z1 > 0).view(B, H).cpu().numpy()
        assert _ulp_close(o1.float().cpu().numpy(), o2.float().cpu().numpy())
        assert np.allclose(l1.cpu().numpy()[live], l2.cpu().numpy()[live], atol=1e-3)
    server.attn_server.check()


@pytest.mark.parametrize("B,H,Hkv,D,K,L,n,M", [
    (1, 4, 2, 64, 4, 1100, 600, 640),        # more tables than threads in a workgroup, head_dim 64
    (2, 8, 8, 128, 15, 12, 3000, 3001),      # widest codes, odd max_length, no GQA
    (3, 6, 2, 64, 9, 33, 2000, 2048),        # B*H not a multiple of 8: padded grid, head_dim 64
    (1, 1, 1, 128, 6, 50, 70, 64 * 3),       # one head, a list shorter than the cluster
    (8, 32, 8, 128, 10, 40, 3000, 3072),     # 256 heads: one workgroup per head
])
def test_lean_decode_unusual_shapes(mp, B, H, Hkv, D, K, L, n, M):
    server, _ = _fused_server(mp, B, H, Hkv, n, M, D, K, L, 1000 + K)
    gen = torch.Generator(device="cuda").manual_seed(K * L)
    for it in range(3):
        q = torch.randn((B, H, 1, D), device="cuda", generator=gen).to(torch.bfloat16)
        server.by_products = True
        o1, l1 = (t.clone() for t in server.decode(q, 0))
        z1 = server.nnz.clone()
        server.by_products = False
        o2, l2 = server.decode(q, 0)
        assert torch.equal(server.nnz, z1)
        live = (z1 > 0).view(B, H).cpu().numpy()
        a, b_ = o1.float().cpu().numpy(), o2.float().cpu().numpy()
        assert _ulp_close(a, b_)
        assert np.all(b_[~live] == 0) and np.all(np.isneginf(l2.cpu().numpy()[~live]))
        assert np.allclose(l1.cpu().numpy()[live], l2.cpu().numpy()[live], atol=1e-3)
    server.attn_server.check()


def test_lean_decode_long_lists_go_through_the_spill_list(mp):
    """K = 1 selects almost every token: a member's list is longer than the 4 096-entry LDS stage; what does not fit goes
    to the spill list in HBM, which the wave that draws the last ticket folds."""
    B, H, Hkv, n, M, D, K, L = 1, 2, 1, 40000, 40960, 128, 1, 40
    server, _ = _fused_server(mp, B, H, Hkv, n, M, D, K, L, 99)
    q = torch.randn((B, H, 1, D), device="cuda", generator=torch.Generator(device="cuda").manual_seed(3)).to(torch.bfloat16)
    o1, l1 = (t.clone() for t in server.decode(q, 0))
    z1 = server.nnz.clone()
    assert int(z1.min()) > 4096 * 8
    server.by_products = False
    o2, l2 = server.decode(q, 0)
    assert torch.equal(server.nnz, z1)
    assert _ulp_close(o1.float().cpu().numpy(), o2.float().cpu().numpy())
    assert np.allclose(l1.cpu().numpy(), l2.cpu().numpy(), atol=1e-3)


@pytest.mark.parametrize("B,H,Hkv", [(1, 32, 8), (1, 8, 2), (8, 32, 8)])
def test_lean_decode_under_graph_replay(mp, B, H, Hkv):
    """A captured step of four lean launches replayed on changing queries: counts equal to the eager by-products-on
    launch bit for bit, outputs within the parity bar (the order of the unordered list is not reproducible, so neither
    are the last bits), the per-launch placement and ticket checks stay quiet."""
    n, M, D, K, L = 6000, 6144, 128, 8, 75
    server, _ = _fused_server(mp, B, H, Hkv, n, M, D, K, L, 777)
    gen = torch.Generator(device="cuda").manual_seed(11)
    qs = torch.randn((6, B, H, 1, D), device="cuda", generator=gen).to(torch.bfloat16)
    q_static = qs[0].clone()
    eager = []
    for i in range(6):
        o, l = server.decode(qs[i], 0)
        eager.append((o.clone(), l.clone(), server.nnz.clone()))
    server.by_products = False
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        server.decode(q_static, 0)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(4):
            o_g, l_g = server.decode(q_static, 0)
    for rep in range(5):
        for i in range(6):
            q_static.copy_(qs[i])
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(server.nnz, eager[i][2])
            live = (eager[i][2] > 0).view(B, H).cpu().numpy()
            assert _ulp_close(o_g.float().cpu().numpy(), eager[i][0].float().cpu().numpy())
            assert np.allclose(l_g.cpu().numpy()[live], eager[i][1].cpu().numpy()[live], atol=1e-3)
    server.attn_server.check()


def test_lean_decode_with_the_static_window(mp):
    """mp_decode_layer_window_ex: the window folded into the lean launch against the by-products-on launch."""
    for (H, Hkv, B, K, L, seq) in [(8, 2, 2, 8, 40, 700), (32, 8, 1, 8, 40, 3000), (4, 4, 1, 12, 6, 90)]:
        D = 128
        gen = torch.Generator().manual_seed(23)
        W = synth.normal_bf16_bits(92, (D, K * L))
        mk = lambda: mp.LSHSparseAttnServer(1, H, Hkv, D, K=K, L=L, batch_size=B, max_length=4096,   # noqa: E731
                                            dense_layers=(), hash_func=bf16_t(W, "cuda"), generation_buffer=8)
        a, b_ = mk(), mk()
        b_.by_products = False
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
            ref = a.decode_full_fused(q, k_new, v_new, 0).float().cpu().numpy().reshape(B * H, D)
            got = b_.decode_full_fused(q, k_new, v_new, 0).float().cpu().numpy().reshape(B * H, D)
            assert torch.equal(a.nnz, b_.nnz)
            assert _ulp_close(ref, got)
            assert np.allclose(a.max_value_expsum[1].cpu().numpy(), b_.max_value_expsum[1].cpu().numpy(), atol=1e-3)


def test_lean_flag_is_checked(mp):
    """An unknown flag bit is an error, not a silently different launch."""
    import magicpig_amd._lib as L_

    server, _ = _fused_server(mp, 1, 8, 2, 600, 640, 128, 6, 20, 5)
    q = torch.randn((8, 128), device="cuda").to(torch.bfloat16)
    rc = L_.lib().mp_decode_sparse_layer_ex(server.hasher._h, server.lsh_retriever._h, server.attn_server._h, 0,
                                            L_.ptr(q), L_.ptr(server.output), L_.ptr(server.max_value_expsum), None, 2,
                                            L_.current_stream(q))
    assert rc == 1      # MP_ERR_INVALID


def test_cfg1_shaped_lean_decode(mp):
    """BASELINE cfg 1 at full size (B = 1, H = 32, Hkv = 8, n = 97 932, K10 L150), one layer: the lean launch against
    hash -> batch_retrieve -> attention_wrapper on the same stores (counts bit for bit, outputs up to summation order),
    and against the by-products-on launch; V -> 2 V doubles the outputs within the same bar."""
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
    for it in range(3):
        q = torch.randn((B, H, 1, D), device=dev, generator=gen).to(torch.bfloat16)
        server.by_products = True
        o1, l1 = (t.clone() for t in server.decode(q, 0))
        z1 = server.nnz.clone()
        server.by_products = False
        o2, l2 = (t.clone() for t in server.decode(q, 0))
        assert torch.equal(server.nnz, z1) and int(z1.min()) > 500
        assert _ulp_close(o1.float().cpu().numpy(), o2.float().cpu().numpy())
        assert np.allclose(l1.cpu().numpy(), l2.cpu().numpy(), atol=1e-3)
        codes, qn = server.hasher.query(q.reshape(BH, D))
        res = torch.zeros((BH, M), dtype=torch.int32, device=dev)
        nz = torch.zeros((BH,), dtype=torch.int32, device=dev)
        server.lsh_retriever.batch_retrieve(0, codes, res, nz)
        assert torch.equal(nz, z1)
        o_ref = torch.zeros((BH, D), dtype=torch.bfloat16, device=dev)
        mve = torch.zeros((2, BH), dtype=torch.float32, device=dev)
        server.attn_server.attention_wrapper(0, K, L, o_ref, mve, q.reshape(BH, D), qn, res, nz)
        assert np.allclose(o2.reshape(BH, D).float().cpu().numpy(), o_ref.float().cpu().numpy(), rtol=2 ** -6, atol=2e-3)
        assert np.allclose(l2.reshape(-1).cpu().numpy(), mve[1].cpu().numpy(), atol=2e-3)
    server.attn_server.check()
