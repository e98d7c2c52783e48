"""The quad MFMA query hash (round 6): at one workgroup per head the four heads of an XCD residue hash
together -- a quarter of the hyperplanes each, against the four query rows, on the matrix pipe -- and exchange sign words
through the XCD's L2.  Codes are the exact sign either way (models/attnserver.py:264-270): the launch with the option on,
on-but-nobody-publishes (every head times out and hashes alone) and off must give bit-identical codes, counts, outputs and
LSE, launch after launch and under graph replay.  Needs a real MI355X: `pytest -m gpu`."""
import pytest
import torch

from test_gpu_parity import _fused_server, mp  # noqa: F401  (mp: fixture)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,H,Hkv,K,L", [(8, 32, 8, 10, 150), (8, 32, 8, 10, 170), (4, 32, 8, 8, 61), (1, 256, 32, 11, 64)])
def test_quad_hash_equals_the_per_head_hash(mp, B, H, Hkv, K, L):
    import magicpig_amd._lib as L_

    n, M, D = 3000, 3072, 128
    L_.set_option("decode_cluster", 1)           # (B*H = 128 would be split over two workgroups per head)
    try:
        server, _ = _fused_server(mp, B, H, Hkv, n, M, D, K, L, 900 + K + L)
    finally:
        L_.set_option("decode_cluster", 0)
    assert server.lsh_retriever.R == 1
    BH = B * H
    gen = torch.Generator(device="cuda").manual_seed(K + L)
    qs = torch.randn((5, B, H, 1, D), device="cuda", generator=gen).to(torch.bfloat16)
    qs[1] *= 37.0                       # other scales of the rows: the fast / exact normalisation paths
    qs[2] *= 1e-3
    ref = []
    try:
        for mode in (0, 1, 2, 1):
            L_.set_option("decode_quad_hash", mode)
            for i in range(5):
                out, lse = server.decode(qs[i], 0)
                codes = server.lsh_retriever.get_mask()          # (recomputed from the launch's codes)
                got = (out.clone(), lse.clone(), server.nnz.clone(), codes)
                if mode == 0:
                    ref.append(got)
                else:
                    assert torch.equal(got[2], ref[i][2]), (mode, i)
                    assert torch.equal(got[3], ref[i][3]), (mode, i)
                    assert torch.equal(got[0], ref[i][0]) and torch.equal(got[1], ref[i][1]), (mode, i)
        server.attn_server.check()
        # a captured step of three launches replayed on changing queries
        L_.set_option("decode_quad_hash", 1)
        q_static = qs[0].clone()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            server.decode(q_static, 0)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            for _ in range(3):
                o_g, l_g = server.decode(q_static, 0)
        for rep in range(4):
            for i in range(5):
                q_static.copy_(qs[i])
                graph.replay()
                torch.cuda.synchronize()
                assert torch.equal(o_g, ref[i][0]) and torch.equal(l_g, ref[i][1]) and torch.equal(server.nnz, ref[i][2])
    finally:
        L_.set_option("decode_quad_hash", -1)
    assert int(torch.stack([r[2] for r in ref]).sum()) > 0
