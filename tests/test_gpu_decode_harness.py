"""the decode-step harness (magicpig_amd/decode_harness.py) on a tiny Llama
shape.  With K = 1, L = 64 practically every offloaded token collides in >= 2 tables and its
importance weight is ~1, so the LSH-sampled decode must reproduce exact dense attention: the harness'
logits are compared with a torch restatement of the same decoder that attends densely in f32."""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def test_decoder_matches_dense_attention_when_everything_is_sampled():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from magicpig_amd import decode_harness as dh

    shape = dh.LlamaShape(hidden_size=512, num_hidden_layers=3, num_attention_heads=4, num_key_value_heads=2,
                          intermediate_size=1024, vocab_size=1000)
    B, P, steps = 2, 600, 3
    dec = dh.SyntheticLlamaDecoder(shape, K=1, L=64, batch_size=B, max_length=1024, generation_buffer=8,
                                   dense_layers=(0,), seed=3)
    H, Hkv, D = 4, 2, 128
    G = H // Hkv
    gen = torch.Generator(device="cuda").manual_seed(5)
    kvs = [[(torch.randn((P, Hkv, D), device="cuda", generator=gen).to(torch.bfloat16),
             torch.randn((P, Hkv, D), device="cuda", generator=gen).to(torch.bfloat16)) for _ in range(3)]
           for _ in range(B)]
    for b in range(B):
        dec.prefill_synthetic(b, P, kv=lambda layer, b=b: kvs[b][layer])
    # torch reference state: per request, per layer growing K/V [T, Hkv, D]
    ref_k = [[kvs[b][l][0].float() for l in range(3)] for b in range(B)]
    ref_v = [[kvs[b][l][1].float() for l in range(3)] for b in range(B)]
    ids = torch.randint(0, 1000, (B, steps), device="cuda", generator=gen)
    for t in range(steps):
        pos = torch.full((B, 1), P + t, device="cuda", dtype=torch.long)
        logits = dec.inference(ids[:, t:t + 1], pos)
        # ---- reference decoder: same bf16 projections, dense f32 attention; RMSNorm and RoPE restated HERE (f32
        #      arithmetic, the rotation written out per pair), not taken from the module under test -- the harness'
        #      own versions are pinned bit for bit by tests/golden/llama_ops.npz (test_host_logic.py)
        def ref_norm(x, w):
            xf = x.float()
            return (xf / torch.sqrt((xf * xf).mean(-1, keepdim=True) + shape.rms_norm_eps) * w.float()).to(x.dtype)

        def ref_rope(x, p):                     # x [B, heads, 1, D]; pairs (i, i + D/2) rotate by p * theta^(-2i/D)
            half = D // 2
            ang = p[:, :, None].float() * (shape.rope_theta ** (-torch.arange(half, device="cuda").float() * 2 / D))
            cs, sn = ang.cos().to(torch.bfloat16)[:, None], ang.sin().to(torch.bfloat16)[:, None]   # [B, 1, 1, half]
            a, b_ = x[..., :half], x[..., half:]
            return torch.cat((a * cs - b_ * sn, b_ * cs + a * sn), dim=-1)

        hs = F.embedding(ids[:, t:t + 1], dec.embed_tokens)
        for l in range(3):
            W = dec.layers[l]
            x = ref_norm(hs, W["ln1"])
            q = F.linear(x, W["wq"]).view(B, 1, H, D).transpose(1, 2)
            k = F.linear(x, W["wk"]).view(B, 1, Hkv, D).transpose(1, 2)
            v = F.linear(x, W["wv"]).view(B, 1, Hkv, D).transpose(1, 2)
            k = ref_rope(k, pos)
            q = ref_rope(q, pos)
            attn = torch.zeros((B, 1, H * D), device="cuda", dtype=torch.bfloat16)
            for b in range(B):
                ref_k[b][l] = torch.cat([ref_k[b][l], k[b].transpose(0, 1).float()], 0)
                ref_v[b][l] = torch.cat([ref_v[b][l], v[b].transpose(0, 1).float()], 0)
                for h in range(H):
                    s = (ref_k[b][l][:, h // G] @ q[b, h, 0].float()) / math.sqrt(D)
                    attn[b, 0, h * D:(h + 1) * D] = (torch.softmax(s, 0) @ ref_v[b][l][:, h // G]).to(torch.bfloat16)
            hmid = hs + F.linear(attn, W["wo"])
            y = ref_norm(hmid, W["ln2"])
            hs = hmid + F.linear(F.silu(F.linear(y, W["gate"])) * F.linear(y, W["up"]), W["down"])
        ref_logits = F.linear(ref_norm(hs, dec.norm_weight), dec.lm_head).float()
        err = (logits - ref_logits).abs().max().item()
        scale = ref_logits.abs().max().item()
        assert err < 0.03 * scale + 0.03, (t, err, scale)
    # nearly every offloaded token was sampled in the sparse layers
    assert float(dec.attention_server.nnz.float().mean()) > 0.97 * (P - 68)
    dec.attention_server.window_server.check()
    dec.dense_server.check()


def test_decoder_in_the_sampling_regime_against_the_pinned_oracle_parts():
    """One decode step of the harness with K = 8, L = 48 on 4 096 offloaded tokens (~1.6 % of them sampled, importance
    weights over several orders of magnitude): every sparse layer's attention, as the harness really called it (query
    after RoPE, this step's k / v, the stores as the prefill left them), is recomputed from the PINNED oracle parts --
    oracle SimHash + retrieve + importance-corrected attention over the offloaded tokens, exact attention over the
    static window, one softmax over their union (tests/test_gpu_configs._oracle_union) -- and must agree within 1 bf16
    ulp; the selected counts must agree exactly."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import numpy as np

    import synth
    from magicpig_amd import decode_harness as dh
    from test_gpu_configs import _oracle_union

    shape = dh.LlamaShape(hidden_size=512, num_hidden_layers=3, num_attention_heads=4, num_key_value_heads=2,
                          intermediate_size=1024, vocab_size=1000)
    B, H, Hkv, D, K, L = 2, 4, 2, 128, 8, 48
    P, M = 4096 + 68, 4224
    n = P - 68
    dec = dh.SyntheticLlamaDecoder(shape, K=K, L=L, batch_size=B, max_length=M, generation_buffer=8,
                                   dense_layers=(0,), seed=11)
    for b in range(B):
        dec.prefill_synthetic(b, P, seed=20 + b)
    srv = dec.attention_server
    calls = []
    inner = srv.decode_full_fused

    def recording(q, k, v, li):
        out = inner(q, k, v, li)
        calls.append((li, q.reshape(B * H, D).clone(), out.reshape(B * H, D).clone(), srv.nnz.clone(),
                      srv.max_value_expsum[1].clone()))
        return out

    srv.decode_full_fused = recording
    gen = torch.Generator(device="cuda").manual_seed(9)
    ids = torch.randint(0, 1000, (B, 1), device="cuda", generator=gen)
    pos = torch.full((B, 1), P, device="cuda", dtype=torch.long)
    logits = dec.inference(ids, pos)
    torch.cuda.synchronize()
    srv.window_server.check()
    assert torch.isfinite(logits).all() and len(calls) == 2
    bits = lambda t: t.detach().cpu().contiguous().view(torch.int16).numpy().view(np.uint16)       # noqa: E731
    W = bits(srv.hash_func)
    rows = 68 + 1                                                  # sink + local + this step's token
    c = dict(B=B, H=H, Hkv=Hkv, D=D, K=K, L=L, n=n, M=M)
    for li, q, out, nnz, lse in calls:
        kc, vc = srv.attn_server.get_key_cache(li), srv.attn_server.get_value_cache(li)
        kn = srv.attn_server.get_key_norm(li)
        keys = [bits(kc[b, :, :n]) for b in range(B)]
        vals = [bits(vc[b, :, :n]) for b in range(B)]
        kns = [kn[b, :, :n].cpu().numpy().copy() for b in range(B)]
        wkc, wvc = srv.window_server.get_key_cache(li), srv.window_server.get_value_cache(li)
        wk = [bits(wkc[b, :, :rows]) for b in range(B)]
        wv = [bits(wvc[b, :, :rows]) for b in range(B)]
        ref, ref_lse = _oracle_union(c, keys, kns, vals, W, bits(q), wk, wv, nnz.cpu().numpy())
        assert 20 < float(nnz.float().mean()) < 0.05 * n           # the sampling regime, not "everything"
        assert np.allclose(out.float().cpu().numpy(), ref, rtol=2 ** -7, atol=2e-4), li
        assert np.allclose(lse.cpu().numpy(), ref_lse, atol=1e-3), li


def test_llama_8b_shaped_step_is_finite_and_both_window_forms_agree():
    """The timed configuration of bench.py --end-to-end, checked: one decode step of the Llama-3.1-8B-shaped decoder
    (32 layers, hidden 4096, 32 / 8 heads, K10 L150, synthetic weights, 4 096 offloaded tokens) gives finite logits,
    every sparse layer samples, and on every sparse layer's REAL inputs (query after RoPE, this step's k / v) the two
    forms of the layer -- decode_full_fused (window folded into the decode launch) and decode_full (append, window
    attention, hot path, merge_state) -- agree up to the bf16 rounding of the four-launch form's partial outputs.
    (Layer by layer, not on the final logits: LSH sampling is discontinuous in the query, so a one-ulp difference in
    one layer's output selects other tokens in the next and the two 30-layer trajectories part ways.)"""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import numpy as np

    from magicpig_amd import decode_harness as dh

    P = 4096 + 68
    dec = dh.SyntheticLlamaDecoder(dh.LLAMA_3_1_8B, K=10, L=150, batch_size=1, max_length=4224, generation_buffer=8,
                                   dense_layers=(0, 16), seed=2)
    dec.prefill_synthetic(0, P, seed=4)
    gen = torch.Generator(device="cuda").manual_seed(6)
    ids = torch.randint(0, dec.shape.vocab_size, (1, 1), device="cuda", generator=gen)
    pos = torch.full((1, 1), P, device="cuda", dtype=torch.long)
    srv = dec.attention_server
    calls = []
    inner = srv.decode_full_fused

    def recording(q, k, v, li):
        out = inner(q, k, v, li)
        calls.append((li, q.clone(), k.clone(), v.clone(), out.clone(), srv.nnz.clone()))
        return out

    srv.decode_full_fused = recording
    logits = dec.inference(ids, pos)
    torch.cuda.synchronize()
    srv.window_server.check()
    dec.dense_server.check()
    assert torch.isfinite(logits).all() and logits.shape == (1, 1, dec.shape.vocab_size)
    assert len(calls) == 30 and min(float(c[5].float().mean()) for c in calls) > 10
    # the four-launch form on the same inputs: its append rewrites the row the fused step appended (same row, same
    # values), the window lengths still stand
    for li, q, k, v, out, nnz in calls:
        h4 = srv.decode_full(q, k, v, li)
        assert torch.equal(srv.nnz, nnz), li
        assert np.allclose(h4.float().cpu().numpy(), out.float().cpu().numpy(), rtol=2 ** -6, atol=4e-3), li
    srv.window_server.check()


def test_decode_benchmark_loop_runs():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from magicpig_amd import decode_harness as dh

    shape = dh.LlamaShape(hidden_size=512, num_hidden_layers=4, num_attention_heads=4, num_key_value_heads=2,
                          intermediate_size=1024, vocab_size=1000)
    dec = dh.SyntheticLlamaDecoder(shape, K=8, L=40, batch_size=2, max_length=2048, generation_buffer=16,
                                   dense_layers=(0,), seed=1)
    ms, tps = dh.run_decode_benchmark(dec, prompt_len=1500, warmup=4, steps=8)
    assert ms > 0 and tps > 0


def test_tensor_parallel_decoder_equals_the_unsharded_one():
    """The TP variant of the harness (evaluations/RULER/pred/llama_dist.py:195-220 + attnserver_dist.py:252-254: heads
    and MLP columns sharded, the partial o_proj / down_proj outputs all-reduced) with TWO ranks emulated by two threads
    on one device -- each rank its own decoder (own LSH tables, KV store, window: nothing of the attention path crosses
    ranks) and an all_reduce that meets the other rank's partial at a barrier -- against the TP = 1 decoder holding
    the same model (weights and prompt KV are drawn per block).  K = 1, L = 64: practically every token is sampled, so
    the comparison is not at the mercy of a token flipping in or out of a sample.  Logits agree to bf16 summation
    order (two bf16 partials added vs one bf16 sum)."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import threading

    from magicpig_amd import decode_harness as dh

    shape = dh.LlamaShape(hidden_size=512, num_hidden_layers=3, num_attention_heads=4, num_key_value_heads=2,
                          intermediate_size=1024, vocab_size=1000)
    B, P, steps, W = 2, 600, 3, 2
    common = dict(K=1, L=64, batch_size=B, max_length=1024, generation_buffer=8, dense_layers=(0,), seed=11)
    one = dh.SyntheticLlamaDecoder(shape, tp_blocks=W, **common)
    for b in range(B):
        one.prefill_synthetic(b, P, seed=b)
    gen = torch.Generator(device="cuda").manual_seed(6)
    ids = torch.randint(0, 1000, (B, steps), device="cuda", generator=gen)
    want = []
    for t in range(steps):
        pos = torch.full((B, 1), P + t, device="cuda", dtype=torch.long)
        want.append(one.inference(ids[:, t:t + 1], pos).clone())
    torch.cuda.synchronize()

    # ---- two ranks, two threads, one device: partial sums meet at a barrier (all launches go to the default stream,
    # so "both partials enqueued, then the sum" is also their order on the device)
    meet = threading.Barrier(W)
    slot = [None] * W
    errors = []

    def make_reduce(rank):
        def all_reduce(t):
            slot[rank] = t
            meet.wait()
            total = slot[0] + slot[1]
            meet.wait()                     # both have read both partials
            t.copy_(total)
        return all_reduce

    got = [[None] * steps for _ in range(W)]

    def run(rank):
        try:
            torch.cuda.set_device(0)
            dec = dh.SyntheticLlamaDecoder(shape, tp_rank=rank, tp_world=W, all_reduce=make_reduce(rank), **common)
            assert dec.shape.num_attention_heads == 2 and dec.shape.num_key_value_heads == 1
            for b in range(B):
                dec.prefill_synthetic(b, P, seed=b)
            for t in range(steps):
                pos = torch.full((B, 1), P + t, device="cuda", dtype=torch.long)
                got[rank][t] = dec.inference(ids[:, t:t + 1], pos).clone()
            torch.cuda.synchronize()
        except BaseException as e:          # noqa: BLE001 -- a failing rank must not leave the other at the barrier
            errors.append((rank, repr(e)))
            meet.abort()

    threads = [threading.Thread(target=run, args=(r,)) for r in range(W)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=600)
    assert not errors, errors
    for t in range(steps):
        assert torch.equal(got[0][t], got[1][t])                      # replicated lm_head on identical hidden states
        a, b = got[0][t].float(), want[t].float()
        assert torch.isfinite(a).all()
        assert (a - b).abs().max() <= 0.06 * b.abs().max(), (t, float((a - b).abs().max()), float(b.abs().max()))
        assert torch.equal(a.argmax(-1), b.argmax(-1)) or (a - b).abs().max() <= 0.02 * b.abs().max()
    # the slices really are the unsharded model's: rank r's rows of wq / columns of wo are block r of the TP = 1 matrices
    r1 = dh.SyntheticLlamaDecoder(shape, tp_rank=1, tp_world=W, **{**common, "dense_layers": (0,)})
    D, H = 128, 4
    assert torch.equal(r1.layers[1]["wq"], one.layers[1]["wq"][H // W * D:])
    assert torch.equal(r1.layers[1]["wo"], one.layers[1]["wo"][:, H // W * D:])
    assert torch.equal(r1.layers[2]["down"], one.layers[2]["down"][:, 512:])
    assert torch.equal(r1.embed_tokens, one.embed_tokens)
