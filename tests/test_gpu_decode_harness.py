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
        # ---- reference decoder: same bf16 projections, dense f32 attention
        hs = F.embedding(ids[:, t:t + 1], dec.embed_tokens)
        for l in range(3):
            W = dec.layers[l]
            x = dh.rms_norm(hs, W["ln1"], shape.rms_norm_eps)
            q = F.linear(x, W["wq"]).view(B, 1, H, D).transpose(1, 2)
            k = F.linear(x, W["wk"]).view(B, 1, Hkv, D).transpose(1, 2)
            v = F.linear(x, W["wv"]).view(B, 1, Hkv, D).transpose(1, 2)
            k = dh.apply_rotary_pos_emb(k, dec.cos_cache, dec.sin_cache, pos)
            q = dh.apply_rotary_pos_emb(q, dec.cos_cache, dec.sin_cache, pos)
            attn = torch.zeros((B, 1, H * D), device="cuda", dtype=torch.bfloat16)
            for b in range(B):
                ref_k[b][l] = torch.cat([ref_k[b][l], k[b].transpose(0, 1).float()], 0)
                ref_v[b][l] = torch.cat([ref_v[b][l], v[b].transpose(0, 1).float()], 0)
                for h in range(H):
                    s = (ref_k[b][l][:, h // G] @ q[b, h, 0].float()) / math.sqrt(D)
                    attn[b, 0, h * D:(h + 1) * D] = (torch.softmax(s, 0) @ ref_v[b][l][:, h // G]).to(torch.bfloat16)
            hmid = hs + F.linear(attn, W["wo"])
            y = dh.rms_norm(hmid, W["ln2"], shape.rms_norm_eps)
            hs = hmid + F.linear(F.silu(F.linear(y, W["gate"])) * F.linear(y, W["up"]), W["down"])
        ref_logits = F.linear(dh.rms_norm(hs, dec.norm_weight, shape.rms_norm_eps), dec.lm_head).float()
        err = (logits - ref_logits).abs().max().item()
        scale = ref_logits.abs().max().item()
        assert err < 0.03 * scale + 0.03, (t, err, scale)
    # nearly every offloaded token was sampled in the sparse layers
    assert float(dec.attention_server.nnz.float().mean()) > 0.97 * (P - 68)
    dec.attention_server.window_server.check()
    dec.dense_server.check()


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
