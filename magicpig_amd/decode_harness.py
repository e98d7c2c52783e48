"""Decode-step harness at Llama shapes with SYNTHETIC weights.

Restates the decode side of the reference's minimal Llama runtime -- `LLM.inference` /
`layer_compute` (models/llama.py:185-220, 288-301) with the helpers of models/utils.py -- around
this repository's attention path, so that the loop and the printed metrics of examples/bench.py
(:43-59: prefill B requests, 32 warm-up + 128 timed `llm.inference` steps, ms/token and token/s)
can be reproduced without HF weights (no network here): weights are random tensors of the model's
shapes, the prompt's KV cache is synthetic.

What runs where:
  * sparse layers : LSHSparseAttnServer.decode_full_fused -- q-hash + retrieve + sampled attention
                    (the hot path, HIP), static-window attention + LSE merge (HIP);
  * dense layers  : (0, 16, ...; models/attnserver.py:235-259) exact attention over the whole
                    sequence = the dense mode of the same HIP kernel on a full-length store;
  * everything else (embedding, RMSNorm, q/k/v/o and MLP projections, RoPE, lm_head) is plain
    torch-ROCm -- model plumbing outside the north-star path, kept only so that an end-to-end
    tokens/s can be quoted next to the hot-path number.

Tensor-parallel variant (round 4; the reference's evaluations/RULER/pred/llama_dist.py:195-220 with
attnserver_dist.py:252-254): rank r of W holds the query / kv heads [r H/W, (r+1) H/W) -- its own LSH tables, KV store and
window, nothing of the attention path crosses ranks -- the matching rows of wq / wk / wv / gate / up and columns of wo /
down, and the partial outputs of o_proj and down_proj are summed over the ranks (`dist.all_reduce`: RCCL); embedding and
lm_head are replicated.  Weights and the synthetic prompt KV are drawn per BLOCK (a rank's slice, seeded by the block's
global index), so a TP = W run and a TP = 1 run built with `tp_blocks = W` hold the same model.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch
import torch.nn.functional as F

from .attnserver import LSHSparseAttnServer
from .sparse_attention import SparseAttentionServer


@dataclass
class LlamaShape:
    hidden_size: int = 4096
    num_hidden_layers: int = 32
    num_attention_heads: int = 32
    num_key_value_heads: int = 8
    intermediate_size: int = 14336
    vocab_size: int = 128256
    rms_norm_eps: float = 1e-5
    rope_theta: float = 500000.0

    @property
    def head_dim(self) -> int:
        return self.hidden_size // self.num_attention_heads


LLAMA_3_1_8B = LlamaShape()
LLAMA_3_1_70B = LlamaShape(hidden_size=8192, num_hidden_layers=80, num_attention_heads=64,
                           num_key_value_heads=8, intermediate_size=28672)


def rms_norm(x: torch.Tensor, weight: torch.Tensor, eps: float) -> torch.Tensor:
    """models/utils.py:47-56 (`flashinfer.rmsnorm`): x * rsqrt(mean(x^2) + eps) * w, evaluated in f32 and rounded
    once (pinned by tests/golden/llama_ops.npz)."""
    xf = x.float()
    return (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps) * weight.float()).to(x.dtype)


def rotate_half(x: torch.Tensor) -> torch.Tensor:
    x1, x2 = x[..., : x.shape[-1] // 2], x[..., x.shape[-1] // 2:]
    return torch.cat((-x2, x1), dim=-1)


def rope_tables(head_dim: int, max_length: int, theta: float, device, dtype=torch.bfloat16):
    """models/llama.py:114-126 (attention_scaling = 1): cos / sin caches [max_length, head_dim]."""
    inv_freq = 1.0 / (theta ** (torch.arange(0, head_dim, 2, device=device).float() / head_dim))
    pos = torch.arange(0, max_length, device=device).float()
    freqs = torch.outer(pos, inv_freq)
    emb = torch.cat((freqs, freqs), dim=-1)
    return emb.cos().to(dtype), emb.sin().to(dtype)


def apply_rotary_pos_emb(x, cos, sin, position_ids, unsqueeze_dim=1):
    """models/utils.py:36-45 (pinned by tests/golden/llama_ops.npz)."""
    c = cos[position_ids].unsqueeze(unsqueeze_dim)
    s = sin[position_ids].unsqueeze(unsqueeze_dim)
    return (x * c) + (rotate_half(x) * s)


class SyntheticLlamaDecoder:
    """`LLM` of models/llama.py:64-101 reduced to what `inference` needs, with random weights."""

    def __init__(self, shape: LlamaShape = LLAMA_3_1_8B, K: int = 10, L: int = 150, batch_size: int = 1,
                 max_length: int = 8192, generation_buffer: int = 256, dense_layers=(0, 16, 32, 48, 64),
                 device: str = "cuda:0", dtype=torch.bfloat16, seed: int = 0,
                 tp_rank: int = 0, tp_world: int = 1, tp_blocks: int | None = None, all_reduce=None):
        """tp_rank / tp_world: this process's slice of the heads and of the MLP (llama_dist.py); tp_blocks: the number of
        blocks the weights are DRAWN in (default tp_world; a TP = 1 decoder with tp_blocks = W holds the model of a TP = W
        run); all_reduce: callable summing a tensor over the ranks in place (default: torch.distributed when a process
        group exists and tp_world > 1, else nothing to do)."""
        self.tp_rank, self.tp_world = tp_rank, tp_world
        self.tp_blocks = tp_blocks if tp_blocks is not None else tp_world
        assert 0 <= tp_rank < tp_world and self.tp_blocks % tp_world == 0
        full = shape
        assert full.num_key_value_heads % self.tp_blocks == 0 and full.intermediate_size % self.tp_blocks == 0
        self.full_shape = full
        # the shape this rank computes with: its heads, its share of the MLP (hidden size and head_dim are the model's)
        shape = LlamaShape(hidden_size=full.hidden_size, num_hidden_layers=full.num_hidden_layers,
                           num_attention_heads=full.num_attention_heads // tp_world,
                           num_key_value_heads=full.num_key_value_heads // tp_world,
                           intermediate_size=full.intermediate_size // tp_world, vocab_size=full.vocab_size,
                           rms_norm_eps=full.rms_norm_eps, rope_theta=full.rope_theta)
        self._head_dim = full.head_dim
        self._all_reduce_fn = all_reduce
        self.shape, self.K, self.L = shape, K, L
        self.fused_window = True      # sparse layers: decode_full_fused (two launches); False: decode_full (four)
        self.batch_size, self.max_length = batch_size, max_length
        self.device, self.dtype = torch.device(device), dtype
        self.num_layers = shape.num_hidden_layers
        self.dense_layers = tuple(i for i in dense_layers if i < self.num_layers)
        self.sparse_layers = tuple(i for i in range(self.num_layers) if i not in self.dense_layers)
        H, Hkv, D = shape.num_attention_heads, shape.num_key_value_heads, self._head_dim
        g = torch.Generator(device=self.device).manual_seed(seed)

        def w(*dims, scale):
            return (torch.randn(dims, device=self.device, dtype=torch.float32, generator=g) * scale).to(dtype)

        # a sharded matrix is drawn block by block, every block from a generator seeded by (seed, layer, matrix, GLOBAL
        # block index): the blocks a rank owns are the same numbers whatever tp_world is
        nb = self.tp_blocks
        mine = range(tp_rank * nb // tp_world, (tp_rank + 1) * nb // tp_world)

        def wb(layer, which, rows, cols, scale, dim):
            parts = []
            for blk in mine:
                gb = torch.Generator(device=self.device).manual_seed(((seed * 1009 + layer) * 16 + which) * 4096 + blk + 1)
                parts.append((torch.randn((rows, cols), device=self.device, dtype=torch.float32, generator=gb) * scale).to(dtype))
            return torch.cat(parts, dim=dim).contiguous()

        hs, it = full.hidden_size, full.intermediate_size
        Hf, Hkvf = full.num_attention_heads, full.num_key_value_heads
        self.embed_tokens = w(full.vocab_size, hs, scale=1.0)                     # replicated (same seed on every rank)
        self.lm_head = w(full.vocab_size, hs, scale=hs ** -0.5)
        self.norm_weight = torch.ones(hs, device=self.device, dtype=dtype)
        self.layers = []
        for li in range(self.num_layers):
            self.layers.append(dict(
                wq=wb(li, 0, Hf // nb * D, hs, hs ** -0.5, 0), wk=wb(li, 1, Hkvf // nb * D, hs, hs ** -0.5, 0),
                wv=wb(li, 2, Hkvf // nb * D, hs, hs ** -0.5, 0), wo=wb(li, 3, hs, Hf // nb * D, (Hf * D) ** -0.5, 1),
                gate=wb(li, 4, it // nb, hs, hs ** -0.5, 0), up=wb(li, 5, it // nb, hs, hs ** -0.5, 0),
                down=wb(li, 6, hs, it // nb, it ** -0.5, 1),
                ln1=torch.ones(hs, device=self.device, dtype=dtype),
                ln2=torch.ones(hs, device=self.device, dtype=dtype)))
        # RoPE tables (models/llama.py:114-126)
        self.cos_cache, self.sin_cache = rope_tables(D, max_length, shape.rope_theta, self.device, dtype)
        # attention state: sparse layers -> LSH server (indexed by position in sparse_layers);
        # dense layers -> one full-length store (indexed by position in dense_layers)
        self.sparse_index = {l: i for i, l in enumerate(self.sparse_layers)}
        self.dense_index = {l: i for i, l in enumerate(self.dense_layers)}
        self.attention_server = LSHSparseAttnServer(
            max(1, len(self.sparse_layers)), H, Hkv, D, K=K, L=L, batch_size=batch_size,
            generation_buffer=generation_buffer, max_length=max_length, dense_layers=(), device=device,
            seed=seed + 7)
        if self.dense_layers:
            with torch.cuda.device(self.device):
                self.dense_server = SparseAttentionServer()
                self.dense_server.alloc(len(self.dense_layers), H, Hkv, D, batch_size, max_length)
        BH = batch_size * H
        self.dense_len = torch.zeros((batch_size,), dtype=torch.int32, device=self.device)
        self.dense_nnz = torch.zeros((BH,), dtype=torch.int32, device=self.device)
        self.dense_out = torch.zeros((BH, D), dtype=torch.bfloat16, device=self.device)
        self.dense_mve = torch.zeros((2, BH), dtype=torch.float32, device=self.device)

    # ------------------------------------------------------------------ prefill (synthetic KV)
    def prefill_synthetic(self, request_id: int, seq_len: int, seed: int = 0, kv=None) -> None:
        """Stands in for `LLM.prefill` (models/llama.py:304-325): per layer a KV cache bf16
        [seq_len, Hkv, D] (random unless `kv(layer) -> (k, v)` supplies one) is handed to the attention
        state exactly as `layer_prefill` does (:264, 282: fill + build_table)."""
        Hkv, D = self.shape.num_key_value_heads, self._head_dim
        for layer in range(self.num_layers):
            if kv is not None:
                k, v = kv(layer)
            else:       # per GLOBAL kv head: a rank's prompt KV does not depend on tp_world
                ks, vs = [], []
                for gkv in range(self.tp_rank * Hkv, (self.tp_rank + 1) * Hkv):
                    g = torch.Generator(device=self.device).manual_seed((seed * 1000 + layer) * 64 + gkv)
                    ks.append(torch.randn((seq_len, 1, D), device=self.device, generator=g).to(self.dtype))
                    vs.append(torch.randn((seq_len, 1, D), device=self.device, generator=g).to(self.dtype))
                k, v = torch.cat(ks, dim=1), torch.cat(vs, dim=1)
            if layer in self.dense_index:
                kk = k.transpose(0, 1).contiguous()
                self.dense_server.fill(self.dense_index[layer], request_id, kk, v.transpose(0, 1).contiguous(),
                                       kk.norm(p=2, dim=-1).float())
            else:
                li = self.sparse_index[layer]
                self.attention_server.fill(li, request_id, k, v, seq_len)
                self.attention_server.build_table(li, request_id, seq_len)
        self.dense_len[request_id] = seq_len

    # ------------------------------------------------------------------ decode
    def _dense_attention(self, q, k, v, layer: int) -> torch.Tensor:
        """models/attnserver.py:235-259: append, then exact attention over the whole sequence."""
        B, H, Hkv, D = self.batch_size, self.shape.num_attention_heads, self.shape.num_key_value_heads, self._head_dim
        di = self.dense_index[layer]
        self.dense_server.append(di, k.reshape(B, Hkv, D).contiguous(), v.reshape(B, Hkv, D).contiguous(),
                                 self.dense_len - 1)
        self.dense_server.full_attention(di, self.dense_out, self.dense_mve, q.reshape(B * H, D), self.dense_nnz)
        return self.dense_out.view(B, 1, H * D)

    def plan(self) -> None:
        self.attention_server.plan()
        self.dense_len += 1
        self.dense_nnz.copy_(self.dense_len.repeat_interleave(self.shape.num_attention_heads))

    @torch.inference_mode()
    def layer_compute(self, layer: int, hidden_states: torch.Tensor, position_ids: torch.Tensor) -> torch.Tensor:
        """models/llama.py:185-220 (+ pre/post_attention_compute :134-183)."""
        W = self.layers[layer]
        B, H, Hkv, D = self.batch_size, self.shape.num_attention_heads, self.shape.num_key_value_heads, self._head_dim
        eps = self.shape.rms_norm_eps
        residual = hidden_states
        x = rms_norm(hidden_states, W["ln1"], eps)
        q = F.linear(x, W["wq"]).view(B, 1, H, D).transpose(1, 2)
        k = F.linear(x, W["wk"]).view(B, 1, Hkv, D).transpose(1, 2)
        v = F.linear(x, W["wv"]).view(B, 1, Hkv, D).transpose(1, 2)
        k = apply_rotary_pos_emb(k, self.cos_cache, self.sin_cache, position_ids)
        q = apply_rotary_pos_emb(q, self.cos_cache, self.sin_cache, position_ids)
        if layer in self.dense_index:
            attn = self._dense_attention(q.contiguous(), k, v, layer)
        else:
            decode = self.attention_server.decode_full_fused if self.fused_window else self.attention_server.decode_full
            attn = decode(q.contiguous(), k.contiguous(), v.contiguous(), self.sparse_index[layer])
        o = F.linear(attn.reshape(B, 1, H * D), W["wo"])
        self._all_reduce(o)                                   # llama_dist.py:209: partial o_proj outputs summed over the ranks
        h = residual + o
        y = rms_norm(h, W["ln2"], eps)
        y = F.linear(F.silu(F.linear(y, W["gate"])) * F.linear(y, W["up"]), W["down"])
        self._all_reduce(y)                                   # llama_dist.py:218: partial down_proj outputs
        return h + y

    def _all_reduce(self, t: torch.Tensor) -> None:
        """Sum `t` over the tensor-parallel ranks in place (RCCL through torch.distributed); nothing to do at TP = 1."""
        if self._all_reduce_fn is not None:
            self._all_reduce_fn(t)
            return
        if self.tp_world > 1:
            import torch.distributed as dist

            if not (dist.is_available() and dist.is_initialized()):
                raise RuntimeError("tensor-parallel decoder: no process group (torch.distributed.init_process_group first)")
            dist.all_reduce(t, op=dist.ReduceOp.SUM)

    @torch.inference_mode()
    def inference(self, input_ids: torch.Tensor, position_ids: torch.Tensor) -> torch.Tensor:
        """models/llama.py:288-301.  input_ids, position_ids: int64 [B, 1]."""
        self.plan()
        hidden_states = F.embedding(input_ids, self.embed_tokens)
        for layer in range(self.num_layers):
            hidden_states = self.layer_compute(layer, hidden_states, position_ids)
        hidden_states = rms_norm(hidden_states[:, -1:, :], self.norm_weight, self.shape.rms_norm_eps)
        return F.linear(hidden_states, self.lm_head).float()


def run_decode_benchmark(decoder: SyntheticLlamaDecoder, prompt_len: int, warmup: int = 32, steps: int = 128,
                         use_graph: bool = True):
    """examples/bench.py:43-59: prefill every request, 32 warm-up + 128 timed inference steps; returns
    (ms per step, tokens/s).  The window's generation buffer must hold every new token.  With
    use_graph the whole step (torch ops + HIP launches, including plan()'s length increments) is
    captured once in a hipGraph and replayed; the token ids / positions are copied into static
    buffers before each replay."""
    import time

    B = decoder.batch_size
    for b in range(B):
        decoder.prefill_synthetic(b, prompt_len, seed=b)
    g = torch.Generator(device=decoder.device).manual_seed(123)
    total = warmup + steps + 3
    ids = torch.randint(0, decoder.shape.vocab_size, (B, total), device=decoder.device, generator=g)
    pos = torch.arange(prompt_len, prompt_len + total, device=decoder.device).unsqueeze(0).repeat(B, 1)
    s_ids, s_pos = ids[:, :1].clone(), pos[:, :1].clone()
    graph = None
    done = 0
    if use_graph:
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):                       # eager passes: lazy initialisation outside the capture
                s_ids.copy_(ids[:, done:done + 1]); s_pos.copy_(pos[:, done:done + 1])
                decoder.inference(s_ids, s_pos)
                done += 1
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        s_ids.copy_(ids[:, done:done + 1]); s_pos.copy_(pos[:, done:done + 1])
        with torch.cuda.graph(graph):
            decoder.inference(s_ids, s_pos)
        # (a capture records the step without executing it: plan() leaves its host mirror alone under capture, the
        # lengths advance per replay -- LSHSparseAttnServer.replay)

    def one(i):
        if graph is not None:
            s_ids.copy_(ids[:, i:i + 1]); s_pos.copy_(pos[:, i:i + 1])
            decoder.attention_server.replay(graph)        # the replayed plan() advances the device counter
        else:
            decoder.inference(ids[:, i:i + 1], pos[:, i:i + 1])

    for i in range(done, done + warmup):
        one(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(done + warmup, done + warmup + steps):
        one(i)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    decoder.attention_server.window_server.check()
    if decoder.dense_layers:
        decoder.dense_server.check()
    return dt / steps * 1e3, B * steps / dt


def split_decode_step(decoder: SyntheticLlamaDecoder, reps: int = 8) -> dict:
    """Where a decode step's time goes (bench.py's `e2e` leg): the step's launches re-captured as five hipGraphs, one per
    kind of work over ALL layers -- (1) the sparse layers' attention (append + q-hash + retrieve + sampled attention +
    static window: the north-star path plus rows f-2), (2) the dense layers' attention, (3) the q / k / v / o and MLP
    projections, (4) RMSNorm + RoPE + residual adds, (5) embedding + final norm + lm_head -- each timed by HIP events
    around back-to-back replays.  The parts run on static inputs of the step's shapes; their sum is compared with the whole
    captured step by the caller.  Call after run_decode_benchmark (the stores are filled, lazy initialisation is done)."""
    d = decoder
    B, H, Hkv, D = d.batch_size, d.shape.num_attention_heads, d.shape.num_key_value_heads, d._head_dim
    hs, dev, dt = d.full_shape.hidden_size, d.device, d.dtype
    g = torch.Generator(device=dev).manual_seed(321)
    rnd = lambda *dims: torch.randn(dims, device=dev, generator=g).to(dt)          # noqa: E731
    x, q, k, v = rnd(B, 1, hs), rnd(B, H, 1, D), rnd(B, Hkv, 1, D), rnd(B, Hkv, 1, D)
    attn = rnd(B, 1, H * D)
    pos = torch.full((B, 1), d.max_length - 2, device=dev, dtype=torch.long)
    ids = torch.zeros((B, 1), device=dev, dtype=torch.long)
    eps = d.shape.rms_norm_eps

    def sparse_attn():
        for layer in d.sparse_layers:
            d.attention_server.decode_full_fused(q, k, v, d.sparse_index[layer])

    def dense_attn():
        for layer in d.dense_layers:
            d._dense_attention(q, k, v, layer)

    def projections():
        for W in d.layers:
            F.linear(x, W["wq"]); F.linear(x, W["wk"]); F.linear(x, W["wv"]); F.linear(attn, W["wo"])
            F.linear(F.silu(F.linear(x, W["gate"])) * F.linear(x, W["up"]), W["down"])

    def norms_rope():
        for W in d.layers:
            y = rms_norm(x, W["ln1"], eps)
            apply_rotary_pos_emb(k, d.cos_cache, d.sin_cache, pos)
            apply_rotary_pos_emb(q, d.cos_cache, d.sin_cache, pos)
            h = x + y
            h + rms_norm(h, W["ln2"], eps)

    def head():
        hdn = F.embedding(ids, d.embed_tokens)
        F.linear(rms_norm(hdn, d.norm_weight, eps), d.lm_head).float()

    out = {}
    with torch.inference_mode():
        for name, fn in (("sparse_attention", sparse_attn), ("dense_attention", dense_attn), ("projections_mlp", projections),
                         ("norms_rope_residual", norms_rope), ("embed_lm_head", head)):
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                fn()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                fn()
            graph.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                graph.replay()
            e1.record()
            torch.cuda.synchronize()
            out[name] = e0.elapsed_time(e1) / reps
            del graph
    d.attention_server.window_server.check()
    return out
