"""`LSHSparseAttnServer` -- device-resident counterpart of the reference's attention server
(models/attnserver.py:7-333) restricted to the LSH-sampled sparse path:

    fill        (:112-175)  split sink / local / offload, centre keys, key norms, key SimHash,
                            offloaded K/V/|k| -> HBM store
    build_table (:178-193)  per-table sort of the key codes -> CSR tables in HBM
    decode      (:264-300)  q SimHash -> batch_retrieve -> attention_wrapper, all in HBM
    decode_full (:228-312, sparse-layer branch) additionally does what the reference delegates to
                FlashInfer around the hot path: append this step's centred (k, v) to the static
                window (:275-290), exact attention over the window with a base-2 LSE (:293-296) and
                the LSE merge of the two partial attentions (:302-308)  [SURVEY.md 8(f) row f-2]
    clear       (:314-331)

Dense layers (0, 16, ...: FlashInfer full attention over the whole sequence, :235-259) are not part
of this path.  Everything runs on one device; there is no PCIe hop and no CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L
from .lsh import LSH
from .simhash import SimHash
from .sparse_attention import SparseAttentionServer


class LSHSparseAttnServer:
    def __init__(self, num_layers: int, num_attention_heads: int, num_key_value_heads: int,
                 head_dim: int, K: int = 10, L: int = 150, batch_size: int = 1,
                 num_sink_tokens: int = 4, num_local_tokens: int = 64, generation_buffer: int = 256,
                 max_length: int = 8192,
                 dense_layers=(0, 16, 32, 48, 64), device: str = "cuda:0",
                 dtype=torch.bfloat16, hash_func: torch.Tensor | None = None, seed: int = 7,
                 table_build: str = "counting", accel_budget_bytes: int | None = None, ranges: int = 0):
        """Mirrors models/attnserver.py:9-57 (the LlamaConfig is replaced by its four numbers).
        hash_func: bf16 [head_dim, K*L]; the reference draws it unseeded (:55), here it is
        seeded or supplied (e.g. broadcast from rank 0, attnserver_dist.py:279)."""
        assert dtype == torch.bfloat16
        self.K, self.L = K, L
        self.num_layers = num_layers
        self.batch_size = batch_size
        self.num_attention_heads = num_attention_heads
        self.num_key_value_heads = num_key_value_heads
        self.head_dim = head_dim
        self.max_length = max_length
        self.dense_layers = tuple(dense_layers)
        self.num_sink_tokens = num_sink_tokens
        self.num_local_tokens = num_local_tokens
        self.device = torch.device(device)
        self.table_build = table_build
        if hash_func is None:
            gen = torch.Generator(device="cpu").manual_seed(seed)
            hash_func = torch.randn((head_dim, K * L), generator=gen, dtype=torch.float32).to(dtype)
        self.hash_func = hash_func.to(self.device).contiguous()
        with torch.cuda.device(self.device):
            self.hasher = SimHash(self.hash_func, K, L)
            self.attn_server = SparseAttentionServer()
            self.attn_server.alloc(num_layers, num_attention_heads, num_key_value_heads, head_dim,
                                   batch_size, max_length)
            self.lsh_retriever = LSH()
            self.lsh_retriever.alloc(K, L, num_layers, num_attention_heads, num_key_value_heads,
                                     batch_size, max_length, accel_budget_bytes=accel_budget_bytes, ranges=ranges)
        BH = batch_size * num_attention_heads
        self.avg_k = [torch.zeros(batch_size, num_key_value_heads, 1, head_dim, device=self.device,
                                  dtype=dtype) for _ in range(num_layers)]
        self.hash_code_buffer = None
        self._filled = {}           # (layer, request) slots whose store was filled by fill(): build_table packs their norms
        self.output = torch.zeros((BH, head_dim), dtype=torch.bfloat16, device=self.device)
        self.max_value_expsum = torch.zeros((2, BH), dtype=torch.float32, device=self.device)
        self.nnz = torch.zeros((BH,), dtype=torch.int32, device=self.device)
        self.collect_nnz = True     # copy the per-head selected counts into self.nnz every decode
        # True (default): decode* leave what the reference's two calls leave -- query codes, result rows, logits
        # (get_mask / get_score work).  False: mp_decode_*_ex with MP_DECODE_NO_BYPRODUCTS -- output, LSE and counts only,
        # the selected ids handed to the gather unordered (a serving loop's setting; bench.py times this form)
        self.by_products = True
        # static window (sink + local + generated tokens), attnserver.py:25, 73-78, 97-106
        self.length = num_sink_tokens + num_local_tokens + generation_buffer
        with torch.cuda.device(self.device):
            self.window_server = SparseAttentionServer()
            self.window_server.alloc(num_layers, num_attention_heads, num_key_value_heads, head_dim,
                                     batch_size, self.length)
        self.kv_last_page_len = torch.zeros((batch_size,), dtype=torch.int32, device=self.device)
        self._window_rows = [0] * batch_size          # host mirror of kv_last_page_len (overflow check in plan())
        self.window_nnz = torch.zeros((BH,), dtype=torch.int32, device=self.device)
        self.window_out = torch.zeros((BH, head_dim), dtype=torch.bfloat16, device=self.device)
        self.window_mve = torch.zeros((2, BH), dtype=torch.float32, device=self.device)

    # ------------------------------------------------------------------ prefill side
    def fill(self, layer_idx: int, request_id: int, key_cache: torch.Tensor,
             value_cache: torch.Tensor, seq_len: int) -> None:
        """models/attnserver.py:112-175, sparse-layer branch.  key_cache / value_cache:
        bf16 [seq_len, Hkv, D] on the device."""
        s, l = self.num_sink_tokens, self.num_local_tokens
        # offloaded tokens: mean, centring, norms, K|V store and key SimHash (:136-175) in the store's own kernels
        # -- three passes over the KV cache, no transposed / centred intermediates
        avg_k, self.hash_code_buffer = self.attn_server.fill_offload(
            layer_idx, request_id, key_cache, value_cache, seq_len, s, l, hasher=self.hasher)
        self._filled[(layer_idx, request_id)] = True
        self.avg_k[layer_idx][request_id] = avg_k
        # sink + local tokens -> the static window, centred with the same avg_k (:126-153): a few dozen rows
        if s + l > 0:
            wkey = torch.cat([key_cache[:s], key_cache[seq_len - l:seq_len]], dim=0).transpose(0, 1) - avg_k
            wval = torch.cat([value_cache[:s], value_cache[seq_len - l:seq_len]], dim=0).transpose(0, 1)
            wkey = wkey.contiguous()
            self.window_server.fill(layer_idx, request_id, wkey, wval.contiguous(),
                                    wkey.norm(p=2, dim=-1).float())
        self.set_window_rows(request_id, s + l)

    def build_table(self, layer_idx: int, request_id: int, seq_len: int) -> None:
        """models/attnserver.py:178-193: sort the codes of every (kv head, table) row, then
        LSH::fill.  table_build='sort' follows the reference (torch.sort on the device);
        'counting' uses the device counting sort (LSH.fastfill)."""
        codes = self.hash_code_buffer
        if self.table_build == "counting":
            # the store of (layer, request) was filled first (fill() above, as models/attnserver.py:174 precedes :178):
            # the sort packs the key norms into the table words in the same pass
            self.lsh_retriever.fastfill(layer_idx, request_id, codes,
                                        self.attn_server if self._filled.get((layer_idx, request_id)) else None)
        else:
            sorted_values, sorted_indices = codes.sort(dim=-1)
            self.lsh_retriever.fill(layer_idx, request_id, sorted_values.contiguous(),
                                    sorted_indices.int().contiguous())
        self.hash_code_buffer = None

    # ------------------------------------------------------------------ decode side
    def decode(self, query_states: torch.Tensor, layer_idx: int):
        """models/attnserver.py:264-300 on one device: returns (cpu_hidden_states bf16 [B, H, D],
        cpu_lse f32 [B, H]) -- the two operands the reference hands to flashinfer.merge_state
        (:305-308) -- for the offloaded part of the context.

        Both are VIEWS of this server's persistent output buffers (as the reference's self.output_cuda /
        self.max_value_expsum_cuda are, :302-304): the next decode* call of any layer overwrites them.
        Clone them to keep per-layer results."""
        BH = self.batch_size * self.num_attention_heads
        q = query_states.reshape(BH, self.head_dim)
        L.expect(q, torch.bfloat16, (BH, self.head_dim), "query_states")
        L.check(L.lib().mp_decode_sparse_layer_ex(
            self.hasher._h, self.lsh_retriever._h, self.attn_server._h, layer_idx, L.ptr(q),
            L.ptr(self.output), L.ptr(self.max_value_expsum),
            L.ptr(self.nnz if self.collect_nnz else None), 0 if self.by_products else L.DECODE_NO_BYPRODUCTS,
            L.current_stream(q)))
        out = self.output.view(self.batch_size, self.num_attention_heads, self.head_dim)
        lse = self.max_value_expsum[1].view(self.batch_size, self.num_attention_heads)
        return out, lse

    def set_window_rows(self, request_id: int, rows: int) -> None:
        """Number of live rows of request `request_id`'s static window (device counter + host mirror)."""
        if not 0 <= rows <= self.length:
            raise ValueError(f"window of request {request_id}: {rows} rows do not fit the {self.length} allocated")
        self.kv_last_page_len[request_id] = rows
        self._window_rows[request_id] = rows

    def account_steps(self, steps: int) -> None:
        """Host-side bookkeeping for callers that advance the device counter WITHOUT calling plan() on the
        host -- a hipGraph that captured plan() and is replayed `steps` times (or, with steps = -1, the
        capture itself, which runs plan() on the host but executes nothing on the device).  Raises like
        plan() when the replays would overflow the window."""
        rows = [r + steps for r in self._window_rows]
        full = [b for b, r in enumerate(rows) if r > self.length]
        if full:
            raise L.MagicPigError(6, f"static window full: request(s) {full} would hold more than {self.length} rows")
        self._window_rows = rows

    def replay(self, graph: "torch.cuda.CUDAGraph") -> None:
        """Replay a hipGraph that captured one decode step (plan() + the layers' decode_full* calls) with the
        host-side bookkeeping plan() would have done: raises instead of replaying once the window is full."""
        self.account_steps(1)
        graph.replay()

    def plan(self) -> None:
        """models/attnserver.py:196-198: one more token in every request's window this step.  The static
        window holds sink + local + generation_buffer rows (:25): a step past that raises here -- the
        reference would index past its FlashInfer pages (kv_last_page_len > page_size)."""
        full = [b for b, r in enumerate(self._window_rows) if r + 1 > self.length]
        if full:
            raise L.MagicPigError(6, f"static window full: request(s) {full} already hold {self.length} rows "
                                     "(num_sink_tokens + num_local_tokens + generation_buffer); "
                                     "raise generation_buffer")
        # under stream capture nothing executes on the device: the host mirror advances per REPLAY (replay())
        if not torch.cuda.is_current_stream_capturing():
            self._window_rows = [r + 1 for r in self._window_rows]
        self.kv_last_page_len += 1
        self.window_nnz.copy_(self.kv_last_page_len.repeat_interleave(self.num_attention_heads))

    def decode_full(self, query_states: torch.Tensor, key_states: torch.Tensor,
                    value_states: torch.Tensor, layer_idx: int) -> torch.Tensor:
        """models/attnserver.py:261-312 (sparse-layer branch) entirely on the device: returns
        hidden_states bf16 [B, 1, H*D] (a fresh tensor).  Call plan() once per step before the first layer;
        plan() raises once the static window (sink + local + generation_buffer rows) is full."""
        B, H, Hkv, D = self.batch_size, self.num_attention_heads, self.num_key_value_heads, self.head_dim
        q = query_states.reshape(B * H, D)
        k = (key_states.reshape(B, Hkv, 1, D) - self.avg_k[layer_idx]).reshape(B, Hkv, D).contiguous()
        v = value_states.reshape(B, Hkv, D).contiguous()
        self.window_server.append(layer_idx, k, v, self.kv_last_page_len - 1)          # :275-290
        self.window_server.full_attention(layer_idx, self.window_out, self.window_mve, q,
                                          self.window_nnz)                             # :293-296
        sparse_out, sparse_lse = self.decode(query_states, layer_idx)                  # :264-300
        hidden, _ = self.merge(self.window_out.view(B, H, D), self.window_mve[1].view(B, H),
                               sparse_out, sparse_lse)                                 # :302-308
        return hidden.reshape(B, 1, H * D)

    def decode_full_fused(self, query_states: torch.Tensor, key_states: torch.Tensor,
                          value_states: torch.Tensor, layer_idx: int) -> torch.Tensor:
        """decode_full in two launches instead of four: the append, then ONE kernel in which the exact
        attention over the static window joins the softmax of the LSH-sampled tokens
        (mp_decode_layer_window) -- no window partial, no merge_state.  Falls back to decode_full when
        the one-launch form does not exist for the shape.  The result is a VIEW of the persistent output
        buffer (overwritten by the next decode* call); plan() guards the window's capacity."""
        B, H, Hkv, D = self.batch_size, self.num_attention_heads, self.num_key_value_heads, self.head_dim
        q = query_states.reshape(B * H, D)
        L.expect(q, torch.bfloat16, (B * H, D), "query_states")
        k = key_states.reshape(B, Hkv, D).contiguous()
        v = value_states.reshape(B, Hkv, D).contiguous()
        self.window_server.append_centred(layer_idx, k, v, self.avg_k[layer_idx].view(B, Hkv, D),
                                          self.kv_last_page_len, -1)
        rc = L.lib().mp_decode_layer_window_ex(
            self.hasher._h, self.lsh_retriever._h, self.attn_server._h, self.window_server._h, layer_idx,
            L.ptr(q), L.ptr(self.window_nnz), L.ptr(self.output), L.ptr(self.max_value_expsum),
            L.ptr(self.nnz if self.collect_nnz else None), 0 if self.by_products else L.DECODE_NO_BYPRODUCTS,
            L.current_stream(q))
        if rc == L.ERR_UNSUPPORTED:
            self.window_server.full_attention(layer_idx, self.window_out, self.window_mve, q, self.window_nnz)
            sparse_out, sparse_lse = self.decode(query_states, layer_idx)
            hidden, _ = self.merge(self.window_out.view(B, H, D), self.window_mve[1].view(B, H),
                                   sparse_out, sparse_lse)
            return hidden.reshape(B, 1, H * D)
        L.check(rc)
        return self.output.view(B, 1, H * D)

    @staticmethod
    def merge(gpu_hidden_states, gpu_lse, cpu_hidden_states, cpu_lse):
        """flashinfer.merge_state as used at models/attnserver.py:308 (base-2 LSEs)."""
        D = gpu_hidden_states.shape[-1]
        va = gpu_hidden_states.reshape(-1, D).contiguous()
        vb = cpu_hidden_states.reshape(-1, D).contiguous()
        sa = gpu_lse.reshape(-1).float().contiguous()
        sb = cpu_lse.reshape(-1).float().contiguous()
        R = va.shape[0]
        v = torch.empty_like(va)
        s = torch.empty_like(sa)
        with torch.cuda.device(va.device):               # mp_merge_state has no handle: launch where the data is
            L.check(L.lib().mp_merge_state(L.ptr(va), L.ptr(sa), L.ptr(vb), L.ptr(sb), R, D, L.ptr(v),
                                           L.ptr(s), L.current_stream(va)))
        return v.view_as(gpu_hidden_states), s.view_as(gpu_lse)

    def clear(self) -> None:
        """models/attnserver.py:314-331."""
        self.nnz.zero_()
        self.max_value_expsum.zero_()
        self.output.zero_()
        for i in range(self.num_layers):
            self.avg_k[i].zero_()
        self.kv_last_page_len.zero_()
        self._window_rows = [0] * self.batch_size
        self.window_nnz.zero_()
        self._filled = {}
        self.lsh_retriever.clear()
        self.attn_server.clear()
        self.window_server.clear()
