"""`SparseAttentionServer` -- host-side mirror of the reference's pybind11 class
(library/sparse_attention/sparse_attention.cc:1243-1263) over the gfx950 C ABI.  KV cache, key
norms and scratch live in HBM; per-call tensors are caller-owned and written in place; they may
be CPU tensors (staged) or CUDA tensors (zero-copy fast path)."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L


class SparseAttentionServer:
    def __init__(self):                                   # sparse_attention.cc:519-527
        self._h = C.c_void_p()
        L.check(L.lib().mp_attn_create(C.byref(self._h)))

    def __del__(self):                                    # sparse_attention.cc:529-544
        try:
            if getattr(self, "_h", None) and self._h.value:
                L.lib().mp_attn_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def alloc(self, num_layers: int, num_attention_heads: int, num_key_value_heads: int,
              head_dim: int, batch_size: int, max_length: int) -> None:
        """SparseAttentionServer::alloc, sparse_attention.cc:546-583."""
        L.check(L.lib().mp_attn_alloc(self._h, num_layers, num_attention_heads,
                                      num_key_value_heads, head_dim, batch_size, max_length))
        self.num_layers = num_layers
        self.H, self.Hkv, self.D, self.B, self.M = (num_attention_heads, num_key_value_heads,
                                                    head_dim, batch_size, max_length)
        self._device = L.current_device()      # the handle's state lives here (the C ABI switches to it)

    def fill(self, layer_id: int, request_id: int, k: torch.Tensor, v: torch.Tensor,
             kn: torch.Tensor) -> None:
        """SparseAttentionServer::fill, sparse_attention.cc:601-627: k, v bf16 [Hkv,n,D], kn f32 [Hkv,n]."""
        n = k.shape[1]
        L.expect(k, torch.bfloat16, (self.Hkv, n, self.D), "k")
        L.expect(v, torch.bfloat16, (self.Hkv, n, self.D), "v")
        L.expect(kn, torch.float32, (self.Hkv, n), "kn")
        mem = L.same_memory(k, v, kn)
        L.check(L.lib().mp_attn_fill(self._h, layer_id, request_id, L.ptr(k), L.ptr(v), L.ptr(kn), n,
                                     mem, L.current_stream(k, self._device)))

    def fill_offload(self, layer_id: int, request_id: int, key_cache: torch.Tensor, value_cache: torch.Tensor,
                     seq_len: int, num_sink: int, num_local: int, hasher=None):
        """Not in the reference class: the torch lines around SparseAttentionServer::fill in
        LSHSparseAttnServer.fill (models/attnserver.py:126-175) folded into the store's own kernels.
        key_cache / value_cache: bf16 CUDA tensors [>= seq_len, Hkv, D] (token-major).  Stores the centred keys,
        values and key norms of tokens [num_sink, seq_len - num_local) and returns (avg_k bf16 [Hkv, 1, D],
        key codes int16 [Hkv, L, n] or None when no hasher is given)."""
        L.expect(key_cache[:seq_len], torch.bfloat16, (seq_len, self.Hkv, self.D), "key_cache")
        L.expect(value_cache[:seq_len], torch.bfloat16, (seq_len, self.Hkv, self.D), "value_cache")
        if not (key_cache.is_cuda and value_cache.is_cuda):
            raise ValueError("fill_offload takes CUDA tensors")
        if key_cache.device.index != self._device or value_cache.device.index != self._device:
            raise ValueError(f"fill_offload: caches on {key_cache.device} / {value_cache.device}, the store lives on "
                             f"cuda:{self._device}")
        n = seq_len - num_sink - num_local
        if n <= 0:
            raise ValueError(f"fill_offload: nothing to offload (seq_len {seq_len} <= sink {num_sink} + local {num_local})")
        avg = torch.empty((self.Hkv, 1, self.D), dtype=torch.bfloat16, device=key_cache.device)
        codes = None
        if hasher is not None:
            codes = torch.empty((self.Hkv, hasher.L, n), dtype=torch.int16, device=key_cache.device)
        L.check(L.lib().mp_attn_fill_offload(self._h, hasher._h if hasher is not None else None, layer_id, request_id,
                                             L.ptr(key_cache), L.ptr(value_cache), seq_len, num_sink, num_local,
                                             L.ptr(avg), L.ptr(codes), L.current_stream(key_cache, self._device)))
        return avg, codes

    def attention_wrapper(self, layer_id: int, K: int, L_: int, output: torch.Tensor,
                          max_value_expsum: torch.Tensor, query: torch.Tensor,
                          query_norm: torch.Tensor, ind: torch.Tensor, nnz: torch.Tensor) -> None:
        """attention_wrapper, sparse_attention.cc:629-745 (all dispatch targets are one kernel
        here).  output bf16 [B*H,D]; max_value_expsum f32 [2,B*H]; query bf16 (or f32, the
        non-AVX512BF16 build's `.to(kFloat32)`) [B*H,D]; query_norm f32 [B*H]; ind int32 [B*H,M];
        nnz int32 [B*H]."""
        BH = self.B * self.H
        L.expect(output, torch.bfloat16, (BH, self.D), "output")
        L.expect(max_value_expsum, torch.float32, (2, BH), "max_value_expsum")
        qdt = query.dtype
        if qdt is not torch.bfloat16 and qdt is not torch.float32:
            query = query.float()
            qdt = torch.float32
        L.expect(query, None, (BH, self.D), "query", same_numel_ok=True)
        L.expect(query_norm, torch.float32, (BH,), "query_norm")
        L.expect(ind, torch.int32, (BH, self.M), "ind")
        L.expect(nnz, torch.int32, (BH,), "nnz")
        mem = L.same_memory(output, max_value_expsum, query, query_norm, ind, nnz)
        qd = L.DTYPE_BF16 if qdt is torch.bfloat16 else L.DTYPE_F32
        L.check(L.lib().mp_attn_sparse(self._h, layer_id, K, L_, output.data_ptr(), max_value_expsum.data_ptr(),
                                       query.data_ptr(), qd, query_norm.data_ptr(), ind.data_ptr(), nnz.data_ptr(),
                                       mem, L.current_stream(output, self._device)))

    # every reference variant computes the same function (sparse_attention.cc:748-986, 1039-1211)
    attention = attention_wrapper
    scheduled_attention = attention_wrapper
    attention_bf16 = attention_wrapper
    attention_wrapper_bf16 = attention_wrapper

    def full_attention(self, layer_id: int, output: torch.Tensor, max_value_expsum: torch.Tensor,
                       query: torch.Tensor, nnz: torch.Tensor) -> None:
        """full_attention, sparse_attention.cc:988-1037 (dense, K == 0 baseline)."""
        BH = self.B * self.H
        L.expect(output, torch.bfloat16, (BH, self.D), "output")
        L.expect(max_value_expsum, torch.float32, (2, BH), "max_value_expsum")
        if query.dtype not in (torch.bfloat16, torch.float32):
            query = query.float()
        L.expect(query, None, (BH, self.D), "query", same_numel_ok=True)
        L.expect(nnz, torch.int32, (BH,), "nnz")
        mem = L.same_memory(output, max_value_expsum, query, nnz)
        qd = L.DTYPE_BF16 if query.dtype == torch.bfloat16 else L.DTYPE_F32
        L.check(L.lib().mp_attn_full(self._h, layer_id, L.ptr(output), L.ptr(max_value_expsum),
                                     L.ptr(query), qd, L.ptr(nnz), mem, L.current_stream(output, self._device)))

    def append(self, layer_id: int, k: torch.Tensor, v: torch.Tensor, pos: torch.Tensor) -> None:
        """Not in the reference class: the role of flashinfer.append_paged_kv_cache
        (models/attnserver.py:281-290) for a store used as the static window.  k, v bf16
        [B, Hkv, D], pos int32 [B] (row to write for each request); CUDA tensors."""
        L.expect(k, torch.bfloat16, (self.B, self.Hkv, self.D), "k")
        L.expect(v, torch.bfloat16, (self.B, self.Hkv, self.D), "v")
        L.expect(pos, torch.int32, (self.B,), "pos")
        if not (k.is_cuda and v.is_cuda and pos.is_cuda):
            raise ValueError("append takes CUDA tensors")
        L.check(L.lib().mp_attn_append(self._h, layer_id, L.ptr(k), L.ptr(v), L.ptr(pos), L.current_stream(k, self._device)))

    def append_centred(self, layer_id: int, k: torch.Tensor, v: torch.Tensor, centre: torch.Tensor,
                       pos: torch.Tensor, pos_delta: int = 0) -> None:
        """append() with `k - centre` (bf16 semantics of models/attnserver.py:267) and the row offset
        folded into the same launch.  centre bf16 [B, Hkv, D]."""
        L.expect(k, torch.bfloat16, (self.B, self.Hkv, self.D), "k")
        L.expect(v, torch.bfloat16, (self.B, self.Hkv, self.D), "v")
        L.expect(centre, torch.bfloat16, (self.B, self.Hkv, self.D), "centre")
        L.expect(pos, torch.int32, (self.B,), "pos")
        if not (k.is_cuda and v.is_cuda and pos.is_cuda and centre.is_cuda):
            raise ValueError("append_centred takes CUDA tensors")
        L.check(L.lib().mp_attn_append_centred(self._h, layer_id, L.ptr(k), L.ptr(v), L.ptr(centre), L.ptr(pos),
                                               pos_delta, L.current_stream(k, self._device)))

    def invalidate_norms(self, layer_id: int, request_id: int) -> None:
        """After writing key norms through the get_key_norm() view: the request's norms get a new version, so that the
        one-launch decode packs them into its LSH table words again (include/magicpig_hip.h: mp_attn_invalidate_norms)."""
        L.check(L.lib().mp_attn_invalidate_norms(self._h, layer_id, request_id, L.current_stream(device=self._device)))

    def check(self) -> None:
        """Raise if a device-side validation failed since the last check (append past max_length)."""
        L.check(L.lib().mp_attn_check(self._h, L.current_stream(device=self._device)))

    def clear(self) -> None:
        """SparseAttentionServer::clear, sparse_attention.cc:586-598."""
        L.check(L.lib().mp_attn_clear(self._h, L.current_stream(device=self._device)))

    # ---- views of handle-owned HBM (sparse_attention.cc:1213-1241); K|V are interleaved per
    # token in HBM, so the key/value caches come back as strided (non-contiguous) CUDA views.
    def _kv(self, layer_id: int, which: int) -> torch.Tensor:
        k, v, stride = C.c_void_p(), C.c_void_p(), C.c_int64()
        L.check(L.lib().mp_attn_get_kv(self._h, layer_id, C.byref(k), C.byref(v), C.byref(stride)))
        rs = stride.value * 2
        t = L.device_tensor((k.value, v.value)[which], (self.B, self.Hkv, self.M, self.D), "<i2",
                            (self.Hkv * self.M * rs, self.M * rs, rs, 2), device=torch.device("cuda", self._device))
        return t.view(torch.bfloat16)

    def get_key_cache(self, layer_id: int, contiguous: bool = False) -> torch.Tensor:
        """get_key_cache, sparse_attention.cc:1213-1222: bf16 [B, Hkv, M, D].  The reference returns an alias of one
        contiguous blob; here a token's K and V rows are interleaved in HBM, so the alias is a STRIDED view (element
        stride 2 D between tokens: `.view(-1)` on it raises).  contiguous=True returns a contiguous COPY for callers
        that reshape the cache (writes to the copy do not reach the store)."""
        t = self._kv(layer_id, 0)
        return t.contiguous() if contiguous else t

    def get_value_cache(self, layer_id: int, contiguous: bool = False) -> torch.Tensor:
        """get_value_cache, sparse_attention.cc:1223-1233 (see get_key_cache)."""
        t = self._kv(layer_id, 1)
        return t.contiguous() if contiguous else t

    def footprint(self) -> dict:
        """HBM bytes per layer of the store (mp_attn_get_footprint): interleaved K | V rows and f32 key norms."""
        b = (C.c_int64 * 2)()
        L.check(L.lib().mp_attn_get_footprint(self._h, b))
        return {"kv": int(b[0]), "key_norms": int(b[1])}

    def get_key_norm(self, layer_id: int) -> torch.Tensor:
        p = C.c_void_p()
        L.check(L.lib().mp_attn_get_key_norm(self._h, layer_id, C.byref(p)))
        return L.device_tensor(p.value, (self.B, self.Hkv, self.M), "<f4", device=torch.device("cuda", self._device))

    def get_score(self) -> torch.Tensor:
        """get_score, sparse_attention.cc:1235-1241: f32 [B,H,M] probabilities of the last call
        (first nnz entries per head, in `ind` order)."""
        p = C.c_void_p()
        L.check(L.lib().mp_attn_get_score(self._h, C.byref(p), L.current_stream(device=self._device)))
        return L.device_tensor(p.value, (self.B, self.H, self.M), "<f4", device=torch.device("cuda", self._device))
