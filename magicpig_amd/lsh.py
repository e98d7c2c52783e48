"""`LSH` -- host-side mirror of the reference's pybind11 class (library/lsh/lsh.cc:316-326) over
the gfx950 C ABI.  Same method names, argument order and in-place outputs, so a caller written
against models/attnserver.py:50-53,191-193,299,330 works unchanged; tensors may be CPU tensors
(staged through HBM, like the reference's pinned buffers) or CUDA tensors (used in place)."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L


class LSH:
    def __init__(self):                                   # LSH::LSH(), lsh.cc:25-27
        self._h = C.c_void_p()
        L.check(L.lib().mp_lsh_create(C.byref(self._h)))
        self._alloc = False

    def __del__(self):                                    # LSH::~LSH(), lsh.cc:29-42
        try:
            if getattr(self, "_h", None) and self._h.value:
                L.lib().mp_lsh_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def alloc(self, K: int, L_: int, num_layers: int, num_attention_heads: int,
              num_key_value_heads: int, batch_size: int, max_length: int, accel_budget_bytes: int | None = None,
              ranges: int = 0) -> None:
        """LSH::alloc, lsh.cc:44-91.  accel_budget_bytes / ranges (not in the reference; mp_lsh_alloc_ex): how much HBM the
        handle may spend on structures that only make the decode faster (direct piece slots, the host-buffer mode's row
        copy; None = the library's rule: a third of what is free at the time), and the token ranges per table row
        (0 = auto)."""
        if accel_budget_bytes is None and ranges == 0:
            L.check(L.lib().mp_lsh_alloc(self._h, K, L_, num_layers, num_attention_heads,
                                         num_key_value_heads, batch_size, max_length))
        else:
            L.check(L.lib().mp_lsh_alloc_ex(self._h, K, L_, num_layers, num_attention_heads, num_key_value_heads, batch_size,
                                            max_length, -1 if accel_budget_bytes is None else int(accel_budget_bytes), ranges))
        self.K, self.L, self.num_layers = K, L_, num_layers
        self.H, self.Hkv, self.B, self.M = (num_attention_heads, num_key_value_heads, batch_size,
                                            max_length)
        self.NB = 1 << K
        r, rl = C.c_int(0), C.c_int(0)
        L.check(L.lib().mp_lsh_get_ranges(self._h, C.byref(r), C.byref(rl)))
        self.R, self.range_len = r.value, rl.value      # token ranges per table row (= workgroups per head in decode)
        self._device = L.current_device()      # the handle's state lives here (the C ABI switches to it)
        self._alloc = True

    def fill(self, layer_id: int, request_id: int, sorted_hash_code: torch.Tensor,
             sorted_indices: torch.Tensor) -> None:
        """LSH::fill, lsh.cc:143-201: sorted codes int16 [Hkv,L,n] + token ids int32 [Hkv,L,n]."""
        n = sorted_hash_code.shape[-1]
        L.expect(sorted_hash_code, torch.int16, (self.Hkv, self.L, n), "sorted_hash_code")
        L.expect(sorted_indices, torch.int32, (self.Hkv, self.L, n), "sorted_indices")
        mem = L.same_memory(sorted_hash_code, sorted_indices)
        L.check(L.lib().mp_lsh_fill(self._h, layer_id, request_id, L.ptr(sorted_hash_code),
                                    L.ptr(sorted_indices), n, mem, L.current_stream(sorted_hash_code, self._device)))

    def fastfill(self, layer_id: int, request_id: int, hash_code: torch.Tensor, attn_server=None) -> None:
        """Working version of LSH::fastfill (lsh.cc:93-142, unfinished in the reference): builds
        the tables on device from UNSORTED codes int16 [Hkv,L,n].  attn_server: the SparseAttentionServer whose
        (layer, request) slot was ALREADY filled (the reference's order, models/attnserver.py:174 before :178-193):
        the sort then packs the key norms into the table words itself (mp_lsh_build_with_norms) and the first decode
        of the layer has nothing left to do; same tables, same results."""
        n = hash_code.shape[-1]
        L.expect(hash_code, torch.int16, (self.Hkv, self.L, n), "hash_code")
        if attn_server is not None:
            L.check(L.lib().mp_lsh_build_with_norms(self._h, attn_server._h, layer_id, request_id, L.ptr(hash_code), n,
                                                    L.mem_kind(hash_code), L.current_stream(hash_code, self._device)))
            return
        L.check(L.lib().mp_lsh_build(self._h, layer_id, request_id, L.ptr(hash_code), n,
                                     L.mem_kind(hash_code), L.current_stream(hash_code, self._device)))

    def batch_retrieve(self, layer_id: int, query: torch.Tensor, results: torch.Tensor,
                       nnz: torch.Tensor) -> None:
        """LSH::batch_retrieve, lsh.cc:210-241: query int32 [B*H,L] -> results int32 [B*H,M]
        (first nnz[h] valid, ascending ids), nnz int32 [B*H]."""
        BH = self.B * self.H
        L.expect(query, torch.int32, (BH, self.L), "query")
        L.expect(results, torch.int32, (BH, self.M), "results")
        L.expect(nnz, torch.int32, (BH,), "nnz")
        mem = L.same_memory(query, results, nnz)
        L.check(L.lib().mp_lsh_batch_retrieve(self._h, layer_id, query.data_ptr(), results.data_ptr(),
                                              nnz.data_ptr(), mem, L.current_stream(query, self._device)))

    def clear(self) -> None:
        """LSH::clear, lsh.cc:293-306."""
        L.check(L.lib().mp_lsh_clear(self._h, L.current_stream(device=self._device)))

    def copy(self, query: torch.Tensor) -> None:
        """LSH::copy, lsh.cc:203-207: empty in the reference; kept for API parity."""
        return None

    def get_mask(self) -> torch.Tensor:
        """LSH::get_mask, lsh.cc:308-314: int8 [B,H,M] collision counters min(count,2) of the last
        batch_retrieve (recomputed on demand; returned as a CPU tensor like the reference's)."""
        out = torch.zeros((self.B, self.H, self.M), dtype=torch.int8)
        L.check(L.lib().mp_lsh_get_mask(self._h, L.ptr(out), L.MEM_HOST, L.current_stream(device=self._device)))
        return out

    # debug views (declared but never defined in the reference, lsh.h:24-26)
    def id_bits(self, layer_id: int) -> int:
        """17 while every id of the layer's tables is below 2^17 (a table word = token id | payload << 17,
        include/magicpig_hip.h), else 0."""
        bits = C.c_int32(0)
        L.check(L.lib().mp_lsh_get_id_bits(self._h, layer_id, C.byref(bits)))
        return bits.value

    def footprint(self) -> dict:
        """HBM bytes per layer of the index structures (mp_lsh_get_footprint): the reference's table (lsh.cc:44-91)
        plus this implementation's sub-bounds and direct piece slots."""
        b = (C.c_int64 * 8)()
        L.check(L.lib().mp_lsh_get_footprint_ex(self._h, b))
        return {"bounds": int(b[0]), "table": int(b[1]), "slots": int(b[2]), "slot_bytes": int(b[3]),
                "host_mode_row_copy": int(b[4]), "host_mode_pinned": int(b[5]), "accel_budget": int(b[6]),
                "accel_in_use": int(b[7])}

    def get_tables(self, layer_id: int, raw: bool = False):
        b, t = C.c_void_p(), C.c_void_p()
        L.check(L.lib().mp_lsh_get_tables(self._h, layer_id, C.byref(b), C.byref(t)))
        groups = self.B * self.Hkv
        dev = torch.device("cuda", self._device)
        # entry 0 = start, entry R = end of a bucket; entry r = first position whose token id >= r * range_len
        bounds = L.device_tensor(b.value, (groups, self.L, self.NB, self.R + 1), "<i4", device=dev)
        table = L.device_tensor(t.value, (groups, self.L, self.M), "<i4", device=dev)
        bits = self.id_bits(layer_id)
        if bits and not raw:    # entries may carry a key-norm payload above the id (packed by the decode entry): show ids
            table = table & ((1 << bits) - 1)
        return bounds, table
