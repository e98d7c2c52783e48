"""`SimHash` -- the query / key hashing of models/attnserver.py:55-57, 159-168, 264-270 on the
MFMA matrix cores, behind the C ABI (mp_simhash_*).  Codes are bit-exact (exact-sign guard)."""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib as L


class SimHash:
    def __init__(self, hash_func: torch.Tensor, K: int, L_: int):
        """hash_func: bf16 [head_dim, K*L], the tensor of models/attnserver.py:55."""
        D = hash_func.shape[0]
        L.expect(hash_func, torch.bfloat16, (D, K * L_), "hash_func")
        self._h = C.c_void_p()
        L.check(L.lib().mp_simhash_create(C.byref(self._h)))
        self.D, self.K, self.L = D, K, L_
        # the planes live on hash_func's device when it is a CUDA tensor, else on the current device
        self._device = hash_func.device.index if hash_func.is_cuda else L.current_device()
        with torch.cuda.device(self._device):
            L.check(L.lib().mp_simhash_set_planes(self._h, D, K, L_, L.ptr(hash_func),
                                                  L.mem_kind(hash_func), L.current_stream(hash_func, self._device)))

    def __del__(self):
        try:
            if getattr(self, "_h", None) and self._h.value:
                L.lib().mp_simhash_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    def query(self, q: torch.Tensor, codes: torch.Tensor | None = None,
              qnorm: torch.Tensor | None = None):
        """models/attnserver.py:264-270: q bf16 [..., D] -> (q_hashcode int32 [R, L],
        ||q||_2 f32 [R]) on the same side (CPU / GPU) as q."""
        q2 = q.reshape(-1, self.D)
        L.expect(q2, torch.bfloat16, None, "q")
        R = q2.shape[0]
        if codes is None:           # (a tensor made here needs no checking)
            codes = torch.empty((R, self.L), dtype=torch.int32, device=q.device)
        else:
            L.expect(codes, torch.int32, (R, self.L), "codes")
        if qnorm is None:
            qnorm = torch.empty((R,), dtype=torch.float32, device=q.device)
        else:
            L.expect(qnorm, torch.float32, (R,), "qnorm")
        mem = L.same_memory(q2, codes, qnorm)
        L.check(L.lib().mp_simhash_query(self._h, L.ptr(q2), R, L.ptr(codes), L.ptr(qnorm), mem,
                                         L.current_stream(q2, self._device)))
        return codes, qnorm

    def keys(self, keys: torch.Tensor, codes: torch.Tensor | None = None) -> torch.Tensor:
        """models/attnserver.py:159-168: centred keys bf16 [Hkv, n, D] -> int16 [Hkv, L, n]."""
        Hkv, n, D = keys.shape
        L.expect(keys, torch.bfloat16, (Hkv, n, self.D), "keys")
        if codes is None:
            codes = torch.empty((Hkv, self.L, n), dtype=torch.int16, device=keys.device)
        L.expect(codes, torch.int16, (Hkv, self.L, n), "codes")
        mem = L.same_memory(keys, codes)
        L.check(L.lib().mp_simhash_keys(self._h, L.ptr(keys), Hkv, n, L.ptr(codes), mem,
                                        L.current_stream(keys, self._device)))
        return codes

    def _debug_acc(self, buf: torch.Tensor | None) -> None:
        """test hook: route raw MFMA accumulators f32 [R, K*L] of later query() calls to `buf`."""
        L.check(L.lib().mp_simhash_debug_acc(self._h, L.ptr(buf)))
