"""ctypes/numpy binding of the CPU oracle (oracle/mp_oracle.c).

TEST INFRASTRUCTURE, NOT PRODUCT (see oracle/__init__.py).  The two classes mirror the
reference's pybind11 classes (library/lsh/lsh.cc:316-326,
library/sparse_attention/sparse_attention.cc:1243-1263) so that the parity tests read like
the reference's own tests: same method names, same argument order, caller-owned outputs
written in place.  Arguments may be torch CPU tensors or numpy arrays.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_BUILD = os.path.join(_HERE, "_build")
_lock = threading.Lock()
_lib = None


def _cpu_has(*flags: str) -> bool:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    have = set(line.split(":", 1)[1].split())
                    return all(fl in have for fl in flags)
    except OSError:
        pass
    return False


def build(native: bool = False, force: bool = False) -> str:
    """Compile mp_oracle.c with gcc (oracle/Makefile).  Returns the path of the .so."""
    target = "native" if native else "all"
    name = "libmp_oracle_native.so" if native else "libmp_oracle.so"
    path = os.path.join(_BUILD, name)
    src = os.path.join(_HERE, "mp_oracle.c")
    if force or not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        subprocess.run(["make", "-C", _HERE, target], check=True, capture_output=True)
    return path


def lib() -> C.CDLL:
    """Load (building on demand) the oracle shared library."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        # the prebuilt portable build needs AVX2+FMA; otherwise (or if missing) build natively
        if _cpu_has("avx2", "fma"):
            path = build(native=False)
        else:
            path = build(native=True)
        L = C.CDLL(path)
        i32, i64, f32 = C.c_int, C.c_int64, C.c_float
        p = C.c_void_p
        L.mpo_version.restype = i32
        L.mpo_max_threads.restype = i32
        L.mpo_simhash_query.argtypes = [p, p, i32, i32, i32, i32, p, p]
        L.mpo_simhash_keys.argtypes = [p, p, i64, i32, i32, i32, p]
        L.mpo_lsh_fill.argtypes = [p, p, i32, i32, i64, i32, i64, p, p, p]
        L.mpo_lsh_batch_retrieve.argtypes = [p, p, p, p, i32, i32, i32, i32, i64, p, p, p, i32]
        L.mpo_sparse_attention.argtypes = [p, p, p, p, i32, p, p, p, i32, i32, i32, i64, i32,
                                           i32, i32, i32, p, p, p, i32]
        L.mpo_full_attention.argtypes = [p, p, p, p, i32, i32, i32, i64, p, p, p, i32, i32]
        L.mpo_merge_state.argtypes = [p, p, p, p, i32, i32, p, p]
        for fn in ("mpo_simhash_query", "mpo_simhash_keys", "mpo_lsh_fill",
                   "mpo_lsh_batch_retrieve", "mpo_sparse_attention", "mpo_full_attention",
                   "mpo_merge_state"):
            getattr(L, fn).restype = None
        del f32
        _lib = L
        return L


# ------------------------------------------------------------------ array plumbing

def _is_torch(x) -> bool:
    return type(x).__module__.startswith("torch")


def _np_view(x, dtype: str | None = None) -> np.ndarray:
    """A numpy array sharing memory with `x` (torch CPU tensor or ndarray); bf16 -> uint16."""
    if _is_torch(x):
        import torch

        assert x.device.type == "cpu", "the oracle only takes CPU tensors"
        assert x.is_contiguous(), "the reference casts raw data_ptr(): tensors must be contiguous"
        if x.dtype == torch.bfloat16:
            a = x.view(torch.int16).numpy().view(np.uint16)
        else:
            a = x.numpy()
    else:
        a = x
        assert a.flags["C_CONTIGUOUS"]
    if dtype is not None:
        assert a.dtype == np.dtype(dtype), f"expected {dtype}, got {a.dtype}"
    return a


def _ptr(a: np.ndarray) -> C.c_void_p:
    return C.c_void_p(a.ctypes.data)


def bf16_bits(x) -> np.ndarray:
    """uint16 bit patterns of a bf16 torch tensor / f32 array rounded RNE (torch semantics)."""
    if _is_torch(x):
        import torch

        if x.dtype != torch.bfloat16:
            x = x.to(torch.bfloat16)
        return _np_view(x.contiguous()).copy()
    a = np.ascontiguousarray(x)
    if a.dtype == np.uint16:
        return a
    a = a.astype(np.float32)
    out = np.empty(a.shape, np.uint16)
    lib().mpo_f32_to_bf16_rne(_ptr(a), _ptr(out), C.c_int64(a.size))
    return out


def bf16_to_f32(bits: np.ndarray) -> np.ndarray:
    return (bits.astype(np.uint32) << 16).view(np.float32)


# ------------------------------------------------------------------ free functions

def simhash_query(q, hash_func, K: int, L: int):
    """models/attnserver.py:264-270.  q bf16 [R,D], hash_func bf16 [D,K*L] ->
    (codes int32 [R,L], qnorm f32 [R])."""
    qb = bf16_bits(q).reshape(-1, bf16_bits(q).shape[-1])
    wb = bf16_bits(hash_func)
    R, D = qb.shape
    assert wb.shape == (D, K * L)
    codes = np.zeros((R, L), np.int32)
    qn = np.zeros((R,), np.float32)
    lib().mpo_simhash_query(_ptr(qb), _ptr(wb), R, D, K, L, _ptr(codes), _ptr(qn))
    return codes, qn


def simhash_keys(keys, hash_func, K: int, L: int) -> np.ndarray:
    """models/attnserver.py:159-168.  keys bf16 [Hkv,n,D] -> int16 [Hkv,L,n]."""
    kb = bf16_bits(keys)
    wb = bf16_bits(hash_func)
    Hkv, n, D = kb.shape
    codes = np.zeros((Hkv, L, n), np.int16)
    for i in range(Hkv):
        lib().mpo_simhash_keys(_ptr(kb[i]), _ptr(wb), n, D, K, L, _ptr(codes[i]))
    return codes


def centre_keys(key_cache, value_cache, seq_len: int, num_sink: int, num_local: int):
    """models/attnserver.py:136-148 for one request, as DEFINED for the HIP path (csrc/attention.hip,
    key_centre_fill_kernel): key_cache / value_cache bf16 [>= seq_len, Hkv, D] ->
        avg_k   bf16 bits [Hkv, D]    mean over the offloaded tokens [num_sink, seq_len - num_local): exact sum
                                      (f64 of bf16 addends), rounded f64 -> f32 -> bf16 (RNE)
        keys    bf16 bits [Hkv, n, D] bf16(f32(k) - f32(avg_k)) (RNE)
        values  bf16 bits [Hkv, n, D]
        kn      f32 [Hkv, n]          sqrt of the exact sum of squares of the centred key, f64 -> f32 -> bf16
    torch evaluates the same three lines with f32 sums in its own order: equal except where the exact value sits
    within ~1e-6 relative of a bf16 rounding boundary (tests/golden/fill_centre.npz lists those positions)."""
    kb = bf16_bits(key_cache)[num_sink:seq_len - num_local]
    vb = bf16_bits(value_cache)[num_sink:seq_len - num_local]
    kf = bf16_to_f32(kb).astype(np.float64)                                   # [n, Hkv, D]
    avg32 = (kf.sum(axis=0) / kf.shape[0]).astype(np.float32)
    avg = bf16_bits(avg32)                                                    # RNE
    cen = bf16_bits((bf16_to_f32(kb) - bf16_to_f32(avg)[None]).astype(np.float32))
    cf = bf16_to_f32(cen).astype(np.float64)
    kn = bf16_to_f32(bf16_bits(np.sqrt((cf * cf).sum(-1)).astype(np.float32)))
    return (avg, np.ascontiguousarray(cen.transpose(1, 0, 2)), np.ascontiguousarray(vb.transpose(1, 0, 2)),
            np.ascontiguousarray(kn.T))


def merge_state(va, sa, vb, sb):
    """flashinfer.merge_state restated (models/attnserver.py:308; pinned by tests/golden/window_merge.npz)."""
    a = bf16_bits(va)
    b = bf16_bits(vb)
    R, D = a.reshape(-1, a.shape[-1]).shape
    sa_ = np.ascontiguousarray(_np_view(sa), np.float32).reshape(R)
    sb_ = np.ascontiguousarray(_np_view(sb), np.float32).reshape(R)
    v = np.zeros((R, D), np.uint16)
    s = np.zeros((R,), np.float32)
    lib().mpo_merge_state(_ptr(a), _ptr(sa_), _ptr(b), _ptr(sb_), R, D, _ptr(v), _ptr(s))
    return v, s


# ------------------------------------------------------------------ LSH

class LSH:
    """Mirror of the reference `lsh.LSH` (library/lsh/lsh.h:14-43)."""

    def __init__(self, nthreads: int = 0):
        self.allocated = False
        self.nthreads = nthreads

    def alloc(self, K, L, num_layers, num_attention_heads, num_key_value_heads, batch_size,
              max_length):
        # library/lsh/lsh.cc:44-91
        self.K, self.L = K, L
        self.NB = 1 << K
        self.num_layers = num_layers
        self.H, self.Hkv, self.B, self.M = (num_attention_heads, num_key_value_heads,
                                            batch_size, max_length)
        self.G = self.H // self.Hkv
        shp = (self.B * self.Hkv, L, self.NB)
        self.table_start = [np.zeros(shp, np.int32) for _ in range(num_layers)]
        self.table_end = [np.zeros(shp, np.int32) for _ in range(num_layers)]
        self.table = [np.zeros((self.B * self.Hkv, L, self.M), np.int32)
                      for _ in range(num_layers)]
        self.mask = np.zeros((self.B * self.H, self.M), np.uint8)
        self.allocated = True

    def fill(self, layer_id, request_id, sorted_hash_code, sorted_indices):
        # library/lsh/lsh.cc:143-201
        codes = _np_view(sorted_hash_code, "int16")
        ids = _np_view(sorted_indices, "int32")
        assert codes.shape[0] == self.Hkv and codes.shape[1] == self.L
        n = codes.shape[2]
        assert n <= self.M and ids.shape == codes.shape
        s = slice(request_id * self.Hkv, (request_id + 1) * self.Hkv)
        lib().mpo_lsh_fill(_ptr(codes), _ptr(ids), self.Hkv, self.L, n, self.NB, self.M,
                           _ptr(self.table_start[layer_id][s]),
                           _ptr(self.table_end[layer_id][s]), _ptr(self.table[layer_id][s]))

    def batch_retrieve(self, layer_id, query, results, nnz):
        # library/lsh/lsh.cc:210-241
        q = _np_view(query, "int32")
        r = _np_view(results, "int32")
        z = _np_view(nnz, "int32")
        BH = self.B * self.H
        assert q.shape == (BH, self.L) and r.shape == (BH, self.M) and z.shape == (BH,)
        assert q.min() >= 0 and q.max() < self.NB
        lib().mpo_lsh_batch_retrieve(_ptr(self.table_start[layer_id]),
                                     _ptr(self.table_end[layer_id]), _ptr(self.table[layer_id]),
                                     _ptr(q), BH, self.G, self.L, self.NB, self.M, _ptr(r),
                                     _ptr(z), _ptr(self.mask), self.nthreads)

    def clear(self):
        # library/lsh/lsh.cc:293-306
        for i in range(self.num_layers):
            self.table_start[i][...] = 0
            self.table_end[i][...] = 0
            self.table[i][...] = 0
        self.mask[...] = 0

    def get_mask(self):
        # library/lsh/lsh.cc:308-314: int8 [B, H, M] view of the counters (0/1/2)
        return self.mask.view(np.int8).reshape(self.B, self.H, self.M)

    def copy(self, query):  # library/lsh/lsh.cc:203-207: empty in the reference
        return None


# ------------------------------------------------------------------ SparseAttentionServer

class SparseAttentionServer:
    """Mirror of `sparse_attention_cpu.SparseAttentionServer`
    (library/sparse_attention/sparse_attention.h:14-52)."""

    def __init__(self, nthreads: int = 0, exp_mode: int = 0, clamp_cos: int = 0):
        self.nthreads = nthreads
        self.exp_mode = exp_mode      # bit 0: reference polynomial exp; bit 1: cancellation-free f64 weight
        self.clamp_cos = clamp_cos    # reference: no clamp (sparse_attention.cc:177)
        self.allocated = False

    def alloc(self, num_layers, num_attention_heads, num_key_value_heads, head_dim, batch_size,
              max_length):
        # library/sparse_attention/sparse_attention.cc:546-583
        self.num_layers = num_layers
        self.H, self.Hkv, self.D, self.B, self.M = (num_attention_heads, num_key_value_heads,
                                                    head_dim, batch_size, max_length)
        self.G = self.H // self.Hkv
        kv = (self.B * self.Hkv, self.M, self.D)
        self.key_cache = [np.zeros(kv, np.uint16) for _ in range(num_layers)]
        self.value_cache = [np.zeros(kv, np.uint16) for _ in range(num_layers)]
        self.key_norm = [np.zeros(kv[:2], np.float32) for _ in range(num_layers)]
        self.attention_score = np.zeros((self.B * self.H, self.M), np.float32)
        self.allocated = True

    def fill(self, layer_id, request_id, k, v, kn):
        # library/sparse_attention/sparse_attention.cc:601-627
        kb, vb = bf16_bits(k), bf16_bits(v)
        knf = np.ascontiguousarray(_np_view(kn), np.float32)
        n = kb.shape[1]
        assert kb.shape == (self.Hkv, n, self.D) and vb.shape == kb.shape
        assert knf.shape == (self.Hkv, n) and n <= self.M
        s = slice(request_id * self.Hkv, (request_id + 1) * self.Hkv)
        self.key_cache[layer_id][s, :n] = kb
        self.value_cache[layer_id][s, :n] = vb
        self.key_norm[layer_id][s, :n] = knf

    def _sparse(self, layer_id, K, L, output, max_value_expsum, query, query_norm, ind, nnz):
        BH = self.B * self.H
        out = _np_view(output)
        assert out.dtype == np.uint16 and out.shape == (BH, self.D)
        mve = _np_view(max_value_expsum, "float32")
        assert mve.shape == (2, BH)
        if _is_torch(query):
            import torch

            q_is_bf16 = query.dtype == torch.bfloat16
            q = _np_view(query.contiguous() if q_is_bf16 else query.float().contiguous())
        else:
            q = np.ascontiguousarray(query)
            q_is_bf16 = q.dtype == np.uint16
            if not q_is_bf16:
                q = q.astype(np.float32)
        assert q.size == BH * self.D
        qn = np.ascontiguousarray(_np_view(query_norm), np.float32).reshape(BH)
        idx = _np_view(ind, "int32")
        z = _np_view(nnz, "int32")
        assert idx.shape == (BH, self.M) and z.shape == (BH,)
        lib().mpo_sparse_attention(_ptr(self.key_cache[layer_id]),
                                   _ptr(self.value_cache[layer_id]),
                                   _ptr(self.key_norm[layer_id]), _ptr(q), int(q_is_bf16),
                                   _ptr(qn), _ptr(idx), _ptr(z), BH, self.G, self.D, self.M,
                                   K, L, self.exp_mode, self.clamp_cos, _ptr(out), _ptr(mve),
                                   _ptr(self.attention_score), self.nthreads)

    # sparse_attention.cc:629-745 -- every dispatch target computes the same function
    attention_wrapper = _sparse
    attention = _sparse
    scheduled_attention = _sparse
    attention_bf16 = _sparse
    attention_wrapper_bf16 = _sparse

    def full_attention(self, layer_id, output, max_value_expsum, query, nnz, quirks: int = 0):
        # library/sparse_attention/sparse_attention.cc:988-1037; quirks: see mp_oracle.c (bit 0 the
        # reference's polynomial exp, bit 1 its 16-slot softmax tail); 0 = the definition
        BH = self.B * self.H
        out = _np_view(output)
        mve = _np_view(max_value_expsum, "float32")
        if _is_torch(query):
            q = query.float().contiguous().numpy()
        else:
            q = np.ascontiguousarray(query, np.float32)
        z = _np_view(nnz, "int32")
        lib().mpo_full_attention(_ptr(self.key_cache[layer_id]),
                                 _ptr(self.value_cache[layer_id]), _ptr(q), _ptr(z), BH,
                                 self.G, self.D, self.M, _ptr(out), _ptr(mve),
                                 _ptr(self.attention_score), int(quirks), self.nthreads)

    def clear(self):
        # library/sparse_attention/sparse_attention.cc:586-598
        for i in range(self.num_layers):
            self.key_cache[i][...] = 0
            self.value_cache[i][...] = 0
            self.key_norm[i][...] = 0
        self.attention_score[...] = 0

    def get_key_cache(self, layer_id):
        return self.key_cache[layer_id].reshape(self.B, self.Hkv, self.M, self.D)

    def get_value_cache(self, layer_id):
        return self.value_cache[layer_id].reshape(self.B, self.Hkv, self.M, self.D)

    def get_key_norm(self, layer_id):
        return self.key_norm[layer_id].reshape(self.B, self.Hkv, self.M)

    def get_score(self):
        return self.attention_score.reshape(self.B, self.H, self.M)
