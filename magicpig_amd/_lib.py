"""ctypes loader of the gfx950 C-ABI library (include/magicpig_hip.h).

There is NO CPU fallback: if the library is missing or a call fails, this module raises.
torch is imported first on purpose: the torch ROCm wheel bundles its own libamdhip64.so.7, and
loading it first makes this library bind to the SAME HIP runtime instance (same SONAME), so
torch tensors, torch streams and our kernels share one device context.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must precede the CDLL below, see module docstring)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libmagicpig_hip.so")

MP_OK = 0
ERR_UNSUPPORTED = 5
MEM_HOST, MEM_DEVICE = 0, 1
DTYPE_BF16, DTYPE_F32 = 0, 1
DECODE_NO_BYPRODUCTS = 1      # mp_decode_*_ex flag (include/magicpig_hip.h)

_ERR_NAMES = {1: "MP_ERR_INVALID", 2: "MP_ERR_STATE", 3: "MP_ERR_HIP", 4: "MP_ERR_NOMEM",
              5: "MP_ERR_UNSUPPORTED", 6: "MP_ERR_DATA"}


class MagicPigError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"{_ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m magicpig_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the product path.")
    L = C.CDLL(LIB_PATH)
    i32, i64, p = C.c_int, C.c_int64, C.c_void_p
    pp = C.POINTER(C.c_void_p)
    sig = {
        "mp_version": ([], i32),
        "mp_last_error": ([], C.c_char_p),
        "mp_arch": ([], C.c_char_p),
        "mp_simhash_create": ([pp], i32),
        "mp_simhash_destroy": ([p], i32),
        "mp_simhash_set_planes": ([p, i32, i32, i32, p, i32, p], i32),
        "mp_simhash_query": ([p, p, i32, p, p, i32, p], i32),
        "mp_simhash_keys": ([p, p, i32, i64, p, i32, p], i32),
        "mp_simhash_debug_acc": ([p, p], i32),
        "mp_lsh_create": ([pp], i32),
        "mp_lsh_destroy": ([p], i32),
        "mp_lsh_alloc": ([p, i32, i32, i32, i32, i32, i32, i32], i32),
        "mp_lsh_alloc_ex": ([p, i32, i32, i32, i32, i32, i32, i32, i64, i32], i32),
        "mp_lsh_get_footprint_ex": ([p, C.POINTER(i64)], i32),
        "mp_lsh_fill": ([p, i32, i32, p, p, i64, i32, p], i32),
        "mp_lsh_build": ([p, i32, i32, p, i64, i32, p], i32),
        "mp_lsh_build_with_norms": ([p, p, i32, i32, p, i64, i32, p], i32),
        "mp_lsh_batch_retrieve": ([p, i32, p, p, p, i32, p], i32),
        "mp_lsh_clear": ([p, p], i32),
        "mp_lsh_get_mask": ([p, p, i32, p], i32),
        "mp_lsh_get_tables": ([p, i32, pp, pp], i32),
        "mp_lsh_get_ranges": ([p, C.POINTER(i32), C.POINTER(i32)], i32),
        "mp_lsh_get_id_bits": ([p, i32, C.POINTER(i32)], i32),
        "mp_lsh_get_footprint": ([p, C.POINTER(i64)], i32),
        "mp_attn_get_footprint": ([p, C.POINTER(i64)], i32),
        "mp_attn_create": ([pp], i32),
        "mp_attn_destroy": ([p], i32),
        "mp_attn_alloc": ([p, i32, i32, i32, i32, i32, i32], i32),
        "mp_attn_fill": ([p, i32, i32, p, p, p, i64, i32, p], i32),
        "mp_attn_fill_offload": ([p, p, i32, i32, p, p, i64, i32, i32, p, p, p], i32),
        "mp_attn_sparse": ([p, i32, i32, i32, p, p, p, i32, p, p, p, i32, p], i32),
        "mp_attn_full": ([p, i32, p, p, p, i32, p, i32, p], i32),
        "mp_attn_clear": ([p, p], i32),
        "mp_attn_append": ([p, i32, p, p, p, p], i32),
        "mp_attn_append_centred": ([p, i32, p, p, p, p, i32, p], i32),
        "mp_attn_check": ([p, p], i32),
        "mp_debug_set_stamp_buffer": ([p], i32),
        "mp_debug_xcd_round_robin": ([], i32),
        "mp_debug_set_option": ([C.c_char_p, i32], i32),
        "mp_debug_get_option": ([C.c_char_p, C.POINTER(i32)], i32),
        "mp_attn_get_kv": ([p, i32, pp, pp, C.POINTER(i64)], i32),
        "mp_attn_get_key_norm": ([p, i32, pp], i32),
        "mp_attn_get_score": ([p, pp, p], i32),
        "mp_attn_invalidate_norms": ([p, i32, i32, p], i32),
        "mp_decode_sparse_layer": ([p, p, p, i32, p, p, p, p, p], i32),
        "mp_decode_layer_window": ([p, p, p, p, i32, p, p, p, p, p, p], i32),
        "mp_decode_sparse_layer_ex": ([p, p, p, i32, p, p, p, p, C.c_uint, p], i32),
        "mp_decode_layer_window_ex": ([p, p, p, p, i32, p, p, p, p, p, C.c_uint, p], i32),
        "mp_merge_state": ([p, p, p, p, i32, i32, p, p, p], i32),
    }
    for name, (args, res) in sig.items():
        fn = getattr(L, name)  # AttributeError here = the library does not export the ABI
        fn.argtypes = args
        fn.restype = res
    _lib = L
    return L


def check(rc: int) -> None:
    if rc != MP_OK:
        raise MagicPigError(rc, lib().mp_last_error().decode("utf-8", "replace"))


def set_option(name: str, value: int) -> None:
    """A/B switches of the library (include/magicpig_hip.h: mp_debug_set_option)."""
    check(lib().mp_debug_set_option(name.encode(), int(value)))


def get_option(name: str) -> int:
    v = C.c_int(0)
    check(lib().mp_debug_get_option(name.encode(), C.byref(v)))
    return v.value


# ---------------------------------------------------------------- tensor plumbing

def mem_kind(t: torch.Tensor) -> int:
    return MEM_DEVICE if t.is_cuda else MEM_HOST


def ptr(t):
    """Address of a tensor's storage as ctypes takes it for a c_void_p argument (a plain int: the per-call cost of
    this layer is part of the host-buffer mode's budget, EXPERIMENTS.md R4-5), None for an absent tensor."""
    if t is None:
        return None
    return t.data_ptr()


def current_stream(ref: torch.Tensor | None = None, device: int | None = None) -> C.c_void_p:
    """hipStream_t of torch's current stream (so torch events / graphs see our launches) on the device of
    `ref` when it is a CUDA tensor, else on `device` (the handle's own device), else on the current one."""
    if torch.cuda.is_available():
        dev = ref.device if (ref is not None and ref.is_cuda) else device
        if _raw_stream is not None:
            if dev is None:
                idx = torch.cuda.current_device()
            elif isinstance(dev, int):
                idx = dev
            else:
                idx = dev.index if dev.index is not None else torch.cuda.current_device()
            return _raw_stream(idx)
        return torch.cuda.current_stream(dev).cuda_stream
    return None


# torch's raw current-stream lookup (one C call; torch.cuda.current_stream builds a Stream object per call)
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def current_device() -> int:
    """Index of the current GPU (0 when there is none: the C ABI then fails with MP_ERR_HIP)."""
    return torch.cuda.current_device() if torch.cuda.is_available() else 0


def expect(t: torch.Tensor, dtype, shape, name: str, same_numel_ok: bool = False) -> torch.Tensor:
    """The reference casts raw data_ptr() with no checks; we check instead: dtype, the
    EXACT shape and contiguity.  same_numel_ok is for the one argument the reference's callers pass in two
    layouts of the same memory (the query: [B*H, D] at models/attnserver.py:274, [B, H, 1, D] in
    library/sparse_attention/test.py:44)."""
    try:      # the common case first: three attribute reads
        if (dtype is None or t.dtype is dtype) and t.shape == shape and t.is_contiguous():
            return t
    except AttributeError:
        pass
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name}: expected a torch.Tensor")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if shape is not None and tuple(t.shape) != tuple(shape):
        if not (same_numel_ok and t.numel() == int(torch.Size(shape).numel())):
            raise ValueError(f"{name}: expected shape {tuple(shape)}, got {tuple(t.shape)}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")
    return t


def same_memory(*tensors) -> int:
    kind = -1
    for t in tensors:
        if t is None:
            continue
        k = MEM_DEVICE if t.is_cuda else MEM_HOST
        if kind < 0:
            kind = k
        elif k != kind:
            raise ValueError("all tensors of one call must live on the same side (all CPU or all on the GPU)")
    if kind < 0:
        raise ValueError("all tensors of one call must live on the same side (all CPU or all on the GPU)")
    return kind


class DeviceView:
    """Zero-copy torch view of handle-owned HBM through __cuda_array_interface__."""

    def __init__(self, address: int, shape, typestr: str, strides=None):
        self.__cuda_array_interface__ = {
            "shape": tuple(int(s) for s in shape), "typestr": typestr, "data": (int(address), False),
            "version": 3, "strides": None if strides is None else tuple(int(s) for s in strides),
        }


def device_tensor(address: int, shape, typestr: str, strides_bytes=None, device=None) -> torch.Tensor:
    dev = device if device is not None else torch.device("cuda", torch.cuda.current_device())
    return torch.as_tensor(DeviceView(address, shape, typestr, strides_bytes), device=dev)
