"""Platform-independent synthetic inputs for the parity tests and golden fixtures.

torch.randn on CPU is vectorised differently per CPU family, so fixtures regenerate their
inputs from an integer-only generator instead: splitmix64 counters -> four 16-bit uniforms
summed (Irwin-Hall, near-normal, unit variance after scaling) -> exact float32 -> bf16 by
integer round-to-nearest-even.  Identical bits on every machine, numpy only.
"""
from __future__ import annotations

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def u64(seed: int, n: int) -> np.ndarray:
    ctr = np.arange(n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        base = _splitmix64(np.full(1, seed, np.uint64))[0]
        return _splitmix64(ctr ^ base)


def normal_f32(seed: int, shape) -> np.ndarray:
    """Near-normal (Irwin-Hall n=4), mean 0, variance 1; exactly representable steps."""
    n = int(np.prod(shape))
    r = u64(seed, n)
    s = np.zeros(n, np.int64)
    for k in range(4):
        s += ((r >> np.uint64(16 * k)) & np.uint64(0xFFFF)).astype(np.int64)
    # sum of 4 U{0..65535}: mean 131070, var 4*(65536^2-1)/12
    x = (s - 131070).astype(np.float32)  # exact
    scale = np.float32(1.0 / np.sqrt(4 * (65536.0 ** 2 - 1) / 12))
    return (x * scale).reshape(shape)


def f32_to_bf16_bits(x: np.ndarray) -> np.ndarray:
    u = np.ascontiguousarray(x, np.float32).view(np.uint32).astype(np.uint64)
    u = u + np.uint64(0x7FFF) + ((u >> np.uint64(16)) & np.uint64(1))
    return (u >> np.uint64(16)).astype(np.uint16)


def bf16_bits_to_f32(b: np.ndarray) -> np.ndarray:
    return (b.astype(np.uint32) << 16).view(np.float32)


def normal_bf16_bits(seed: int, shape) -> np.ndarray:
    return f32_to_bf16_bits(normal_f32(seed, shape))


def randint(seed: int, low: int, high: int, shape) -> np.ndarray:
    n = int(np.prod(shape))
    return (low + (u64(seed, n) % np.uint64(high - low)).astype(np.int64)).reshape(shape)


def to_torch_bf16(bits: np.ndarray):
    import torch

    return torch.from_numpy(bits.view(np.int16).copy()).view(torch.bfloat16)


def centred_keys(seed: int, Hkv: int, n: int, D: int):
    """Keys as models/attnserver.py:136-146 leaves them: centred per kv head over the
    offloaded tokens, rounded to bf16; kn = bf16 L2 norm widened to f32.
    Returns (key bits uint16 [Hkv,n,D], kn f32 [Hkv,n]) -- deterministic numpy arithmetic:
    the mean is taken in float64 and the norm from an exact-product float64 sum, so no
    platform-dependent reduction order enters the fixture inputs."""
    raw = bf16_bits_to_f32(normal_bf16_bits(seed, (Hkv, n, D))).astype(np.float64)
    avg = bf16_bits_to_f32(f32_to_bf16_bits(raw.mean(axis=1, keepdims=True).astype(np.float32)))
    k = f32_to_bf16_bits((raw.astype(np.float32) - avg).astype(np.float32))
    kf = bf16_bits_to_f32(k).astype(np.float64)
    kn = np.sqrt((kf * kf).sum(-1)).astype(np.float32)
    kn = bf16_bits_to_f32(f32_to_bf16_bits(kn))  # `.norm()` of a bf16 tensor is bf16, then .float()
    return k, kn
