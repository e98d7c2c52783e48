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


# ---- the `clustered` workload (bench.py --data clustered): keys as a model's KV cache holds them
# after RoPE -- a per-dimension offset (removed by the centring), an anisotropic spectrum (per-dimension scale
# s_d ~ (1 + d)^-alpha; SimHash with Gaussian planes is rotation invariant, so a diagonal spectrum is general), a
# mixture of `clusters` centres with unequal populations, and a low-rank component shared by all tokens.  Element-wise
# float64 arithmetic in a fixed order only (no matmul): identical bits on every machine.
CLUSTERED = dict(alpha=0.2, a=0.5, b=0.2, clusters=64, rank=4)    # cfg 1 with heavy-hitter queries selects ~2.1 % (README.md:43: ~2 %)
SKEWED = dict(alpha=0.5, a=1.0, b=0.5, clusters=64, rank=4)       # stress: ~8 % selected, 3.7 % of the probed pieces > 126 ids


def clustered_raw_bits(seed: int, Hkv: int, n: int, D: int, alpha=None, a=None, b=None, clusters=None, rank=None):
    p = dict(CLUSTERED)
    for k, v in (("alpha", alpha), ("a", a), ("b", b), ("clusters", clusters), ("rank", rank)):
        if v is not None:
            p[k] = v
    s = (1.0 + np.arange(D, dtype=np.float64)) ** (-p["alpha"])
    s = s / np.sqrt((s * s).mean())
    z = normal_f32(seed, (Hkv, n, D)).astype(np.float64)
    C = normal_f32(seed + 101, (Hkv, p["clusters"], D)).astype(np.float64)
    u1 = randint(seed + 102, 0, p["clusters"], (Hkv, n))
    u2 = randint(seed + 106, 0, p["clusters"], (Hkv, n))
    cid = np.minimum(u1, u2)                                   # populations fall linearly from cluster 0 to the last
    U = normal_f32(seed + 103, (Hkv, p["rank"], D)).astype(np.float64)
    w = normal_f32(seed + 104, (Hkv, n, p["rank"])).astype(np.float64)
    off = normal_f32(seed + 105, (Hkv, 1, D)).astype(np.float64) * 0.75
    raw = s * z
    raw = raw + p["a"] * (s * np.take_along_axis(C, cid[:, :, None], axis=1))
    for j in range(p["rank"]):
        raw = raw + p["b"] * (w[:, :, j:j + 1] * (s * U[:, j:j + 1, :]))
    # unit mean square per element like the isotropic workload (SimHash does not see the scale; the logits do)
    raw = raw * (1.0 / np.sqrt(1.0 + p["a"] ** 2 + p["rank"] * p["b"] ** 2)) + off
    return f32_to_bf16_bits(raw.astype(np.float32))


def centre_bits(raw_bits: np.ndarray):
    """models/attnserver.py:139-146 on bf16 keys [Hkv, n, D], with exact (f64) sums: (centred key bits, kn f32)."""
    raw = bf16_bits_to_f32(raw_bits).astype(np.float64)
    avg = bf16_bits_to_f32(f32_to_bf16_bits(raw.mean(axis=1, keepdims=True).astype(np.float32)))
    k = f32_to_bf16_bits((raw.astype(np.float32) - avg).astype(np.float32))
    kf = bf16_bits_to_f32(k).astype(np.float64)
    kn = np.sqrt((kf * kf).sum(-1)).astype(np.float32)
    return k, bf16_bits_to_f32(f32_to_bf16_bits(kn))


def clustered_keys(seed: int, Hkv: int, n: int, D: int, **kw):
    """Centred clustered keys: (key bits uint16 [Hkv, n, D], kn f32 [Hkv, n])."""
    return centre_bits(clustered_raw_bits(seed, Hkv, n, D, **kw))
