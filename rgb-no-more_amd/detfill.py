"""Deterministic, platform-independent tensor fills used by golden vectors, parity tests and bench.

Golden logits (tests/golden/) were produced by loading these fills into the *reference* ViT in the
survey container; the GPU box regenerates bit-identical weights/inputs from (name, shape) alone, so no
22 MB weight file has to travel.  Pure integer hashing in numpy (no RNG library state involved).
"""
import zlib
import numpy as np

_M32 = np.uint64(0xFFFFFFFF)


def _mix(x):
    # 32-bit finaliser of murmur3, vectorised on uint64 lanes masked to 32 bits
    x = x & _M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x85EBCA6B)) & _M32
    x ^= x >> np.uint64(13)
    x = (x * np.uint64(0xC2B2AE35)) & _M32
    x ^= x >> np.uint64(16)
    return x


def uniform(shape, seed, lo=-1.0, hi=1.0, dtype=np.float32):
    """uniform in [lo, hi) from a counter hash; value i depends only on (seed, i)."""
    n = int(np.prod(shape)) if len(shape) else 1
    idx = np.arange(n, dtype=np.uint64)
    s = np.uint64((int(seed) * 0x9E3779B1) & 0xFFFFFFFF)
    h = _mix(idx * np.uint64(0x9E3779B1) + s)
    h = _mix(h ^ s)
    u = h.astype(np.float64) / 4294967296.0
    return (lo + (hi - lo) * u).astype(dtype).reshape(shape)


def normalish(shape, seed, dtype=np.float32):
    """approximately N(0,1): sum of 4 uniforms (Irwin-Hall), deterministic."""
    acc = np.zeros(int(np.prod(shape)), dtype=np.float64)
    for k in range(4):
        acc += uniform((acc.size,), seed * 4 + k + 1, -1.0, 1.0, np.float64)
    return (acc * (3.0 / 4.0) ** 0.5).astype(dtype).reshape(shape)


def integers(shape, seed, lo, hi, dtype=np.int16):
    """integers in [lo, hi] inclusive."""
    u = uniform(shape, seed, 0.0, 1.0, np.float64)
    return (lo + np.floor(u * (hi - lo + 1))).clip(lo, hi).astype(dtype)


def name_seed(name: str) -> int:
    return zlib.crc32(name.encode()) & 0x7FFFFFFF


def fill_state_dict(shapes: dict, base_seed=0):
    """shapes: {param_name: shape}.  Linear-like init: U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for >=2-D
    weights and biases alike (fan_in = last dim of the weight; biases use their own length),
    LayerNorm weights ('lrnorm' in name and name ends with .weight) = 1 + U(-0.1, 0.1)."""
    out = {}
    for name, shp in shapes.items():
        sd = name_seed(name) ^ (base_seed * 7919)
        shp = tuple(shp)
        if "lrnorm" in name and name.endswith(".weight"):
            out[name] = 1.0 + uniform(shp, sd, -0.1, 0.1)
        elif "lrnorm" in name:
            out[name] = uniform(shp, sd, -0.1, 0.1)
        else:
            fan = shp[-1] if len(shp) >= 2 else shp[0]
            a = 1.0 / np.sqrt(float(fan))
            out[name] = uniform(shp, sd, -a, a)
    return out


def fill_swin_params(shapes: dict, base_seed=3):
    """Deterministic SwinV2 test weights for goldens, parity tests and bench.py's parity check: the Linear-like fill above,
    LayerNorm scales 1 + U(-a, a) (so the post-norm branches are O(1), not O(0.1)), logit_scale = ln 10 + U(-.5a, .5a)
    (the trained range)."""
    sd = fill_state_dict(shapes, base_seed=base_seed)
    for n in shapes:
        if n.endswith("logit_scale"):
            sd[n] = (np.log(10.0) + 0.5 * sd[n]).astype(np.float32)
        elif "norm" in n and n.endswith(".weight"):
            sd[n] = (1.0 + sd[n]).astype(np.float32)
    return sd
