"""Host-side counter-based PRNG compatible with jax.random (threefry2x32), numpy only.

The reference draws the coarse jitter with `random.randint(key, [N_c], 0, P)` (rnerf/models.py:241-242) even in eval
mode, and the stratified fine draws with `random.uniform` (rnerf/model_utils.py:349-352).  This module restates
jax 0.2.22's `PRNGKey`, `split`, `_random_bits`, `randint` and `uniform` on top of the Threefry-2x32-20 block cipher.
The cipher is pinned by the Random123 known-answer vector (tests/test_prng.py); the reductions around it follow the
jax 0.2.22 source from memory and cannot be checked against JAX offline (SURVEY.md §8c KAT 10) — which is why the
integer jitter and the uniform draws are also injectable explicitly into `NerfModel.apply`.
"""
from __future__ import annotations

import numpy as np

_ROT = ((13, 15, 26, 6), (17, 29, 16, 24))


def _rotl(x, r):
    return ((x << np.uint32(r)) | (x >> np.uint32(32 - r))).astype(np.uint32)


def threefry2x32(k0, k1, x0, x1):
    """Threefry-2x32 with 20 rounds (Random123).  All arguments uint32 (scalars or arrays)."""
    with np.errstate(over="ignore"):
        k0 = np.uint32(k0); k1 = np.uint32(k1)
        ks = (k0, k1, np.uint32(k0 ^ k1 ^ np.uint32(0x1BD11BDA)))
        x0 = (np.asarray(x0, np.uint32) + ks[0]).astype(np.uint32)
        x1 = (np.asarray(x1, np.uint32) + ks[1]).astype(np.uint32)
        for i in range(5):
            for r in _ROT[i % 2]:
                x0 = (x0 + x1).astype(np.uint32)
                x1 = _rotl(x1, r)
                x1 = (x1 ^ x0).astype(np.uint32)
            x0 = (x0 + ks[(i + 1) % 3]).astype(np.uint32)
            x1 = (x1 + ks[(i + 2) % 3] + np.uint32(i + 1)).astype(np.uint32)
    return x0, x1


def PRNGKey(seed: int) -> np.ndarray:
    seed = int(seed)
    return np.array([(seed >> 32) & 0xFFFFFFFF, seed & 0xFFFFFFFF], np.uint32)


def _threefry_2x32(key, count):
    count = np.asarray(count, np.uint32).ravel()
    odd = count.size % 2
    if odd:
        count = np.concatenate([count, np.zeros(1, np.uint32)])
    half = count.size // 2
    y0, y1 = threefry2x32(key[0], key[1], count[:half], count[half:])
    out = np.concatenate([y0, y1])
    return out[:-1] if odd else out


def split(key, num: int = 2) -> np.ndarray:
    return _threefry_2x32(key, np.arange(num * 2, dtype=np.uint32)).reshape(num, 2)


def random_bits(key, shape) -> np.ndarray:
    size = int(np.prod(shape)) if len(shape) else 1
    return _threefry_2x32(key, np.arange(size, dtype=np.uint32)).reshape(shape)


def randint(key, shape, minval: int, maxval: int) -> np.ndarray:
    """jax.random.randint for int32 ranges."""
    k1, k2 = split(key)
    hi = random_bits(k1, shape).astype(np.uint64)
    lo = random_bits(k2, shape).astype(np.uint64)
    span = np.uint64(max(int(maxval) - int(minval), 1))
    m32 = np.uint64(0xFFFFFFFF)
    mult = (np.uint64(2 ** 16) % span)
    mult = ((mult * mult) & m32) % span
    off = (((hi % span) * mult) & m32) + (lo % span)
    off = (off & m32) % span
    return (np.int64(minval) + off.astype(np.int64)).astype(np.int32)


def uniform(key, shape, minval: float = 0.0, maxval: float = 1.0) -> np.ndarray:
    """jax.random.uniform for float32."""
    bits = random_bits(key, shape)
    fl = ((bits >> np.uint32(9)) | np.uint32(0x3F800000)).view(np.float32) - np.float32(1.0)
    mn, mx = np.float32(minval), np.float32(maxval)
    return np.maximum(mn, fl * (mx - mn) + mn).astype(np.float32)
