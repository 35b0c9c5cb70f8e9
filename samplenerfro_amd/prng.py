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


_ERFINV_CENTRAL = (2.81022636e-08, 3.43273939e-07, -3.5233877e-06, -4.39150654e-06, 0.00021858087, -0.00125372503, -0.00417768164,
                   0.246640727, 1.50140941)
_ERFINV_TAIL = (-0.000200214257, 0.000100950558, 0.00134934322, -0.00367342844, 0.00573950773, -0.0076224613, 0.00943887047,
                1.00167406, 2.83297682)


def erf_inv_f32(x: np.ndarray) -> np.ndarray:
    """erf_inv as XLA evaluates it for float32: M. Giles' single-precision polynomial ("Approximating the erfinv function", 2010) in
    w = -log1p(-x*x): degree 8 in (w - 2.5) for w < 5, degree 8 in (sqrt(w) - 3) otherwise, times x — every operation rounded to float32.
    (jax.random.normal goes through lax.erf_inv; the exact inverse differs from this polynomial by up to ~3 ulp.)"""
    f = np.float32
    x = np.asarray(x, f)
    edge = np.abs(x) == f(1.0)
    w = (-np.log1p(-(np.where(edge, f(0), x) ** 2).astype(f))).astype(f)
    central = w < f(5.0)
    wc = (w - f(2.5)).astype(f)
    wt = (np.sqrt(np.maximum(w, f(0))).astype(f) - f(3.0)).astype(f)

    def horner(coef, t):
        p = np.full_like(t, f(coef[0]))
        for c in coef[1:]:
            p = (f(c) + (p * t).astype(f)).astype(f)
        return p

    p = np.where(central, horner(_ERFINV_CENTRAL, wc), horner(_ERFINV_TAIL, wt))
    return np.where(edge, np.copysign(f(np.inf), x), (p * x).astype(f)).astype(f)


def normal(key, shape) -> np.ndarray:
    """jax.random.normal for float32 (jax/_src/random.py `_normal_real`): sqrt(2) * erf_inv(uniform(key, shape, nextafter(-1, 0), 1)).
    Pinned by the two values jax's own documentation prints (PRNGKey(0) -> -0.20584226; the second key of split(PRNGKey(0)) -> -1.2515389;
    tests/test_prng.py).  Used by the raw-sigma regulariser (rnerf/model_utils.py:438-453)."""
    lo = np.nextafter(np.float32(-1.0), np.float32(0.0))
    u = uniform(key, shape, minval=lo, maxval=1.0)
    return (np.float32(np.sqrt(2.0)) * erf_inv_f32(u)).astype(np.float32)


def uniform(key, shape, minval: float = 0.0, maxval: float = 1.0) -> np.ndarray:
    """jax.random.uniform for float32."""
    bits = random_bits(key, shape)
    fl = ((bits >> np.uint32(9)) | np.uint32(0x3F800000)).view(np.float32) - np.float32(1.0)
    mn, mx = np.float32(minval), np.float32(maxval)
    return np.maximum(mn, fl * (mx - mn) + mn).astype(np.float32)
