"""TEST INFRASTRUCTURE — a second, independently written restatement of the jax.random functions the path uses, in plain Python integers.

The product's host PRNG (samplenerfro_amd/prng.py, numpy uint32 arithmetic, vectorised) and the device kernels (csrc/pipeline.hip) are one
reading of jax 0.2.22's `jax/_src/random.py`; this file is another, written separately, one draw at a time, with Python's unbounded integers
and explicit `% 2**32` wraps — so that a slip in the vectorised arithmetic (a missed wrap, a wrong counter half, an off-by-one in the key
schedule) shows as a disagreement (tests/test_prng.py compares them on 10^6 draws).  What it restates, by function:

  threefry2x32        Random123 Threefry-2x32, 20 rounds, rotation constants (13, 15, 26, 6 | 17, 29, 16, 24), key-schedule parity constant
                      0x1BD11BDA (Salmon et al., SC'11; jax/_src/prng.py `threefry2x32_p`): pinned by the Random123 known-answer vectors.
  bits(key, n)        `_random_bits` / `threefry_random_bits` for 32-bit words: counters 0..n-1 (padded to even), the FIRST half of the counters
                      feeds word 0 of the cipher, the second half word 1, outputs concatenated.
  split(key, num)     `_split`: bits(key, 2 num) reshaped to [num, 2].
  uniform             `_uniform` float32: (bits >> 9) | 0x3F800000 reinterpreted as float32, minus 1, scaled into [minval, maxval), max(minval, .).
  randint             `_randint` int32: two independent 32-bit words per draw (keys split once), span = maxval - minval as uint32,
                      multiplier = ((2^16 % span)^2 mod 2^32) % span, offset = ((hi % span) * multiplier mod 2^32 + lo % span) mod 2^32 % span.
                      No published vector exists for this reduction (SURVEY.md 8c KAT 10): two independent readings agreeing is the most
                      that can be had offline.  Used by the reference at rnerf/models.py:241-242 (coarse jitter) — and injectable there.
Nothing under samplenerfro_amd/ imports this module.
"""
import struct

M32 = 1 << 32
_R0, _R1 = (13, 15, 26, 6), (17, 29, 16, 24)


def _rotl(x, r):
    return ((x << r) % M32) | (x >> (32 - r))


def threefry2x32(k0, k1, c0, c1):
    ks = (k0, k1, k0 ^ k1 ^ 0x1BD11BDA)
    x0, x1 = (c0 + ks[0]) % M32, (c1 + ks[1]) % M32
    for block in range(5):
        for r in (_R0 if block % 2 == 0 else _R1):
            x0 = (x0 + x1) % M32
            x1 = _rotl(x1, r) ^ x0
        x0 = (x0 + ks[(block + 1) % 3]) % M32
        x1 = (x1 + ks[(block + 2) % 3] + block + 1) % M32
    return x0, x1


def bits(key, n):
    """n 32-bit words of the stream of `key` (a pair of ints)."""
    half = (n + 1) // 2
    first, second = [], []
    for i in range(half):
        j = half + i
        a, b = threefry2x32(key[0], key[1], i, j if j < n else 0)
        first.append(a)
        second.append(b)
    return (first + second)[:n]


def split(key, num=2):
    w = bits(key, 2 * num)
    return [(w[2 * i], w[2 * i + 1]) for i in range(num)]


def _f32(x):
    return struct.unpack("<f", struct.pack("<f", x))[0]


def uniform(key, n, minval=0.0, maxval=1.0):
    mn, mx = _f32(minval), _f32(maxval)
    scale = _f32(mx - mn)
    out = []
    for w in bits(key, n):
        f = struct.unpack("<f", struct.pack("<I", (w >> 9) | 0x3F800000))[0] - 1.0        # exact in float32: both in [1, 2)
        v = _f32(_f32(f * scale) + mn)
        out.append(max(mn, v))
    return out


def randint(key, n, minval, maxval):
    k1, k2 = split(key, 2)
    hi, lo = bits(k1, n), bits(k2, n)
    span = (maxval - minval) % M32 if maxval > minval else 1
    mult = (1 << 16) % span
    mult = ((mult * mult) % M32) % span
    out = []
    for h, l in zip(hi, lo):
        off = (((h % span) * mult) % M32 + (l % span)) % M32
        out.append(minval + off % span)
    return out


def normal(key, n):
    """`_normal_real` float32: sqrt(2) * erf_inv(u), u = uniform(key, n, nextafter(-1, 0), 1) — through the exact normal quantile
    (sqrt(2) erf_inv(u) = Phi^-1((u + 1) / 2)) in double precision, rounded to float32.  XLA evaluates erf_inv by a float32 polynomial
    (samplenerfro_amd/prng.erf_inv_f32): this reading bounds it (tests/test_prng.py), it does not reproduce its last bits."""
    from statistics import NormalDist
    lo = struct.unpack("<f", struct.pack("<I", struct.unpack("<I", struct.pack("<f", -1.0))[0] - 1))[0]      # nextafter(-1, 0) in float32
    nd = NormalDist()
    out = []
    for u in uniform(key, n, lo, 1.0):
        out.append(_f32(nd.inv_cdf((u + 1.0) / 2.0)) if -1.0 < u < 1.0 else float("-inf"))
    return out
