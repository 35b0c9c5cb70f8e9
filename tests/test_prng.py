"""Host PRNG (samplenerfro_amd/prng.py): the Threefry-2x32-20 block cipher is pinned by the public Random123 vectors, the
key-derivation / bit-stream layout around it by the values JAX itself documents (jax.random module docstring, the "JAX 101:
pseudo random numbers" tutorial).  `randint`'s range reduction has no published vector and stays pinned only by its statistics."""
import numpy as np

from samplenerfro_amd import prng


def test_threefry_random123_kat():
    # Random123 kat_vectors: threefry2x32 20 rounds
    x0, x1 = prng.threefry2x32(0x13198a2e, 0x03707344, np.uint32(0x243f6a88), np.uint32(0x85a308d3))
    assert (int(x0), int(x1)) == (0xc4923a9c, 0x483df7a0)
    x0, x1 = prng.threefry2x32(0, 0, np.uint32(0), np.uint32(0))
    assert (int(x0), int(x1)) == (0x6b200159, 0x99ba4efe)
    x0, x1 = prng.threefry2x32(0xffffffff, 0xffffffff, np.uint32(0xffffffff), np.uint32(0xffffffff))
    assert (int(x0), int(x1)) == (0x1cb996fc, 0xbb002be7)


def test_split_and_key_shapes():
    k = prng.PRNGKey(20200823)
    assert k.tolist() == [0, 20200823]
    ks = prng.split(k, 3)
    assert ks.shape == (3, 2) and ks.dtype == np.uint32
    assert len({tuple(r) for r in ks.tolist()}) == 3
    np.testing.assert_array_equal(ks, prng.split(k, 3))


def test_randint_range_and_determinism():
    k = prng.PRNGKey(7)
    j = prng.randint(k, (4096,), 0, 12)
    assert j.dtype == np.int32 and j.min() == 0 and j.max() == 11
    np.testing.assert_array_equal(j, prng.randint(k, (4096,), 0, 12))
    counts = np.bincount(j, minlength=12)
    assert counts.min() > 250 and counts.max() < 450


def test_uniform_range():
    k = prng.PRNGKey(3)
    u = prng.uniform(k, (64, 128), maxval=1 / 128 - np.finfo(np.float32).eps)
    assert u.dtype == np.float32 and u.min() >= 0 and u.max() < 1 / 128


def test_published_jax_values():
    """Outputs printed in JAX's own documentation (jax.random module docstring: `random.uniform(random.PRNGKey(0))` ->
    0.41845703, `random.split(PRNGKey(0))`; JAX 101 PRNG tutorial: PRNGKey(42) -> new key / subkey).  They pin PRNGKey, the counter
    layout of split / _random_bits (first half of the counters -> word 0) and the 23-bit mantissa construction of uniform."""
    k0 = prng.PRNGKey(0)
    assert k0.tolist() == [0, 0]
    assert prng.split(k0).tolist() == [[4146024105, 967050713], [2718843009, 1272950319]]
    assert float(prng.uniform(k0, ())) == float(np.float32(0.41845703))
    k42 = prng.PRNGKey(42)
    assert k42.tolist() == [0, 42]
    assert prng.split(k42).tolist() == [[2465931498, 3679230171], [255383827, 267815257]]


def test_normal_published_values_and_the_exact_quantile():
    """jax.random.normal: the two draws JAX 101's PRNG tutorial prints (`random.normal(PRNGKey(0), (1,))` -> -0.20584226; after
    `key, subkey = random.split(key)`, `random.normal(subkey, (1,))` -> -1.2515389) pin the float32 erf_inv polynomial bit for bit (the
    exact inverse gives -0.20584227 / -1.2515386).  A second reading through the exact normal quantile in double precision
    (oracle/prng_ref.normal) bounds what the polynomial does everywhere else: same draw to 2e-7 relative at the median, 1e-5 in the tails."""
    from oracle import prng_ref as PR
    k0 = prng.PRNGKey(0)
    assert float(prng.normal(k0, (1,))[0]) == float(np.float32(-0.20584226))
    assert float(prng.normal(prng.split(k0)[1], (1,))[0]) == float(np.float32(-1.2515389))
    k = prng.PRNGKey(5)
    z = prng.normal(k, (20000,))
    exact = np.array(PR.normal((int(k[0]), int(k[1])), 20000), np.float32)
    rel = np.abs(z - exact) / np.maximum(np.abs(exact), 1e-30)
    assert np.abs(z - exact).max() < 5e-5 and np.median(rel) < 2e-7 and rel.max() < 2e-5
    assert z.dtype == np.float32 and abs(float(z.mean())) < 0.02 and abs(float(z.std()) - 1) < 0.02 and 3.5 < np.abs(z).max() < 5.5
    e = prng.erf_inv_f32(np.array([-1.0, 0.0, 1.0, 0.5], np.float32))
    assert e[0] == -np.inf and e[1] == 0 and e[2] == np.inf and abs(float(e[3]) - 0.4769362762) < 1e-7


def test_two_independent_restatements_agree_on_a_million_draws():
    """oracle/prng_ref.py is a second reading of jax 0.2.22's random.py, written separately in plain Python integers (one draw at a time,
    explicit wraps); the vectorised numpy implementation the product uses must agree with it word for word: 2^18 key splits' worth of
    bits, 2^19 uniforms and 2^18 randints over several spans (the reference's P = 8, 12, 24 and awkward ones) — ~10^6 draws."""
    from oracle import prng_ref as PR
    key = prng.PRNGKey(20200823)
    kt = (int(key[0]), int(key[1]))
    assert PR.threefry2x32(0x13198a2e, 0x03707344, 0x243f6a88, 0x85a308d3) == (0xc4923a9c, 0x483df7a0)      # the same Random123 vector
    n = 1 << 18
    assert prng.random_bits(key, (n + 1,)).tolist() == PR.bits(kt, n + 1)                    # odd count: the padded counter
    assert [tuple(r) for r in prng.split(key, 5).tolist()] == PR.split(kt, 5)
    sub = prng.split(key, 8)
    eps = float(np.finfo(np.float32).eps)
    u = prng.uniform(sub[0], (1 << 19,), maxval=1.0 / 192 - eps)                              # the stratum width of 64 + 128 samples
    ur = PR.uniform((int(sub[0][0]), int(sub[0][1])), 1 << 19, 0.0, 1.0 / 192 - eps)
    assert u.tolist() == [float(np.float32(v)) for v in ur]
    for i, span in enumerate((8, 12, 24, 3, 65537, 1 << 20)):
        k = sub[1 + i]
        m = (1 << 18) // 6 + 7
        assert prng.randint(k, (m,), 0, span).tolist() == PR.randint((int(k[0]), int(k[1])), m, 0, span), span
    assert prng.randint(sub[7], (1001,), -5, 7).tolist() == PR.randint((int(sub[7][0]), int(sub[7][1])), 1001, -5, 7)
