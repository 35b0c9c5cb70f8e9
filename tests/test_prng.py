"""Host PRNG (samplenerfro_amd/prng.py): the Threefry-2x32-20 block cipher is pinned by the public Random123 vectors."""
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
