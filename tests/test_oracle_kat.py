"""Analytic known-answer tests that pin the CPU oracle (KATs 1-9).

The reference ships no tests for this path and JAX is unavailable offline, so these closed-form consequences of the
reference formulas are what anchors oracle/ref_np.py ("parity unpinned" otherwise).
"""
import numpy as np
import pytest

from oracle import ref_np as R
from samplenerfro_amd import synthetic as syn

F32 = np.float32


def _vacuum(G=8, ext=1.5):
    ndim, nmin, nmax = [G] * 3, [-ext] * 3, [ext] * 3
    table = R.build_table(np.ones(G ** 3), ndim, nmin, nmax)
    return table, ndim, nmin, nmax


def test_kat1_vacuum_straight_rays():
    table, ndim, nmin, nmax = _vacuum()
    assert np.all(table[:, 0] == 1) and np.all(table[:, 1:] == 0)
    o, d = syn.sphere_rays(8, seed=3)
    N, near, far = 48, 2.0, 6.0
    pos, dirs, dist, n, g = R.path_sampler(o, d, table, ndim, nmin, nmax, near, far, N)
    step = F32((far - near) / (N - 1))
    p = o + F32(near) * d
    t = np.full((8,), F32(near))
    for k in range(N):
        np.testing.assert_array_equal(pos[:, k], p)                       # repeated addition, bit exact
        np.testing.assert_array_equal(dist[:, k], t)
        q = p + (step / F32(1)) * d
        t = t + np.sqrt(((p - q) ** 2)[:, 0] + ((p - q) ** 2)[:, 1] + ((p - q) ** 2)[:, 2])
        p = q
    np.testing.assert_array_equal(dirs, np.broadcast_to(R.safe_l2_normalize(d)[:, None], dirs.shape))
    assert np.all(n == 1) and np.all(g == 0)


def test_kat2_stratified_slab_keeps_transverse_direction():
    G, ext = 16, 1.5
    ndim, nmin, nmax = [G] * 3, [-ext] * 3, [ext] * 3
    z = np.linspace(0, 1, G)
    grid = np.broadcast_to(1.0 + 0.3 * z[None, None, :], (G, G, G)).copy()
    table = R.build_table(grid, ndim, nmin, nmax)
    assert np.all(table[:, 1] == 0) and np.all(table[:, 2] == 0)
    o, d = syn.sphere_rays(8, seed=4)
    step = F32(4.0 / 63)
    rp = o + F32(2.0) * d
    rd = d.copy()
    for _ in range(64):   # eikonal_utils.py:41-42 with the raw (un-normalised) direction
        c = R.linear3(table, rp, ndim, nmin, nmax)
        rp = rp + step / c[:, :1] * rd
        rd2 = rd + step * c[:, 1:]
        np.testing.assert_array_equal(rd2[:, :2], rd[:, :2])               # discrete Snell invariant: bit constant
        rd = rd2
    assert np.any(rd[:, 2] != d[:, 2])


def test_kat3_clamp_to_edge():
    G, ext = 8, 1.0
    ndim, nmin, nmax = [G] * 3, [-ext] * 3, [ext] * 3
    rng = np.random.default_rng(0)
    table = rng.uniform(1, 2, (G ** 3, 4)).astype(F32)
    pts = np.array([[-5, -5, -5], [5, 5, 5], [-5, 0.3, 5]], F32)
    out, idx = R.linear3(table, pts, ndim, nmin, nmax, return_idx=True)
    np.testing.assert_allclose(out[0], table[0], rtol=1e-6)
    np.testing.assert_allclose(out[1], table[-1], rtol=1e-6)
    assert idx[0].tolist() == [0, 0, 0, 0, 0, 0] and idx[1].tolist() == [7] * 6
    assert idx[2, 0] == 0 and idx[2, 1] == 0 and idx[2, 4] == 7 and idx[2, 5] == 7


def test_kat4_gradient_table_linear_ramp():
    G, ext, a = 8, 1.0, 0.25
    ndim, nmin, nmax = [G] * 3, [-ext] * 3, [ext] * 3
    i = np.arange(G, dtype=np.float64)
    grid = np.broadcast_to(1.0 + a * i[:, None, None], (G, G, G)).copy()
    t = R.build_table(grid, ndim, nmin, nmax, np.float64).reshape(G, G, G, 4)
    nd = 2 * ext / (G - 1)
    np.testing.assert_allclose(t[1:-1, :, :, 1], a / nd, rtol=1e-12)
    np.testing.assert_allclose(t[0, :, :, 1], a / (2 * nd), rtol=1e-12)     # one-sided at the boundary (ior_utils.py:168)
    np.testing.assert_allclose(t[-1, :, :, 1], a / (2 * nd), rtol=1e-12)
    assert np.all(t[..., 2] == 0) and np.all(t[..., 3] == 0)
    # flat index convention: x slowest (ior_utils.py:214)
    flat = R.build_table(grid, ndim, nmin, nmax, np.float64)
    assert flat[3 * G * G + 2 * G + 1, 0] == grid[3, 2, 1]


def test_kat5_prefilter_constant_and_impulse():
    G = 9
    c = R.conv3d_normal(np.full((G ** 3, 1), 1.25), [G] * 3, 3, 1.0, np.float64)
    np.testing.assert_allclose(c, 1.25, rtol=1e-13)
    imp = np.zeros((G, G, G)); imp[4, 4, 4] = 1.0
    out = R.conv3d_normal(imp.reshape(-1, 1), [G] * 3, 5, 3.0, np.float64).reshape(G, G, G)
    k = R.gaussian_kernel3d(5, 3.0, np.float64)
    np.testing.assert_allclose(out[2:7, 2:7, 2:7], k[::-1, ::-1, ::-1], rtol=1e-12)
    np.testing.assert_allclose(k.sum(), 1.0, rtol=1e-13)
    # separability of the normalised kernel (what the HIP prefilter uses)
    e = np.exp(-np.arange(-2, 3) ** 2 / (2 * 3.0 ** 2)); e /= e.sum()
    np.testing.assert_allclose(k, e[:, None, None] * e[None, :, None] * e[None, None, :], rtol=1e-12)


def test_kat6_pos_enc_layout():
    x = np.array([[0.1, 0.2, 0.3]], F32)
    e = R.pos_enc(x, 0, 10)
    assert e.shape == (1, 63) and R.pos_enc(x, 0, 4).shape == (1, 27)
    exp = np.array([.1, .2, .3, np.sin(F32(.1)), np.sin(F32(.2)), np.sin(F32(.3)), np.sin(F32(.2)), np.sin(F32(.4)), np.sin(F32(.6))], F32)
    np.testing.assert_allclose(e[0, :9], exp, rtol=1e-6)
    np.testing.assert_allclose(e[0, 33:36], np.sin(x[0] + F32(0.5 * np.pi)), rtol=1e-6)   # cos block = sin(x + pi/2)
    a = R.annealed_pos_enc(x, 0, 10, 10.0)
    assert a.shape == (1, 60)
    np.testing.assert_allclose(a[0, :3], np.sin(x[0]), rtol=1e-6)
    np.testing.assert_allclose(a[0, 3:6], np.sin(x[0] + F32(0.5 * np.pi)), rtol=1e-6)
    assert np.all(R.annealed_pos_enc(x, 0, 10, 0.0) == 0)


def test_kat7_compositing_constant_density():
    B, S, sig, dl = 4, 16, 0.7, 0.25
    t = (2.0 + dl * np.arange(S))[None].repeat(B, 0)
    dirs = np.zeros((B, S, 3)); dirs[..., 2] = 1.0
    rgb = np.full((B, S, 3), 0.5)
    bk = np.full((B, 3), 0.25)
    out = R.volumetric_rendering(rgb, np.full((B, S, 1), sig), t, dirs, False, bk)
    acc = 1 - np.exp(-sig * (dl * (S - 1) + 1e-3))
    np.testing.assert_allclose(out[2], acc, rtol=1e-12)
    np.testing.assert_allclose(out[0], 0.5 * acc + (1 - acc) * 0.25, rtol=1e-12)
    np.testing.assert_allclose(out[5][:, 0], 1 - acc, rtol=1e-12)
    np.testing.assert_allclose(out[3].sum(-1), acc, rtol=1e-12)
    # acc == 0 -> 0/0 = NaN -> nan_to_num(., copy=inf) -> 0 -> clipped to t_0 (model_utils.py:304-305)
    out0 = R.volumetric_rendering(rgb, np.zeros((B, S, 1)), t, dirs, False, bk)
    assert np.all(out0[2] == 0) and np.all(out0[1] == t[:, 0])
    np.testing.assert_allclose(out0[0], 0.25)


def test_kat8_pdf_uniform_weights():
    B, nb, F = 3, 9, 32
    bins = np.linspace(2.0, 4.0, nb)[None].repeat(B, 0)
    w = np.ones((B, nb - 1))
    u = R.linspace_u(F, B, np.float64)
    z = R.sorted_piecewise_constant_pdf(u, bins, w)
    np.testing.assert_allclose(z, 2.0 + 2.0 * u, rtol=1e-12)
    # all-zero weights are padded to a uniform pdf (model_utils.py:327-331)
    z0 = R.sorted_piecewise_constant_pdf(u, bins, np.zeros((B, nb - 1)))
    np.testing.assert_allclose(z0, 2.0 + 2.0 * u, rtol=1e-9)


def test_kat9_resample_index_convention():
    N = 12
    z_vals = (2.0 + 0.5 * np.arange(N))[None]
    pos = np.zeros((1, N, 3)); pos[0, :, 0] = np.arange(N)
    dirs = np.zeros((1, N, 3)); dirs[..., 0] = 1.0
    jitter = np.array([0, 4, 8])
    bins = .5 * (z_vals[:, jitter][:, 1:] + z_vals[:, jitter][:, :-1])
    u = np.array([[0.0, 0.25, 0.5, 0.75]])
    z, p, d, g, idx = R.sample_pdf(u, bins, np.ones((1, 1)), pos, dirs, z_vals, np.zeros((1, N, 3)), jitter)
    assert z.shape == (1, 7) and np.all(np.diff(z[0]) >= 0)
    for zz, ii in zip(z[0], idx[0]):
        j = np.searchsorted(z_vals[0], zz, side="left")
        assert ii == max(j - 1, 0)
    # a sample exactly at a node depth maps to the PREVIOUS node; z <= z_vals[0] maps to node 0
    assert idx[0, 0] == 0 and z[0, 0] == z_vals[0, 0]
    k = list(z[0]).index(z_vals[0, 4]); assert idx[0, k] == 3
    np.testing.assert_allclose(p[0, :, 0], idx[0] + (z[0] - z_vals[0, idx[0]]))


def test_forward_shapes_and_f64_twin():
    G, ext = 12, 1.5
    grid = syn.scale_ior(syn.sphere_grid(G, ext, 0.6), 0.5)
    table = R.build_table(R.conv3d_normal(grid.reshape(-1, 1), [G] * 3, 3, 1.0), [G] * 3, [-ext] * 3, [ext] * 3)
    o, d = syn.sphere_rays(6, seed=1)
    cfg = R.ModelConfig([G] * 3, [-ext] * 3, [ext] * 3, num_coarse_samples=8, num_fine_samples=8, num_path_samples=3)
    params = syn.params_tree(syn.init_params_flat(0))
    jit = np.arange(0, 24, 3) + 1
    ret, _ = R.nerf_forward(cfg, params, table, o, d, jit)
    assert len(ret) == 2 and [x.shape for x in ret[1]] == [(6, 3), (6,), (6,), (6, 1), (6, 3)]
    assert all(x.dtype == np.float32 for x in ret[1])
    ret64, _ = R.nerf_forward(cfg, params, table.astype(np.float64), o.astype(np.float64), d.astype(np.float64), jit, dtype=np.float64)
    assert np.abs(ret64[0][0] - ret[0][0]).max() < 1e-4


def test_integrated_pos_enc_kat():
    """SURVEY 8f N4 (rnerf/mip.py:26-175).  On a STRAIGHT ray the mean accumulated along the path collapses to mip-NeRF's closed form
    o + d * t_mean; the encoding has 6 L features without identity, equals plain sin for zero variance and is damped by exp(-var/2)."""
    B, S = 5, 7
    rng = np.random.default_rng(3)
    o = rng.standard_normal((B, 3)); d = rng.standard_normal((B, 3)); d /= np.linalg.norm(d, axis=-1, keepdims=True)
    near = 2.0
    t = near + np.cumsum(rng.uniform(0.1, 0.4, (B, S)), axis=1) - 0.1
    pos = o[:, None] + t[..., None] * d[:, None]
    dirs = np.broadcast_to(d[:, None], (B, S, 3)).copy()
    radii = rng.uniform(1e-3, 3e-3, (B, 1))
    means, covs, enc = R.integrated_pos_enc_of_path(pos, dirs, t, radii, near, 0, 10, np.float64)
    t_vals = np.concatenate([t, t[:, -1:] + 1e-3], -1)
    mu, hw = (t_vals[:, :-1] + t_vals[:, 1:]) / 2, (t_vals[:, 1:] - t_vals[:, :-1]) / 2
    t_mean = mu + 2 * mu * hw ** 2 / (3 * mu ** 2 + hw ** 2)
    # straight ray: first sample position + d * (t_mean - t_0) ... the reference adds origins[:, 0:1] = the FIRST sample's position to a
    # cumsum that starts at t_mean_0 - near, so the closed form is pos_0 + d * (t_mean - near)
    want = pos[:, 0:1] + d[:, None] * (t_mean - near)[..., None]
    assert np.allclose(means, want, atol=1e-12)
    assert enc.shape == (B, S, 60) and covs.shape == (B, S, 3) and np.all(covs > 0)
    plain = R.integrated_pos_enc(means, np.zeros_like(covs), 0, 10, np.float64)
    sc = 2.0 ** np.arange(10)
    assert np.allclose(plain[..., :30], np.sin((means[..., None, :] * sc[:, None]).reshape(B, S, 30)), atol=1e-12)
    assert np.all(np.abs(enc) <= np.abs(plain) + 1e-15)                       # the Gaussian only damps
    k = 9                                                                     # highest degree: variance 4^9 times the base one
    ratio = enc[..., 3 * k:3 * k + 3] / plain[..., 3 * k:3 * k + 3]
    assert np.allclose(ratio, np.exp(-0.5 * covs * 4.0 ** k), rtol=1e-9)
    f32 = R.integrated_pos_enc_of_path(pos, dirs, t, radii, near, 0, 10, np.float32)[2]
    # fp32: the argument 2^9 x carries ~6e-5 of rounding at |x| ~ 4
    assert f32.dtype == np.float32 and np.abs(f32 - enc).max() < 5e-4
