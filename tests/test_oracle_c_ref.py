"""Two independent restatements of the reference's table / trilinear lookup / eikonal march — the vectorised numpy oracle (oracle/ref_np.py)
and a scalar C reading written separately from the same reference lines (oracle/c_ref/march_ref.c, gcc -ffp-contract=off) — must agree BIT
FOR BIT: every table entry, every looked-up value, every clamped voxel index, every position, direction and distance along every path.
(The HIP kernels are held bit-equal to the numpy oracle by the GPU tests; the reference itself stays unobserved: parity unpinned.)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from oracle import c_ref as CR, ref_np as R                      # noqa: E402
from samplenerfro_amd import synthetic as syn                    # noqa: E402

F32 = np.float32


def _same_bits(a, b):
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b).reshape(a.shape)
    assert a.dtype == b.dtype and np.array_equal(a.view(np.uint32 if a.dtype == F32 else a.dtype), b.view(np.uint32 if b.dtype == F32 else b.dtype))


def test_table_and_lookup_bit_equal_on_an_anisotropic_grid():
    rng = np.random.default_rng(3)
    ndim, nmin, nmax = [17, 23, 11], [-1.79102, 0.711703, -1.75], [1.70898, 4.2117, 1.75]        # the glass bbox: three different cell sizes
    grid = (1.0 + 0.5 * rng.random(ndim)).astype(F32)
    t_np, t_c = R.build_table(grid, ndim, nmin, nmax), CR.build_table(grid, ndim, nmin, nmax)
    _same_bits(t_np, t_c)
    lo, hi = np.array(nmin) - 0.4, np.array(nmax) + 0.4                                         # a fifth of the points outside: clamp to edge
    pts = (lo + (hi - lo) * rng.random((20000, 3))).astype(F32)
    # points exactly ON cell faces and corners too (floor() ties, the unclamped-weight rule at the upper face)
    ax = [np.linspace(nmin[a], nmax[a], ndim[a]).astype(F32) for a in range(3)]
    on = np.stack([rng.choice(ax[0], 3000), rng.choice(ax[1], 3000), rng.choice(ax[2], 3000)], -1).astype(F32)
    pts = np.concatenate([pts, on], 0)
    v_np, i_np = R.linear3(t_np, pts, ndim, nmin, nmax, F32, True)
    v_c, i_c = CR.linear3(t_np, pts, ndim, nmin, nmax)
    _same_bits(i_np, i_c)
    _same_bits(v_np.astype(F32), v_c)


def test_march_bit_equal_on_the_example_scene_and_a_refractive_sphere():
    import cases
    c = cases.inputs_example()                                            # the reference's camera + the grid voxelised from its OBJ (128^3, N = 768)
    n = 40
    o, d = c["origins"][:n], c["viewdirs"][:n]
    table = R.build_table(c["grid"], c["ndim"], c["nmin"], c["nmax"])
    _same_bits(table, CR.build_table(c["grid"], c["ndim"], c["nmin"], c["nmax"]))
    N = c["S"] * c["P"]
    a = R.path_sampler(o, d, table, c["ndim"], c["nmin"], c["nmax"], c["near"], c["far"], N, F32, return_idx=True)
    b = CR.path_sampler(o, d, table, c["ndim"], c["nmin"], c["nmax"], c["near"], c["far"], N)
    for x, y in zip(a, b):
        _same_bits(x, y)
    assert a[5].shape == (n, N, 6) and (np.diff(a[2], axis=1) > 0).all()                 # every step travels
    bent = np.abs(a[1] - a[1][:, :1]).max(axis=(1, 2)) > 1e-3                             # some of these rays really refract
    assert bent.any()
    # a steeper field: rays through a sphere of n = 1.5 on a coarse grid, 600 steps, a few of them grazing
    G, ext = 40, 1.5
    grid = R.conv3d_normal(syn.scale_ior(syn.sphere_grid(G, ext, 0.6), 0.5).reshape(-1, 1), [G] * 3, 3, 1.0).reshape(G, G, G).astype(F32)
    table = R.build_table(grid, [G] * 3, [-ext] * 3, [ext] * 3)
    o, d = syn.sphere_rays(64, seed=9)
    a = R.path_sampler(o, d, table, [G] * 3, [-ext] * 3, [ext] * 3, 2.0, 6.0, 600, F32, return_idx=True)
    b = CR.path_sampler(o, d, table, [G] * 3, [-ext] * 3, [ext] * 3, 2.0, 6.0, 600)
    for x, y in zip(a, b):
        _same_bits(x, y)


def test_resampling_bit_equal_incl_ties_and_empty_weights():
    """S1 / S2: the numpy oracle restates the reference's dense mask literally, the C reading walks the sorted cdf.  Same fine depths, same
    merged order, same node indices (the bit-exact integer contract), same interpolated positions — on real march output with rendering-like
    weights, on rays whose weights are all zero (the eps padding path), and with draws that hit cdf entries exactly (ties of `>=`)."""
    rng = np.random.default_rng(12)
    G, ext = 32, 1.5
    grid = R.conv3d_normal(syn.scale_ior(syn.sphere_grid(G, ext, 0.6), 0.5).reshape(-1, 1), [G] * 3, 3, 1.0).reshape(G, G, G).astype(F32)
    table = R.build_table(grid, [G] * 3, [-ext] * 3, [ext] * 3)
    B, S, P, Fn = 48, 16, 6, 40
    N = S * P
    o, d = syn.sphere_rays(B, seed=4)
    pos, dr, dist, _, grad = R.path_sampler(o, d, table, [G] * 3, [-ext] * 3, [ext] * 3, 2.0, 6.0, N, F32)
    jitter = (np.arange(0, N, P) + rng.integers(0, P, S)).astype(np.int32)
    t = dist[:, jitter]
    mids = (F32(0.5) * (t[:, 1:] + t[:, :-1])).astype(F32)                     # bins (models.py:371-373): S - 1 edges, S - 2 weights
    w = rng.random((B, S - 2)).astype(F32) ** 4
    w[3] = 0.0                                                                 # a ray that met nothing: weight_sum = 0 -> eps padding
    w[5, :] = 0.0; w[5, 7] = 1.0                                               # all the mass in one bin: cdf steps 0 -> 1, many draws tie
    eps = float(np.finfo(np.float32).eps)
    u = (np.arange(Fn, dtype=F32) * F32(1.0 / Fn))[None, :] + rng.random((B, Fn)).astype(F32) * F32(1.0 / Fn - eps)
    u = np.minimum(u, F32(1.0 - eps)).astype(F32)
    u[7] = R.linspace_u(Fn, 1)[0]                                              # the deterministic draws of randomized=False
    # draws that equal a cdf entry exactly: take row 9's own cdf values as its draws
    pdf = (w[9] + max(0.0, 1e-5 - w[9].sum())) / max(w[9].sum(), 1e-5)
    cdf9 = np.minimum(1.0, np.cumsum(pdf[:-1], dtype=F32)).astype(F32)
    u[9, :len(cdf9)] = np.sort(np.minimum(cdf9, F32(1.0 - eps)))[:Fn][:len(cdf9)]
    u[9] = np.sort(u[9])
    a = R.sample_pdf(u, mids, w, pos, dr, dist, grad, jitter)
    b = CR.sample_pdf(u, mids, w, pos, dr, dist, jitter)
    _same_bits(a[4], b[3])                                                     # node indices
    _same_bits(a[0], b[0]); _same_bits(a[1], b[1]); _same_bits(a[2], b[2])
    assert (np.diff(b[0], axis=1) >= 0).all() and b[3].min() >= 0 and b[3].max() <= N - 1
