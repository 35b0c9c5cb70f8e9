"""numpy restatement of the SampleNeRFRO hot path (TEST INFRASTRUCTURE, parity unpinned).

Pinned by the reference itself: build_table's gradient columns, linear3 and generate_rays — against vectors its own numpy-only methods
compute (rnerf/datasets.py Grid._compute_grad / _linear3, Dataset / OpenCV._generate_rays; tests/test_reference_numpy_pin.py).  Everything
else is held only by analytic known-answer tests, published jax.random values and a second, independently written reading.

Every function cites the reference lines it restates (paths relative to
/root/reference).  The arithmetic is done in `dtype` (float32 by default, the
reference's precision; float64 gives the "truth" twin used to set tolerances).
In float32 mode every elementary op is individually rounded (numpy never
contracts a*b+c into an FMA), which is the op order the HIP kernels reproduce
with -ffp-contract=off, so that all integer outputs (voxel indices, resample
node indices) can be compared bit-exactly.

Not importable from the product package: see oracle/__init__.py.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

F32 = np.float32
HALF_PI_F32 = np.float32(0.5 * np.pi)

# XLA may rewrite a division by a compile-time constant, x / c, into x * RN(1 / c) (its algebraic simplifier does so when the reciprocal
# is "close enough"; whether jax 0.2.22's XLA did it for (p - nmin) / ndelta cannot be observed here: no JAX).  The two forms differ by
# one ulp on a fraction of the inputs, which flips floor() for points within an ulp of a cell face.  With this switch on, the divisions
# by launch constants in linear3 / build_table use the reciprocal form, so that the exposure of the "bit-exact index" claim to that
# rewrite can be MEASURED (tools/xla_rcp_exposure.py, tests/test_oracle_exposure.py).  Default: the division as the reference writes it.
CONST_DIV_AS_RECIPROCAL = False


def _cdiv(x, c, dtype):
    """x / c for a launch-constant c, in the form selected by CONST_DIV_AS_RECIPROCAL."""
    if CONST_DIV_AS_RECIPROCAL:
        return x * dtype(dtype(1) / dtype(c))
    return x / dtype(c)


# ----------------------------------------------------------------------------
# G1: Gaussian prefilter of the IoR grid        (rnerf/ior_utils.py:327-363)
# ----------------------------------------------------------------------------
def scale_ior(data, refractive_index: float) -> np.ndarray:
    """train.py:220-222 / eval.py:80-82: (data - 1) * ri / 0.33 + 1 in float64."""
    return (np.asarray(data, np.float64) - 1.0) * refractive_index / 0.33 + 1.0


def gaussian_kernel3d(ws: int, s: float, dtype=F32) -> np.ndarray:
    """ior_utils.py:345-348: exp(-(x^2+y^2+z^2)/(2 s^2)) / sum."""
    hws = ws // 2
    a = np.linspace(-hws, hws, ws).astype(dtype)
    xx, yy, zz = np.meshgrid(a, a, a)
    kernel = np.exp(-(xx ** 2 + yy ** 2 + zz ** 2) / dtype(2.0 * s ** 2)).astype(dtype)
    return (kernel / kernel.sum(dtype=dtype)).astype(dtype)


def conv3d_normal(grid, ndim: Sequence[int], ws: int, s: float, dtype=F32) -> np.ndarray:
    """ior_utils.py:327-363: edge pad ws//2, VALID dense correlation. -> [G^3, 1]."""
    hws = ws // 2
    data = np.asarray(grid).astype(dtype).reshape(ndim[0], ndim[1], ndim[2])
    data = np.pad(data, ((hws, hws), (hws, hws), (hws, hws)), "edge")
    kernel = gaussian_kernel3d(ws, s, dtype)
    out = np.zeros(tuple(ndim), dtype)
    for i in range(ws):
        for j in range(ws):
            for k in range(ws):
                out += kernel[i, j, k] * data[i:i + ndim[0], j:j + ndim[1], k:k + ndim[2]]
    return out.reshape(-1, 1)


# ----------------------------------------------------------------------------
# G2: gradient table                               (rnerf/ior_utils.py:139-172)
# ----------------------------------------------------------------------------
def compute_ndelta(ndim, nmin, nmax) -> List[float]:
    """ior_utils.py:140-144 (python doubles)."""
    return [(float(nmax[i]) - float(nmin[i])) / (ndim[i] - 1.0) for i in range(3)]


def build_table(grid, ndim, nmin, nmax, dtype=F32) -> np.ndarray:
    """ior_utils.py:161,165-172: data = concat([grid, central differences]) -> [G^3, 4]."""
    ndelta = compute_ndelta(ndim, nmin, nmax)
    g = np.asarray(grid).astype(dtype).reshape(ndim[0], ndim[1], ndim[2])
    p = np.pad(g, ((1, 1), (1, 1), (1, 1)), "edge")
    dx = _cdiv(p[2:, 1:-1, 1:-1] - p[:-2, 1:-1, 1:-1], 2 * ndelta[0], dtype)
    dy = _cdiv(p[1:-1, 2:, 1:-1] - p[1:-1, :-2, 1:-1], 2 * ndelta[1], dtype)
    dz = _cdiv(p[1:-1, 1:-1, 2:] - p[1:-1, 1:-1, :-2], 2 * ndelta[2], dtype)
    return np.stack([g, dx, dy, dz], axis=-1).reshape(-1, 4).astype(dtype)


# ----------------------------------------------------------------------------
# G3: trilinear lookup                             (rnerf/ior_utils.py:188-223)
# ----------------------------------------------------------------------------
def linear3(table, pts, ndim, nmin, nmax, dtype=F32, return_idx: bool = False):
    """ior_utils.py:188-223. pts [...,3] -> [...,4]; optional int32 idx [...,6] = x0,x1,y0,y1,z0,z1 (clamped)."""
    ndelta = compute_ndelta(ndim, nmin, nmax)
    pts = np.asarray(pts, dtype)
    x = _cdiv(pts[..., 0] - dtype(nmin[0]), ndelta[0], dtype)
    y = _cdiv(pts[..., 1] - dtype(nmin[1]), ndelta[1], dtype)
    z = _cdiv(pts[..., 2] - dtype(nmin[2]), ndelta[2], dtype)
    x0 = np.floor(x).astype(np.int32); x1 = x0 + 1
    y0 = np.floor(y).astype(np.int32); y1 = y0 + 1
    z0 = np.floor(z).astype(np.int32); z1 = z0 + 1
    # weights are taken BEFORE clamping (ior_utils.py:201-203)
    xd = ((x - x0.astype(dtype)) / (x1 - x0).astype(dtype))[..., None]
    yd = ((y - y0.astype(dtype)) / (y1 - y0).astype(dtype))[..., None]
    zd = ((z - z0.astype(dtype)) / (z1 - z0).astype(dtype))[..., None]
    x0 = np.clip(x0, 0, ndim[0] - 1); x1 = np.clip(x1, 0, ndim[0] - 1)
    y0 = np.clip(y0, 0, ndim[1] - 1); y1 = np.clip(y1, 0, ndim[1] - 1)
    z0 = np.clip(z0, 0, ndim[2] - 1); z1 = np.clip(z1, 0, ndim[2] - 1)
    s1, s2 = ndim[1] * ndim[2], ndim[2]
    one = dtype(1)
    d = table
    c00 = d[s1 * x0 + s2 * y0 + z0] * (one - xd) + d[s1 * x1 + s2 * y0 + z0] * xd
    c01 = d[s1 * x0 + s2 * y0 + z1] * (one - xd) + d[s1 * x1 + s2 * y0 + z1] * xd
    c10 = d[s1 * x0 + s2 * y1 + z0] * (one - xd) + d[s1 * x1 + s2 * y1 + z0] * xd
    c11 = d[s1 * x0 + s2 * y1 + z1] * (one - xd) + d[s1 * x1 + s2 * y1 + z1] * xd
    c0 = c00 * (one - yd) + c10 * yd
    c1 = c01 * (one - yd) + c11 * yd
    c = c0 * (one - zd) + c1 * zd
    if return_idx:
        return c, np.stack([x0, x1, y0, y1, z0, z1], axis=-1).astype(np.int32)
    return c


# ----------------------------------------------------------------------------
# E3: safe math                                     (rnerf/math_utils.py:6-20)
# ----------------------------------------------------------------------------
def _seqsum(x, axis=-1, keepdims=False):
    """Strictly sequential (left-to-right) sum: the accumulation order of the HIP kernels.  XLA's reduction order is
    unspecified, numpy's np.sum is pairwise; a sequential scan keeps oracle and kernel bit-comparable."""
    c = np.take(np.cumsum(x, axis=axis, dtype=x.dtype), -1, axis=axis)
    return np.expand_dims(c, axis) if keepdims else c


def _sum3_sq(x):
    return (x[..., 0:1] * x[..., 0:1] + x[..., 1:2] * x[..., 1:2]) + x[..., 2:3] * x[..., 2:3]


def safe_l2_norm(x, eps=1e-6):
    return np.sqrt(np.maximum(_sum3_sq(x), x.dtype.type(eps)))


def safe_l2_normalize(x, eps=1e-6):
    return x / safe_l2_norm(x, eps)


# ----------------------------------------------------------------------------
# E1/E2: eikonal march                       (rnerf/eikonal_utils.py:29-49,100-124)
# ----------------------------------------------------------------------------
def normal_loss_and_smooth(table, so3_params, ray_pos, idx_grad, ndim, nmin, nmax, annealed_alpha, noise, dtype=F32):
    """E4: PathSampler.compute_normal_loss_and_smooth (rnerf/eikonal_utils.py:84-98) with the random draw `noise` given explicitly."""
    x = np.asarray(ray_pos, dtype).reshape(-1, 3); g = np.asarray(idx_grad, dtype).reshape(-1, 3)
    ndelta = np.array(compute_ndelta(ndim, nmin, nmax), dtype)
    pred = vox_mlp_call(table, so3_params, x, ndim, nmin, nmax, annealed_alpha, dtype, condition=g)[2]
    pred_r = vox_mlp_call(table, so3_params, x + np.asarray(noise, dtype).reshape(-1, 3) * ndelta, ndim, nmin, nmax, annealed_alpha, dtype,
                          condition=g)[2]
    factor = safe_l2_norm(g)
    return 0.0, float((np.abs((pred - pred_r) / factor)).sum(-1).mean())


def vox_mlp_call(table, so3_params, pts, ndim, nmin, nmax, annealed_alpha=1.0, dtype=F32, max_deg_point=10, condition=None):
    """VoxMLP.__call__ (rnerf/ior_utils.py:269-312) with the shipped gin settings (annealed, use_residual, use_direct_output):
    -> (n [B,1], grad n [B,3], pred_grad [B,3]); pred_grad = grad n rotated by the axis-angle so3_mlp(annealed_pos_enc(x))."""
    ret = linear3(table, pts, ndim, nmin, nmax, dtype)
    n, g = ret[:, :1], ret[:, 1:]
    if condition is not None:                                               # wrapper_grad_mlp (:225-267) rotates the given vector
        g = np.asarray(condition, dtype)
    enc = annealed_pos_enc(np.asarray(pts, dtype)[:, None], 0, max_deg_point, dtype(annealed_alpha) * dtype(max_deg_point), dtype)   # :283
    raw = simple_mlp(so3_params, enc)[:, 0]
    theta = safe_l2_norm(raw)                                               # :305-312
    e = raw / theta
    a = safe_l2_norm(g)
    v = g / a
    cos_t, sin_t = np.cos(theta).astype(dtype), np.sin(theta).astype(dtype)
    cross = np.stack([e[:, 1] * v[:, 2] - e[:, 2] * v[:, 1], e[:, 2] * v[:, 0] - e[:, 0] * v[:, 2], e[:, 0] * v[:, 1] - e[:, 1] * v[:, 0]], -1)
    dot = _seqsum(e * v, axis=-1, keepdims=True)
    pred = a * (cos_t * v + sin_t * cross + (dtype(1) - cos_t) * dot * e)
    return n, g, pred.astype(dtype)


def path_sampler(origins, viewdirs, table, ndim, nmin, nmax, near: float, far: float,
                 num_samples: int, dtype=F32, return_idx: bool = False, so3_params=None, annealed_alpha=1.0):
    """PathSampler.__call__ with stage="radiance" (grad = table gradient), or stage="all" when so3_params is given
    (grad = where(|grad n| > 1e-3, pred_grad, grad n), rnerf/eikonal_utils.py:34-39).

    Returns (ray_pos [B,N,3], ray_dir [B,N,3] normalised, ray_dist [B,N], idx_data [B,N,1], idx_grad [B,N,3])
    (+ voxel idx [B,N,6] when return_idx).  step_size as models.py:122.
    """
    step_size = (float(far) - float(near)) / (num_samples - 1)          # models.py:121-122
    step = dtype(step_size)
    origins = np.asarray(origins, dtype); viewdirs = np.asarray(viewdirs, dtype)
    B = origins.shape[0]
    rp = origins + dtype(near) * viewdirs                                # eikonal_utils.py:104
    rd = viewdirs.copy()                                                 # :105
    rt = dtype(near) * np.ones((B, 1), dtype)                            # :106
    pos = np.empty((B, num_samples, 3), dtype); dirs = np.empty((B, num_samples, 3), dtype)
    dist = np.empty((B, num_samples), dtype)
    idx_data = np.empty((B, num_samples, 1), dtype); idx_grad = np.empty((B, num_samples, 3), dtype)
    vox = np.empty((B, num_samples, 6), np.int32) if return_idx else None
    for k in range(num_samples):
        pos[:, k] = rp; dirs[:, k] = rd; dist[:, k] = rt[:, 0]           # node k = state before step k (:112-114)
        if return_idx:
            ret, vi = linear3(table, rp, ndim, nmin, nmax, dtype, True); vox[:, k] = vi
        else:
            ret = linear3(table, rp, ndim, nmin, nmax, dtype)
        n = ret[:, :1]; g = ret[:, 1:]
        idx_data[:, k] = n; idx_grad[:, k] = g                           # :115-116 (not shifted)
        if so3_params is not None:                                       # stage "all" (:34-39)
            _, _, pred = vox_mlp_call(table, so3_params, rp, ndim, nmin, nmax, annealed_alpha, dtype)
            g = np.where(np.sqrt(_sum3_sq(g)) > dtype(1e-3), pred, g)
        next_rp = rp + step / n * rd                                     # :41
        next_rd = rd + step * g                                          # :42
        dlt = rp - next_rp
        next_rt = rt + np.sqrt(_sum3_sq(dlt))                            # :45
        rp, rd, rt = next_rp, next_rd, next_rt
    out = (pos, safe_l2_normalize(dirs), dist, idx_data, idx_grad)
    return out + (vox,) if return_idx else out


# ----------------------------------------------------------------------------
# P1/P2: positional encodings                  (rnerf/model_utils.py:187-245)
# ----------------------------------------------------------------------------
def pos_enc(x, min_deg: int, max_deg: int, dtype=F32):
    """model_utils.py:187-214, legacy_posenc_order=False."""
    x = np.asarray(x, dtype)
    if min_deg == max_deg:
        return x
    scales = np.array([2 ** i for i in range(min_deg, max_deg)], dtype)
    xb = (x[..., None, :] * scales[:, None]).reshape(list(x.shape[:-1]) + [-1])
    four = np.sin(np.concatenate([xb, xb + dtype(0.5 * np.pi)], axis=-1)).astype(dtype)
    return np.concatenate([x, four], axis=-1)


# ----------------------------------------------------------------------------
# mip-style integrated positional encoding along the curved ray  (rnerf/mip.py:26-57,60-91,116-175)
# (every call site in the reference is commented out, rnerf/models.py:249-254,386-391; restated as those comments would call it)
# ----------------------------------------------------------------------------
def _safe_trig(x, fn, dtype):
    """math_utils.safe_trig_helper (rnerf/math_utils.py:28-39): fn(where(|x| < 100 pi, x, x % (100 pi)))."""
    t = dtype(100 * np.pi)
    return fn(np.where(np.abs(x) < t, x, np.mod(x, t))).astype(dtype)


def conical_frustum_to_gaussian(d, t0, t1, base_radius, near, dtype=F32):
    """mip.py:60-91 (stable form) + lift_gaussian :35-57 with diag=True.  d [B,S,3]; t0, t1 [B,S]; base_radius [B,1].
    -> mean [B,S,3] (before the origin is added), cov_diag [B,S,3]."""
    d = np.asarray(d, dtype); t0 = np.asarray(t0, dtype); t1 = np.asarray(t1, dtype); br = np.asarray(base_radius, dtype)
    two, three = dtype(2), dtype(3)
    mu = (t0 + t1) / two
    hw = (t1 - t0) / two
    mu2, hw2 = mu * mu, hw * hw                   # x**2 / x**4 as jax lowers them (lax.integer_pow: repeated squaring)
    hw4 = hw2 * hw2
    den = three * mu2 + hw2
    t_mean = mu + (two * mu * hw2) / den
    t_var = hw2 / three - dtype(4 / 15) * ((hw4 * (dtype(12) * mu2 - hw2)) / (den * den))
    r_var = (br * br) * (mu2 / dtype(4) + dtype(5 / 12) * hw2 - dtype(4 / 15) * hw4 / den)
    t = np.concatenate([t_mean[:, 0:1] - dtype(near), t_mean[:, 1:] - t_mean[:, :-1]], axis=-1)[..., None]      # :38
    mean = np.zeros_like(d)
    run = np.zeros((d.shape[0], 3), dtype)
    for s in range(d.shape[1]):                                             # jnp.cumsum(d * t, axis=1), sequential order
        run = run + d[:, s] * t[:, s]
        mean[:, s] = run
    d_mag_sq = np.maximum(dtype(1e-10), _seqsum(d * d, axis=-1, keepdims=True))
    d_outer_diag = d * d
    null_outer_diag = dtype(1) - d_outer_diag / d_mag_sq
    cov_diag = t_var[..., None] * d_outer_diag + r_var[..., None] * null_outer_diag
    return mean.astype(dtype), cov_diag.astype(dtype)


def cast_rays_cone(t_vals, origins, directions, radii, near, dtype=F32):
    """mip.cast_rays(t_vals, origins, directions, radii, "cone", near) (mip.py:116-140) as the commented call sites use it:
    t_vals [B,S+1] = [ray_dist_c, last + 1e-3]; origins / directions = the coarse samples' positions / directions [B,S,3]."""
    t_vals = np.asarray(t_vals, dtype)
    means, covs = conical_frustum_to_gaussian(directions, t_vals[..., :-1], t_vals[..., 1:], radii, near, dtype)
    return means + np.asarray(origins, dtype)[:, 0:1], covs


def integrated_pos_enc(x, x_cov_diag, min_deg: int, max_deg: int, dtype=F32):
    """mip.integrated_pos_enc, diag=True (mip.py:143-175) + expected_sin (:26-32): exp(-0.5 var) * safe_sin(x) for
    [y, y + pi/2], y = x * 2^deg degree-major -> [..., 6 * (max_deg - min_deg)] (no identity features)."""
    x = np.asarray(x, dtype); c = np.asarray(x_cov_diag, dtype)
    scales = np.array([2 ** i for i in range(min_deg, max_deg)], dtype)
    shape = list(x.shape[:-1]) + [-1]
    y = (x[..., None, :] * scales[:, None]).reshape(shape)
    y_var = (c[..., None, :] * scales[:, None] ** 2).reshape(shape)
    xx = np.concatenate([y, y + dtype(0.5 * np.pi)], axis=-1)
    vv = np.concatenate([y_var, y_var], axis=-1)
    return (np.exp(dtype(-0.5) * vv) * _safe_trig(xx, np.sin, dtype)).astype(dtype)


def integrated_pos_enc_of_path(ray_pos_c, ray_dir_c, ray_dist_c, radii, near, min_deg=0, max_deg=10, dtype=F32):
    """The commented call sequence of rnerf/models.py:249-254 on the coarse samples of a marched path -> (means, covs, enc)."""
    ray_dist_c = np.asarray(ray_dist_c, dtype)
    t_vals = np.concatenate([ray_dist_c, ray_dist_c[..., -1:] + dtype(1e-3)], axis=-1)
    means, covs = cast_rays_cone(t_vals, ray_pos_c, ray_dir_c, radii, near, dtype)
    return means, covs, integrated_pos_enc(means, covs, min_deg, max_deg, dtype)


def cosine_easing_window(min_freq_log2, max_freq_log2, num_bands, alpha, dtype=F32):
    """model_utils.py:218-233."""
    bands = np.linspace(min_freq_log2, max_freq_log2, num_bands).astype(dtype)
    x = np.clip(dtype(alpha) - bands, dtype(0), dtype(1))
    return (dtype(0.5) * (dtype(1) + np.cos(dtype(np.pi) * x + dtype(np.pi)))).astype(dtype)


def annealed_pos_enc(x, min_deg, max_deg, alpha, dtype=F32):
    """model_utils.py:236-245 (no identity term)."""
    x = np.asarray(x, dtype)
    scales = np.array([2 ** i for i in range(min_deg, max_deg)], dtype)
    xb = x[..., None, :] * scales[:, None]
    window = cosine_easing_window(min_deg, max_deg - 1, len(scales), alpha, dtype)[:, None]
    four = np.concatenate([np.sin(xb).astype(dtype) * window,
                           np.sin(xb + dtype(0.5 * np.pi)).astype(dtype) * window], axis=-1)
    return four.reshape(list(x.shape[:-1]) + [-1])


# ----------------------------------------------------------------------------
# N1/N2: MLPs                                   (rnerf/model_utils.py:30-140)
# ----------------------------------------------------------------------------
def _dense(p, x, acc_dtype=None):
    k, b = p["kernel"], p["bias"]
    if acc_dtype is not None and acc_dtype != x.dtype:
        return (x.astype(acc_dtype) @ k.astype(acc_dtype) + b.astype(acc_dtype)).astype(x.dtype)
    return x @ k.astype(x.dtype) + b.astype(x.dtype)


def nerf_mlp(params: Dict, x, condition, acc_dtype=None,
             net_depth=8, skip_layer=4, net_depth_condition=1):
    """model_utils.py:30-90. x [B,S,63], condition [B,S,27] -> raw_rgb [B,S,3], raw_sigma [B,S,1]."""
    S = x.shape[1]
    x = x.reshape(-1, x.shape[-1]); inputs = x
    li = 0
    for i in range(net_depth):
        x = np.maximum(_dense(params[f"Dense_{li}"], x, acc_dtype), 0); li += 1
        if i % skip_layer == 0 and i > 0:
            x = np.concatenate([x, inputs], axis=-1)
    raw_sigma = _dense(params[f"Dense_{li}"], x, acc_dtype).reshape(-1, S, 1); li += 1
    if condition is not None:
        bottleneck = _dense(params[f"Dense_{li}"], x, acc_dtype); li += 1
        condition = condition.reshape(-1, condition.shape[-1])
        x = np.concatenate([bottleneck, condition], axis=-1)
        for _ in range(net_depth_condition):
            x = np.maximum(_dense(params[f"Dense_{li}"], x, acc_dtype), 0); li += 1
    raw_rgb = _dense(params[f"Dense_{li}"], x, acc_dtype).reshape(-1, S, 3)
    return raw_rgb, raw_sigma


def simple_mlp(params: Dict, x, acc_dtype=None, net_depth=4, skip_layer=2):
    """model_utils.py:93-140 as instantiated at models.py:116-118 / ior_utils.py:148-152 (no condition)."""
    S = x.shape[1]
    x = x.reshape(-1, x.shape[-1]); inputs = x
    li = 0
    for i in range(net_depth):
        x = np.maximum(_dense(params[f"Dense_{li}"], x, acc_dtype), 0); li += 1
        if i % skip_layer == 0 and i > 0:
            x = np.concatenate([x, inputs], axis=-1)
    out = _dense(params[f"Dense_{li}"], x, acc_dtype)
    return out.reshape(-1, S, out.shape[-1])


def sigmoid(x):
    one = x.dtype.type(1)
    with np.errstate(over="ignore"):
        return one / (one + np.exp(-x))


def softplus(x):
    """jax.nn.softplus = logaddexp(x, 0)."""
    with np.errstate(over="ignore"):
        return (np.maximum(x, 0) + np.log1p(np.exp(-np.abs(x)))).astype(x.dtype)


def rgb_activation(raw, rgb_padding=0.001):
    """models.py:334-337."""
    t = raw.dtype.type
    return sigmoid(raw) * t(1 + 2 * rgb_padding) - t(rgb_padding)


def sigma_activation(raw, sigma_bias=-1.0):
    """models.py:338."""
    return softplus(raw + raw.dtype.type(sigma_bias))


# ----------------------------------------------------------------------------
# V1: compositing                              (rnerf/model_utils.py:247-309)
# ----------------------------------------------------------------------------
def volumetric_rendering(rgb, density, t_vals, dirs, white_bkgd, rgb_bkgd, mask_bbox=None):
    """model_utils.py:247-309. dirs is [B,S,3] (per-sample bent direction, models.py:345)."""
    dt = rgb.dtype.type
    t_dists = np.concatenate([t_vals[..., 1:] - t_vals[..., :-1],
                              np.broadcast_to(np.array([1e-3], rgb.dtype), t_vals[..., :1].shape)], -1)
    delta = t_dists * np.sqrt(_sum3_sq(dirs))[..., 0]
    density_delta = density[..., 0] * delta
    if mask_bbox is not None:
        density_delta = density_delta * mask_bbox.astype(rgb.dtype)
    alpha = dt(1) - np.exp(-density_delta)
    trans = np.exp(-np.concatenate([np.zeros_like(density_delta[..., :1]),
                                    np.cumsum(density_delta, axis=-1, dtype=rgb.dtype)], axis=-1))
    weights = alpha * trans[..., :-1]
    if rgb_bkgd is not None:
        comp_rgb = _seqsum(weights[..., None] * rgb, axis=-2) + trans[..., -1:] * rgb_bkgd
    else:
        comp_rgb = _seqsum(weights[..., None] * rgb, axis=-2)
        rgb_bkgd = np.ones(list(trans[..., -1:].shape[:-1]) + [3], rgb.dtype)
    acc = _seqsum(weights, axis=-1)
    with np.errstate(invalid="ignore", divide="ignore"):
        distance = _seqsum(weights * t_vals, axis=-1) / acc
    # jnp.nan_to_num(distance, jnp.inf): the 2nd positional arg is `copy`, so NaN -> 0.0 (model_utils.py:304)
    distance = np.nan_to_num(distance, nan=0.0, posinf=np.finfo(rgb.dtype).max, neginf=np.finfo(rgb.dtype).min)
    distance = np.clip(distance, t_vals[:, 0], t_vals[:, -1])
    if white_bkgd:
        comp_rgb = comp_rgb + (dt(1) - acc[..., None])
    return comp_rgb, distance, acc, weights, alpha, trans[..., -1:], trans[..., -1:] * rgb_bkgd


# ----------------------------------------------------------------------------
# S1/S2: hierarchical resampling               (rnerf/model_utils.py:312-435)
# ----------------------------------------------------------------------------
def sorted_piecewise_constant_pdf(u, bins, weights, return_idx=False):
    """model_utils.py:312-374 with the uniform draws `u` [B,N_f] supplied by the caller.

    (randomized=False: u = linspace(0, 1-eps32, N_f) broadcast, :355-356.)
    The dense [B,Nb,N_f] mask of the reference is restated literally.
    """
    dt = bins.dtype.type
    eps = dt(1e-5)
    weights = weights.copy()
    weight_sum = _seqsum(weights, axis=-1, keepdims=True)
    padding = np.maximum(dt(0), eps - weight_sum)
    weights = weights + padding / dt(weights.shape[-1])
    weight_sum = weight_sum + padding
    pdf = weights / weight_sum
    cdf = np.minimum(dt(1), np.cumsum(pdf[..., :-1], axis=-1, dtype=bins.dtype))
    cdf = np.concatenate([np.zeros(list(cdf.shape[:-1]) + [1], bins.dtype), cdf,
                          np.ones(list(cdf.shape[:-1]) + [1], bins.dtype)], axis=-1)
    mask = u[..., None, :] >= cdf[..., :, None]

    def find_interval(x):
        x0 = np.max(np.where(mask, x[..., None], x[..., :1, None]), -2)
        x1 = np.min(np.where(~mask, x[..., None], x[..., -1:, None]), -2)
        return x0, x1

    bins_g0, bins_g1 = find_interval(bins)
    cdf_g0, cdf_g1 = find_interval(cdf)
    with np.errstate(invalid="ignore", divide="ignore"):
        t = (u - cdf_g0) / (cdf_g1 - cdf_g0)
    t = np.clip(np.nan_to_num(t, nan=0.0, posinf=np.finfo(bins.dtype).max, neginf=np.finfo(bins.dtype).min),
                dt(0), dt(1))
    samples = bins_g0 + t * (bins_g1 - bins_g0)
    if return_idx:
        return samples, (mask.sum(axis=-2) - 1).astype(np.int32)
    return samples


def linspace_u(num_samples: int, batch: int, dtype=F32):
    """model_utils.py:355-356."""
    u = np.linspace(0.0, 1.0 - float(np.finfo(np.float32).eps), num_samples).astype(dtype)
    return np.broadcast_to(u, (batch, num_samples)).copy()


def sample_pdf(u, bins, weights, origins, directions, z_vals, idx_grads, jitter):
    """model_utils.py:377-435.  origins/directions/z_vals/idx_grads are the FULL path [B,N,·].

    Returns z_samples [B,S], pos [B,S,3], dir [B,S,3], grad [B,S,3], idx [B,S] int32 (node index).
    """
    z_samples = sorted_piecewise_constant_pdf(u, bins, weights)
    z_samples = np.sort(np.concatenate([z_vals[:, jitter], z_samples], axis=-1), axis=-1)
    B, S = z_samples.shape
    N = z_vals.shape[1]
    idx = np.empty((B, S), np.int32)
    for i in range(B):
        j = np.searchsorted(z_vals[i], z_samples[i], side="left")
        # y = hstack([y[0], y, y[-1]])[j]  ->  max(j-1, 0)   (model_utils.py:415-421)
        idx[i] = np.maximum(j - 1, 0)
    bi = np.arange(B)[:, None]
    rd = directions[bi, idx]
    pos = origins[bi, idx] + rd * (z_samples - z_vals[bi, idx])[..., None]
    return z_samples, pos, rd, idx_grads[bi, idx], idx


# ----------------------------------------------------------------------------
# M1-M3: the model forward                          (rnerf/models.py:219-535)
# ----------------------------------------------------------------------------
class ModelConfig:
    """The subset of NerfModel attributes the hot path reads (models.py:42-90; flag defaults utils.py:136-181)."""

    def __init__(self, ndim, nmin, nmax, near=2.0, far=6.0, num_coarse_samples=64, num_fine_samples=128,
                 num_path_samples=12, min_deg_point=0, max_deg_point=10, deg_view=4, white_bkgd=False,
                 rgb_padding=0.001, sigma_bias=-1.0, use_online_sparsity=False, use_fine_sparsity=False):
        self.ndim = list(ndim); self.nmin = list(nmin); self.nmax = list(nmax)
        self.near = near; self.far = far
        self.num_coarse_samples = num_coarse_samples; self.num_fine_samples = num_fine_samples
        self.num_path_samples = num_path_samples
        self.min_deg_point = min_deg_point; self.max_deg_point = max_deg_point; self.deg_view = deg_view
        self.white_bkgd = white_bkgd; self.rgb_padding = rgb_padding; self.sigma_bias = sigma_bias
        self.use_online_sparsity = use_online_sparsity; self.use_fine_sparsity = use_fine_sparsity
        self.bd_cut_bbox = None   # [xmin,ymin,zmin,xmax,ymax,zmax] when NerfModel.bd_cut_dist is set (models.py:485-497)
        self.use_mask_bbox = False  # models.py:85,261-271,398-408: density_delta *= 1[sample inside the grid's box], both levels

    @property
    def num_samples(self):
        return self.num_coarse_samples * self.num_path_samples             # models.py:121


def safe_log(x, eps=1e-6):
    return np.log(np.maximum(x, x.dtype.type(eps)))


def nerf_forward(cfg: ModelConfig, params: Dict, table, origins, viewdirs, jitter, u_fine=None,
                 dtype=F32, acc_dtype=None, taps: Optional[Dict] = None, noise_std=None, noise_c=None, noise_f=None):
    """NerfModel.__call__ (models.py:220-535), stage="radiance", use_viewdirs=True, sh off.

    noise_std with noise_c [B,N_c] / noise_f [B,N_c+N_f]: add_gaussian_noise (model_utils.py:438-453, models.py:310-317,445-452) with the
    standard-normal draws supplied by the caller (the reference draws random.normal(key, raw_sigma.shape)).

    jitter: int [N_c] = arange(0,N,P) + randint (models.py:240-242), supplied by the caller.
    u_fine: [B,N_f] uniform draws or None (-> linspace, randomized=False).
    Returns (ret, loss_sp) with ret = [(rgb, dist, acc, trans[B,1], trans_rgb_bkgd)] per level.
    """
    N = cfg.num_samples
    ray_pos, ray_dir, ray_dist, idx_data, idx_grad = path_sampler(
        origins, viewdirs, table, cfg.ndim, cfg.nmin, cfg.nmax, cfg.near, cfg.far, N, dtype,
        so3_params=getattr(cfg, "so3_params", None), annealed_alpha=getattr(cfg, "annealed_alpha", 1.0))   # stage "all" when set
    jitter = np.asarray(jitter, np.int64)
    ray_pos_c = ray_pos[:, jitter]; ray_dir_c = ray_dir[:, jitter]; ray_dist_c = ray_dist[:, jitter]
    idx_grad_c = idx_grad[:, jitter]
    samples_enc = pos_enc(ray_pos_c, cfg.min_deg_point, cfg.max_deg_point, dtype)           # models.py:257
    viewdirs_enc = pos_enc(ray_dir_c, 0, cfg.deg_view, dtype)                                # :289-294
    raw_bkgd = simple_mlp(params["bkgd_mlp"], viewdirs_enc[:, -1:], acc_dtype)[:, 0]        # :303
    raw_rgb, raw_sigma = nerf_mlp(params["coarse_mlp"], samples_enc, viewdirs_enc, acc_dtype)  # :305
    if noise_std is not None and noise_c is not None:                                        # :310-317
        raw_sigma = raw_sigma + np.asarray(noise_c, dtype).reshape(raw_sigma.shape) * dtype(noise_std)
    rgb = rgb_activation(raw_rgb, cfg.rgb_padding)
    bkgd = rgb_activation(raw_bkgd, cfg.rgb_padding)
    sigma = sigma_activation(raw_sigma, cfg.sigma_bias)
    def inside_grid_box(p):                                                                  # use_mask_bbox, "small mask bbox" (:261-271,398-408)
        if not getattr(cfg, "use_mask_bbox", False):
            return None
        m = np.ones(p.shape[:2], bool)
        for a in range(3):
            m &= (p[..., a] >= dtype(cfg.nmin[a])) & (p[..., a] <= dtype(cfg.nmax[a]))
        return m

    comp_rgb, disp, acc, weights, alpha, trans, trans_rgb_bkgd = volumetric_rendering(
        rgb, sigma, ray_dist_c, ray_dir_c, cfg.white_bkgd, bkgd, mask_bbox=inside_grid_box(ray_pos_c))   # :341-349
    if cfg.use_online_sparsity:                                                               # :351-357
        mask = np.sqrt(_sum3_sq(idx_grad_c))[..., 0] > dtype(1e-6)
        loss_sp = (mask * safe_log(alpha)).sum() / (np.sum(mask) + 1)
    else:
        loss_sp = 0.0
    ret = [(comp_rgb, disp, acc, trans, trans_rgb_bkgd)]
    if taps is not None:
        taps.update(ray_pos=ray_pos, ray_dir=ray_dir, ray_dist=ray_dist, idx_data=idx_data, idx_grad=idx_grad,
                    raw_rgb_c=raw_rgb, raw_sigma_c=raw_sigma, raw_bkgd=raw_bkgd, bkgd=bkgd,
                    weights_c=weights, alpha_c=alpha, samples_enc_c=samples_enc, viewdirs_enc_c=viewdirs_enc)
    if cfg.num_fine_samples > 0:
        mid = dtype(.5) * (ray_dist_c[..., 1:] + ray_dist_c[..., :-1])                       # :371
        u = linspace_u(cfg.num_fine_samples, origins.shape[0], dtype) if u_fine is None else np.asarray(u_fine, dtype)
        z_f, pos_f, dir_f, grad_f, idx_f = sample_pdf(u, mid, weights[..., 1:-1], ray_pos, ray_dir, ray_dist,
                                                      idx_grad, jitter)                      # :372-384
        samples_enc = pos_enc(pos_f, cfg.min_deg_point, cfg.max_deg_point, dtype)            # :394
        viewdirs_enc = pos_enc(dir_f, 0, cfg.deg_view, dtype)                                # :426
        raw_rgb, raw_sigma = nerf_mlp(params["fine_mlp"], samples_enc, viewdirs_enc, acc_dtype)  # :441
        if noise_std is not None and noise_f is not None:                                    # :445-452
            raw_sigma = raw_sigma + np.asarray(noise_f, dtype).reshape(raw_sigma.shape) * dtype(noise_std)
        rgb = rgb_activation(raw_rgb, cfg.rgb_padding)
        sigma = sigma_activation(raw_sigma, cfg.sigma_bias)
        comp_rgb, disp, acc, w_f, alpha_f, trans, trans_rgb_bkgd = volumetric_rendering(
            rgb, sigma, z_f, dir_f, cfg.white_bkgd, bkgd, mask_bbox=inside_grid_box(pos_f))  # :468-476 (coarse bkgd)
        if getattr(cfg, "bd_cut_bbox", None) is not None:                                    # :479-524
            bmin, bmax = cfg.bd_cut_bbox[:3], cfg.bd_cut_bbox[3:]
            inside = np.ones(pos_f.shape[:2], bool)
            for a in range(3):
                inside &= (pos_f[..., a] >= dtype(bmin[a])) & (pos_f[..., a] <= dtype(bmax[a]))
            mask_bbox = (np.cumsum(inside[:, ::-1], axis=-1) > 0)[:, ::-1]
            trans = volumetric_rendering(rgb, sigma, z_f, dir_f, cfg.white_bkgd, None, mask_bbox=mask_bbox)[5]
            behind = volumetric_rendering(rgb, sigma, z_f, dir_f, cfg.white_bkgd, bkgd, mask_bbox=(1.0 - mask_bbox))[0]
            trans_rgb_bkgd = trans * behind
        if cfg.use_online_sparsity and cfg.use_fine_sparsity:                                # :526-530
            mask = np.sqrt(_sum3_sq(grad_f))[..., 0] > dtype(1e-6)
            loss_sp = loss_sp + (mask * safe_log(alpha_f)).sum() / (np.sum(mask) + 1)
        ret.append((comp_rgb, disp, acc, trans, trans_rgb_bkgd))
        if taps is not None:
            taps.update(z_f=z_f, pos_f=pos_f, dir_f=dir_f, idx_f=idx_f, raw_rgb_f=raw_rgb, raw_sigma_f=raw_sigma,
                        weights_f=w_f, u_fine=u)
    return ret, loss_sp


def forward_envmap(cfg: ModelConfig, params: Dict, viewdirs, dtype=F32, acc_dtype=None):
    """models.py:181-191."""
    enc = pos_enc(np.asarray(viewdirs, dtype), 0, cfg.deg_view, dtype)
    raw = simple_mlp(params["bkgd_mlp"], enc[:, None], acc_dtype)[:, 0]
    return rgb_activation(raw, cfg.rgb_padding)


# ----------------------------------------------------------------------------
# T3: render_image chunking                         (rnerf/utils.py:331-389)
# ----------------------------------------------------------------------------
def render_image(render_fn, origins, viewdirs, chunk=8192, normalize_disp=False):
    """utils.py:331-389 for one device: render_fn(origins[c], viewdirs[c]) -> ret list; takes ret[-1]."""
    H, W = origins.shape[:2]
    o = origins.reshape(H * W, 3); d = viewdirs.reshape(H * W, 3)
    res = []
    for i in range(0, H * W, chunk):
        ret, _ = render_fn(o[i:i + chunk], d[i:i + chunk])
        res.append(ret[-1])
    rgb, distance, acc, _, _ = [np.concatenate(r, axis=0) for r in zip(*res)]
    if normalize_disp:
        distance = (distance - distance.min()) / (distance.max() - distance.min())
    return rgb.reshape(H, W, -1), distance.reshape(H, W, -1), acc.reshape(H, W, -1)


def compute_psnr(mse):
    """utils.py:392-401."""
    return -10.0 * np.log(mse) / np.log(10.0)


# ----------------------------------------------------------------------------
# synthetic parameters (glorot/xavier uniform as model_utils.py:62-63,124; used by fixtures)
# ----------------------------------------------------------------------------
NERF_MLP_SHAPES = [(63, 256), (256, 256), (256, 256), (256, 256), (256, 256), (319, 256), (256, 256), (256, 256),
                   (256, 1), (256, 256), (283, 128), (128, 3)]
BKGD_MLP_SHAPES = [(27, 128), (128, 128), (128, 128), (155, 128), (128, 3)]
SO3_MLP_SHAPES = [(60, 128), (128, 128), (128, 128), (188, 128), (128, 3)]


def init_mlp(rng: np.random.Generator, shapes, bias_scale=0.0, dtype=F32):
    p = {}
    for i, (fi, fo) in enumerate(shapes):
        lim = math.sqrt(6.0 / (fi + fo))
        p[f"Dense_{i}"] = {"kernel": rng.uniform(-lim, lim, (fi, fo)).astype(dtype),
                           "bias": (bias_scale * rng.standard_normal(fo)).astype(dtype)}
    return p


def init_params(seed=0, fine=True, bias_scale=0.0, dtype=F32):
    rng = np.random.default_rng(seed)
    p = {"coarse_mlp": init_mlp(rng, NERF_MLP_SHAPES, bias_scale, dtype),
         "bkgd_mlp": init_mlp(rng, BKGD_MLP_SHAPES, bias_scale, dtype)}
    if fine:
        p["fine_mlp"] = init_mlp(rng, NERF_MLP_SHAPES, bias_scale, dtype)
    return p


# ----------------------------------------------------------------------------
# ray generation                  (rnerf/datasets.py:216-242, :486-518)
# ----------------------------------------------------------------------------
def generate_rays(camtoworld, h, w, focal=None, cam_mat=None, pixel_center=True):
    """Dataset._generate_rays for ONE view, fp32 like the reference's numpy code. -> origins, directions, viewdirs [h,w,3]."""
    pc = F32(0.5 if pixel_center else 0.0)
    c2w = np.asarray(camtoworld, F32)
    if cam_mat is None:      # Blender (:216-242)
        x, y = np.meshgrid(np.arange(w, dtype=F32) + pc, np.arange(h, dtype=F32) + pc, indexing="xy")
        cam = np.stack([(x - F32(w * 0.5)) / F32(focal), -(y - F32(h * 0.5)) / F32(focal), -np.ones_like(x)], axis=-1)
    else:                    # OpenCV (:486-518)
        x, y = np.meshgrid(np.arange(w, dtype=F32), np.arange(h, dtype=F32), indexing="xy")
        cam = np.stack([(x - F32(cam_mat[0][2]) + pc) / F32(cam_mat[0][0]), (y - F32(cam_mat[1][2]) + pc) / F32(cam_mat[1][1]),
                        np.ones_like(x)], axis=-1)
    prod = cam[..., None, :] * c2w[None, None, :3, :3]
    directions = _seqsum(prod, axis=-1)
    origins = np.broadcast_to(c2w[None, None, :3, -1], directions.shape).astype(F32)
    viewdirs = directions / np.sqrt(_seqsum(directions * directions, axis=-1, keepdims=True))
    return origins, directions.astype(F32), viewdirs.astype(F32)


def ray_radii(directions):
    """The `radii` of Dataset._generate_rays (rnerf/datasets.py:230-239; the cone footprint mip.cast_rays takes): distance of each pixel's
    direction to its neighbour in the NEXT IMAGE ROW (the reference's comment says x, its axis is the rows'), the last row repeating the
    one before, times 2 / sqrt(12).  directions [h, w, 3] -> [h, w, 1] float32."""
    d = np.asarray(directions, F32)
    diff = d[:-1] - d[1:]
    dx = np.sqrt(_seqsum(diff * diff, axis=-1))
    dx = np.concatenate([dx, dx[-2:-1]], 0)
    return (dx[..., None] * F32(2) / np.sqrt(F32(12))).astype(F32)


# ----------------------------------------------------------------------------
# voxeliser                       (voxelize_mesh.py:54-106)
# ----------------------------------------------------------------------------
def mesh_contains(verts, faces, pts):
    """Point-in-mesh by crossing parity along +z (fp64, top-left rule on shared edges).  Stands in for pysdf's SDF.contains
    (voxelize_mesh.py:55-66), which is not available offline; identical away from the surface itself."""
    v = np.asarray(verts, np.float64)[np.asarray(faces)]               # [F,3,3]
    A, B, C = v[:, 0], v[:, 1], v[:, 2]
    area = (B[:, 0] - A[:, 0]) * (C[:, 1] - A[:, 1]) - (B[:, 1] - A[:, 1]) * (C[:, 0] - A[:, 0])
    sgn = np.where(area > 0, 1.0, -1.0)
    pts = np.asarray(pts, np.float64)
    inside = np.zeros(len(pts), bool)
    for n, (x, y, z) in enumerate(pts):
        def edge(P, Q):
            e = ((Q[:, 0] - P[:, 0]) * (y - P[:, 1]) - (Q[:, 1] - P[:, 1]) * (x - P[:, 0])) * sgn
            dx, dy = (Q[:, 0] - P[:, 0]) * sgn, (Q[:, 1] - P[:, 1]) * sgn
            return e, (e > 0) | ((e == 0) & ((dy > 0) | ((dy == 0) & (dx < 0))))
        eab, ab = edge(A, B); ebc, bc = edge(B, C); eca, ca = edge(C, A)
        hit = ab & bc & ca & (area != 0)
        with np.errstate(invalid="ignore", divide="ignore"):
            zc = (ebc * A[:, 2] + eca * B[:, 2] + eab * C[:, 2]) / (eab + ebc + eca)
        inside[n] = (np.count_nonzero(hit & (zc > z)) & 1) == 1
    return inside


def voxelize(verts, faces, num_voxels, nmin, nmax, num_samples=4, ior_inside=1.33, ior_outside=1.0):
    """voxelize_mesh.py:70-106 with the containment test above. -> float64 [G,G,G], x slowest."""
    G, K = num_voxels, num_samples
    nmin = np.asarray(nmin, np.float64); nmax = np.asarray(nmax, np.float64)
    lin = np.linspace(0, 1, G)
    off1 = np.linspace(-1, 1, K)
    offset = np.stack(np.meshgrid(off1, off1, off1, indexing="ij"), -1).reshape(-1, 3)
    offset_scale = (2 * (nmax - nmin))[None] / (G - 1) * 0.5
    out = np.zeros((G, G, G))
    for i in range(G):
        for j in range(G):
            for k in range(G):
                c = np.array([lin[i], lin[j], lin[k]]) * (nmax - nmin) + nmin
                ior = np.where(mesh_contains(verts, faces, c[None] + offset * offset_scale), ior_inside, ior_outside)
                out[i, j, k] = np.mean(ior)
    return out


def voxelize_counts(verts, faces, num_voxels, nmin, nmax, num_samples=4, return_samples=False):
    """The same definition as voxelize() (crossing parity along +z at the K^3 sub-samples of voxelize_mesh.py:70-106), evaluated per
    (triangle, xy sample column) pair instead of per point so that a 128^3 x 4^3 grid over a 55 k-triangle mesh takes seconds.
    -> int32 [G,G,G]: number of sub-samples inside the mesh (voxelize() == (n*inside + (K^3-n)*outside) / K^3)."""
    G, K = int(num_voxels), int(num_samples)
    nmin = np.asarray(nmin, np.float64); nmax = np.asarray(nmax, np.float64)
    lin = np.linspace(0, 1, G)
    off1 = np.linspace(-1, 1, K)
    scale = (2 * (nmax - nmin)) / (G - 1) * 0.5
    # sample coordinate of (voxel i, sub-sample a) along each axis, formed exactly like voxelize(): (lin*(max-min)+min) + off*scale
    coord = [((lin * (nmax[ax] - nmin[ax]) + nmin[ax])[:, None] + (off1 * scale[ax])[None, :]).reshape(-1) for ax in range(3)]
    xs, ys, zs = coord
    zorder = np.argsort(zs, kind="stable"); zsorted = zs[zorder]
    v = np.asarray(verts, np.float64)[np.asarray(faces)]
    A, B, C = v[:, 0], v[:, 1], v[:, 2]
    area = (B[:, 0] - A[:, 0]) * (C[:, 1] - A[:, 1]) - (B[:, 1] - A[:, 1]) * (C[:, 0] - A[:, 0])
    sgn = np.where(area > 0, 1.0, -1.0)
    xo = np.argsort(xs, kind="stable"); yo = np.argsort(ys, kind="stable")
    xsort, ysort = xs[xo], ys[yo]
    x_lo = np.searchsorted(xsort, v[:, :, 0].min(1), "left"); x_hi = np.searchsorted(xsort, v[:, :, 0].max(1), "right")
    y_lo = np.searchsorted(ysort, v[:, :, 1].min(1), "left"); y_hi = np.searchsorted(ysort, v[:, :, 1].max(1), "right")
    nx, ny = x_hi - x_lo, y_hi - y_lo
    keep = (nx > 0) & (ny > 0) & (area != 0)
    tri = np.nonzero(keep)[0]
    cnt = (nx * ny)[tri]
    start = np.concatenate([[0], np.cumsum(cnt)])
    t_of = np.repeat(tri, cnt)                                   # triangle of every candidate (triangle, column) pair
    local = np.arange(start[-1]) - np.repeat(start[:-1], cnt)
    ix = xo[x_lo[t_of] + local // ny[t_of]]; iy = yo[y_lo[t_of] + local % ny[t_of]]
    x, y = xs[ix], ys[iy]
    s = sgn[t_of]

    def edge(P, Q):
        e = ((Q[t_of, 0] - P[t_of, 0]) * (y - P[t_of, 1]) - (Q[t_of, 1] - P[t_of, 1]) * (x - P[t_of, 0])) * s
        dx, dy = (Q[t_of, 0] - P[t_of, 0]) * s, (Q[t_of, 1] - P[t_of, 1]) * s
        return e, (e > 0) | ((e == 0) & ((dy > 0) | ((dy == 0) & (dx < 0))))
    eab, ab = edge(A, B); ebc, bc = edge(B, C); eca, ca = edge(C, A)
    hit = ab & bc & ca
    with np.errstate(invalid="ignore", divide="ignore"):
        zc = (ebc * A[t_of, 2] + eca * B[t_of, 2] + eab * C[t_of, 2]) / (eab + ebc + eca)
    ix, iy, zc = ix[hit], iy[hit], zc[hit]
    m = np.searchsorted(zsorted, zc, "left")                      # the first m samples (in z order) lie strictly below the crossing
    flips = np.zeros((G * K * G * K, G * K + 1), np.int16)
    col = ix.astype(np.int64) * (G * K) + iy
    np.add.at(flips, (col, np.zeros_like(m)), 1)
    np.add.at(flips, (col, m), -1)
    above = np.cumsum(flips[:, :-1], axis=1, dtype=np.int16)     # crossings above each z-sample (z order)
    inside_sorted = (above & 1).astype(np.int8)
    inside = np.empty_like(inside_sorted)
    inside[:, zorder] = inside_sorted
    if return_samples:
        return inside.reshape(G * K, G * K, G * K)                 # [x sample][y sample][z sample]
    return inside.reshape(G, K, G, K, G, K).sum(axis=(1, 3, 5), dtype=np.int32)


def voxelize_counts_robust(verts, faces, num_voxels, nmin, nmax, num_samples=4):
    """Three-axis majority containment (not in the reference: pysdf casts ONE parity ray per point in a random frame,
    sdf/src/sdf.cpp:156-168,270-322): the crossing-parity test along +z, +x and +y, a sample counts as inside when at least two agree.
    Equal to voxelize_counts() on watertight meshes; well-defined where a hole lets one of the rays escape."""
    G, K = int(num_voxels), int(num_samples)
    v = np.asarray(verts, np.float64); lo = np.asarray(nmin, np.float64); hi = np.asarray(nmax, np.float64)
    votes = np.zeros((G * K,) * 3, np.int8)
    for perm, back in (((0, 1, 2), (0, 1, 2)), ((1, 2, 0), (2, 0, 1)), ((2, 0, 1), (1, 2, 0))):
        ins = voxelize_counts(v[:, perm], faces, G, lo[list(perm)], hi[list(perm)], K, return_samples=True)    # indexed by the permuted axes
        votes += ins.transpose(back)                                # back to [X][Y][Z]
    return (votes >= 2).reshape(G, K, G, K, G, K).sum(axis=(1, 3, 5), dtype=np.int32)


def counts_to_ior(counts, num_samples=4, ior_inside=1.33, ior_outside=1.0):
    """voxelize_mesh.py:60,105: mean over the K^3 sub-samples of where(inside, 1.33, 1.0), float64."""
    K3 = int(num_samples) ** 3
    c = np.asarray(counts, np.float64)
    return (c * ior_inside + (K3 - c) * ior_outside) / K3
