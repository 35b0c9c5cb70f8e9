"""torch (CPU, float64) restatement of the DIFFERENTIABLE part of the path, for gradient checks (TEST INFRASTRUCTURE).

Follows rnerf/models.py:334-349 (activations), rnerf/model_utils.py:247-309 (volumetric_rendering) and the radiance-stage
terms of train.py:75-162 (loss_fn); jax.value_and_grad (train.py:164) is replaced by torch.autograd.  Parity unpinned
(see oracle/__init__.py); validated against oracle/ref_np.py in tests/test_oracle_kat.py.
"""
from __future__ import annotations

import torch


def activations(raw, rgb_padding=0.001, sigma_bias=-1.0):
    rgb = torch.sigmoid(raw[..., :3]) * (1 + 2 * rgb_padding) - rgb_padding
    sigma = torch.nn.functional.softplus(raw[..., 3] + sigma_bias)
    return rgb, sigma


def volumetric_rendering(rgb, sigma, t_vals, dirs, bkgd, mask=None, white_bkgd=False):
    """rgb [B,S,3], sigma [B,S], t_vals [B,S], dirs [B,S,3], bkgd [B,3] -> comp_rgb, acc, weights, trans [B,1], trans*sg(bkgd).
    mask [B,S]: density_delta *= mask_bbox (rnerf/model_utils.py:275-276)."""
    t_dists = torch.cat([t_vals[..., 1:] - t_vals[..., :-1], torch.full_like(t_vals[..., :1], 1e-3)], -1)
    delta = t_dists * torch.linalg.norm(dirs, dim=-1)
    dd = sigma * delta
    if mask is not None:
        dd = dd * mask
    alpha = 1 - torch.exp(-dd)
    trans = torch.exp(-torch.cat([torch.zeros_like(dd[..., :1]), torch.cumsum(dd, -1)], -1))
    weights = alpha * trans[..., :-1]
    comp = (weights[..., None] * rgb).sum(-2) + trans[..., -1:] * bkgd
    if white_bkgd:
        comp = comp + (1.0 - weights.sum(-1)[..., None])
    return comp, weights.sum(-1), weights, trans[..., -1:], trans[..., -1:] * bkgd.detach()


def bd_cut_pair(rgb, sigma, t_vals, dirs, bkgd, pos, bbox):
    """rnerf/models.py:498-524: (trans, trans_rgb_bkgd) of the last level when bd_cut_dist is set.  pos [B,S,3], bbox = min xyz + max xyz."""
    lo = torch.tensor(bbox[:3], dtype=pos.dtype); hi = torch.tensor(bbox[3:], dtype=pos.dtype)
    inside = ((pos >= lo) & (pos <= hi)).all(-1).to(rgb.dtype)
    m = (torch.flip(torch.cumsum(torch.flip(inside, [-1]), -1), [-1]) > 0).to(rgb.dtype)
    _, _, _, trans, _ = volumetric_rendering(rgb, sigma, t_vals, dirs, torch.ones_like(bkgd), mask=m)
    behind, _, _, _, _ = volumetric_rendering(rgb, sigma, t_vals, dirs, bkgd, mask=1.0 - m)
    return trans, trans * behind


def radiance_loss(levels, pixels, bg_weight=0.0, annealed_alpha=1.0):
    """levels: list of (comp_rgb, trans, trans_rgb_bkgd), coarse first (train.py:87-110). Returns (loss, parts)."""
    rgb, trans, tb = levels[-1]
    loss = ((rgb - pixels) ** 2).mean()
    parts = {"loss": loss}
    total = loss
    if bg_weight > 0:
        mask = (trans > 0.5).to(rgb.dtype)
        loss_bg = float(annealed_alpha > 0) * (mask * (tb - pixels).abs()).sum() / (mask.sum() + 1)
        parts["loss_bg"] = loss_bg
        total = total + bg_weight * loss_bg
    if len(levels) > 1:
        loss_c = ((levels[0][0] - pixels) ** 2).mean()
        parts["loss_c"] = loss_c
        total = total + loss_c
    return total, parts


NERF_MLP_SHAPES = [(63, 256), (256, 256), (256, 256), (256, 256), (256, 256), (319, 256), (256, 256), (256, 256),
                   (256, 1), (256, 256), (283, 128), (128, 3)]


def nerf_mlp(flat, x, cond):
    """rnerf/model_utils.py:30-90 on a flat parameter vector (flax order). x [R,63], cond [R,27] -> raw [R,4] = (rgb, sigma)."""
    ps, off = [], 0
    for i, o in NERF_MLP_SHAPES:
        ps.append((flat[off:off + i * o].view(i, o), flat[off + i * o:off + i * o + o]))
        off += i * o + o
    h = x
    for i in range(8):
        h = torch.relu(h @ ps[i][0] + ps[i][1])
        if i == 4:
            h = torch.cat([h, x], -1)
    sigma = h @ ps[8][0] + ps[8][1]
    bott = h @ ps[9][0] + ps[9][1]
    v = torch.relu(torch.cat([bott, cond], -1) @ ps[10][0] + ps[10][1])
    rgb = v @ ps[11][0] + ps[11][1]
    return torch.cat([rgb, sigma], -1)


BKGD_MLP_SHAPES = [(27, 128), (128, 128), (128, 128), (155, 128), (128, 3)]


def bkgd_mlp(flat, enc, rgb_padding=0.001):
    """rnerf/model_utils.py:93-140 as built at rnerf/models.py:116-118 + rgb activation (:336-337). enc [n,27] -> [n,3]."""
    ps, off = [], 0
    for i, o in BKGD_MLP_SHAPES:
        ps.append((flat[off:off + i * o].view(i, o), flat[off + i * o:off + i * o + o]))
        off += i * o + o
    h = enc
    for i in range(4):
        h = torch.relu(h @ ps[i][0] + ps[i][1])
        if i == 2:
            h = torch.cat([h, enc], -1)
    raw = h @ ps[4][0] + ps[4][1]
    return torch.sigmoid(raw) * (1 + 2 * rgb_padding) - rgb_padding


# ----------------------------------------------------------------------------------------------------------------------------------
# Stage "all*": the differentiable march (rnerf/eikonal_utils.py:29-49,100-124), VoxMLP.__call__ with so3_mlp + Rodrigues
# (rnerf/ior_utils.py:188-223,269-312) and annealed_pos_enc (rnerf/model_utils.py:218-245), in torch float64 with autograd: the
# reference for the gradient w.r.t. the so3_mlp parameters (train.py:302-310 makes path_sampler trainable).
# ----------------------------------------------------------------------------------------------------------------------------------
SO3_MLP_SHAPES = [(60, 128), (128, 128), (128, 128), (188, 128), (128, 3)]


def pos_enc_t(x, L):
    """model_utils.py:187-214 (non-legacy order) on a torch tensor [..., 3]."""
    scales = 2.0 ** torch.arange(L, dtype=x.dtype)
    xb = (x[..., None, :] * scales[:, None]).reshape(*x.shape[:-1], -1)
    return torch.cat([x, torch.sin(torch.cat([xb, xb + 0.5 * torch.pi], -1))], -1)


def annealed_pos_enc_t(x, alpha, L=10):
    """model_utils.py:236-245: per degree [sin(3) * w_d | sin(. + pi/2)(3) * w_d], w = cosine_easing_window(0, L-1, L, alpha)."""
    scales = 2.0 ** torch.arange(L, dtype=x.dtype)
    xb = x[..., None, :] * scales[:, None]
    bands = torch.linspace(0, L - 1, L, dtype=x.dtype)
    w = 0.5 * (1 + torch.cos(torch.pi * torch.clamp(alpha - bands, 0.0, 1.0) + torch.pi))
    four = torch.cat([torch.sin(xb) * w[:, None], torch.sin(xb + 0.5 * torch.pi) * w[:, None]], -1)
    return four.reshape(*x.shape[:-1], -1)


def so3_mlp(flat, enc):
    """model_utils.py:93-140 as built at ior_utils.py:148-152: 60 -> 128 x 4 (inputs concatenated after layer 2) -> 3."""
    ps, off = [], 0
    for i, o in SO3_MLP_SHAPES:
        ps.append((flat[off:off + i * o].view(i, o), flat[off + i * o:off + i * o + o]))
        off += i * o + o
    h = enc
    for i in range(4):
        h = torch.relu(h @ ps[i][0] + ps[i][1])
        if i == 2:
            h = torch.cat([h, enc], -1)
    return h @ ps[4][0] + ps[4][1]


def safe_norm_t(x, eps=1e-6):
    return torch.sqrt(torch.clamp((x * x).sum(-1, keepdim=True), min=eps))


def linear3_t(table, pts, ndim, nmin, nmax):
    """ior_utils.py:188-223 on a torch table [G^3, 4]; differentiable w.r.t. pts through the weights (floor carries no gradient)."""
    nd = [(float(nmax[i]) - float(nmin[i])) / (ndim[i] - 1.0) for i in range(3)]
    x = (pts[..., 0] - nmin[0]) / nd[0]; y = (pts[..., 1] - nmin[1]) / nd[1]; z = (pts[..., 2] - nmin[2]) / nd[2]
    x0 = torch.floor(x.detach()).long(); y0 = torch.floor(y.detach()).long(); z0 = torch.floor(z.detach()).long()
    xd = (x - x0)[..., None]; yd = (y - y0)[..., None]; zd = (z - z0)[..., None]
    cl = lambda v, n: torch.clamp(v, 0, n - 1)
    x1 = cl(x0 + 1, ndim[0]); y1 = cl(y0 + 1, ndim[1]); z1 = cl(z0 + 1, ndim[2])
    x0 = cl(x0, ndim[0]); y0 = cl(y0, ndim[1]); z0 = cl(z0, ndim[2])
    s1, s2 = ndim[1] * ndim[2], ndim[2]
    d = table
    c00 = d[s1 * x0 + s2 * y0 + z0] * (1 - xd) + d[s1 * x1 + s2 * y0 + z0] * xd
    c01 = d[s1 * x0 + s2 * y0 + z1] * (1 - xd) + d[s1 * x1 + s2 * y0 + z1] * xd
    c10 = d[s1 * x0 + s2 * y1 + z0] * (1 - xd) + d[s1 * x1 + s2 * y1 + z0] * xd
    c11 = d[s1 * x0 + s2 * y1 + z1] * (1 - xd) + d[s1 * x1 + s2 * y1 + z1] * xd
    c0 = c00 * (1 - yd) + c10 * yd
    c1 = c01 * (1 - yd) + c11 * yd
    return c0 * (1 - zd) + c1 * zd


def vox_mlp_call_t(table, so3_flat, pts, ndim, nmin, nmax, annealed_alpha=1.0):
    ret = linear3_t(table, pts, ndim, nmin, nmax)
    n, g = ret[:, :1], ret[:, 1:]
    raw = so3_mlp(so3_flat, annealed_pos_enc_t(pts, annealed_alpha * 10.0))
    theta = safe_norm_t(raw)
    e = raw / theta
    a = safe_norm_t(g)
    v = g / a
    pred = a * (torch.cos(theta) * v + torch.sin(theta) * torch.cross(e, v, dim=-1) + (1 - torch.cos(theta)) * (e * v).sum(-1, keepdim=True) * e)
    return n, g, pred


def path_sampler_all(origins, viewdirs, table, so3_flat, ndim, nmin, nmax, near, far, num_samples, annealed_alpha=1.0):
    """PathSampler.__call__ with stage "all": -> ray_pos [B,N,3], ray_dir [B,N,3] (safe-normalised), ray_dist [B,N] (stop_gradient)."""
    step = (float(far) - float(near)) / (num_samples - 1)
    rp = origins + near * viewdirs
    rd = viewdirs.clone()
    rt = torch.full((origins.shape[0], 1), float(near), dtype=origins.dtype)
    pos, dirs, dist = [], [], []
    for _ in range(num_samples):
        pos.append(rp); dirs.append(rd); dist.append(rt[:, 0])
        n, g, pred = vox_mlp_call_t(table, so3_flat, rp, ndim, nmin, nmax, annealed_alpha)
        grad = torch.where(torch.linalg.norm(g, dim=-1, keepdim=True) > 1e-3, pred, g)
        nrp = rp + step / n * rd
        rd = rd + step * grad
        rt = rt + torch.linalg.norm(rp - nrp, dim=-1, keepdim=True)
        rp = nrp
    dirs = torch.stack(dirs, 1)
    return torch.stack(pos, 1), dirs / safe_norm_t(dirs), torch.stack(dist, 1).detach()
