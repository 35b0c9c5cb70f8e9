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
