"""Host-side mirror of the parts of rnerf/utils.py that sit on the boundary of the hot path.

Rays / Stats / namedtuple_map / render_image / compute_psnr / learning_rate_decay / default flags.  Everything else in
the reference's utils.py (absl flags, gin, SSIM, image IO) is out of scope.
"""
from __future__ import annotations

import collections
import dataclasses
import math
import types
from typing import Callable, Optional

import numpy as np
import torch

from . import prng

# rnerf/utils.py:67 — only origins and viewdirs are read by the path (rnerf/models.py:235-236)
Rays = collections.namedtuple("Rays", ("origins", "directions", "viewdirs", "radii"))


def namedtuple_map(fn, tup):
    """rnerf/utils.py:70-72."""
    return type(tup)(*map(fn, tup))


@dataclasses.dataclass
class Stats:
    """rnerf/utils.py:47-64."""
    loss: float = 0.0
    psnr: float = 0.0
    loss_c: float = 0.0
    psnr_c: float = 0.0
    weight_l2: float = 0.0
    loss_nrm: float = 0.0
    loss_sp: float = 0.0
    annealing_rate: float = 0.0
    loss_bg: float = 0.0
    loss_bg_c: float = 0.0
    loss_bg_smooth: float = 0.0
    coarse_alpha_target: float = 0.0
    fine_alpha_target: float = 0.0


def default_flags(**overrides) -> types.SimpleNamespace:
    """The hot-path subset of rnerf/utils.py:87-245 (flag defaults), overridable like the YAML layer (:248-257)."""
    f = dict(
        config=None, stage="radiance", near=2.0, far=6.0, net_depth=8, net_width=256, net_depth_condition=1,
        net_width_condition=128, weight_decay_mult=0.0, skip_layer=4, num_rgb_channels=3, num_sigma_channels=1,
        randomized=True, min_deg_point=0, max_deg_point=10, deg_view=4, num_coarse_samples=64, num_fine_samples=128,
        use_viewdirs=True, sh_deg=-1, sh_direnc_deg=-1, noise_std=None, lindisp=False, net_activation="relu",
        rgb_activation="sigmoid", sigma_activation="softplus", legacy_posenc_order=False, white_bkgd=True,
        batch_size=1024, lr_init=5e-4, lr_final=5e-6, lr_delay_steps=2500, lr_delay_mult=0.01, grad_max_norm=0.0,
        grad_max_val=0.0, max_steps=1000000, num_path_samples=8, sparsity_weight=0.0, use_fine_sparsity=False,
        use_online_sparsity=True, normal_loss_weight=0.0, normal_smooth_weight=0.0, beta_weight=0.0, bg_weight=0.0,
        bg_smooth_weight=0.0, bg_patch_size=0, chunk=8192,
        backward_precision="f16x3",      # not a reference flag: arithmetic of the HIP backward (train.backward_mode)
        range_retry="lag",               # not a reference flag: re-run a step whose f16-based arithmetic left its range (train.train_step):
                                         # "lag" = decided two steps later (no per-step host read), True = in place (one read per step), False = never
    )
    f.update(overrides)
    return types.SimpleNamespace(**f)


def compute_psnr(mse):
    """rnerf/utils.py:392-401."""
    if isinstance(mse, torch.Tensor):
        return -10.0 / math.log(10.0) * torch.log(mse)
    return -10.0 / math.log(10.0) * math.log(mse)


def learning_rate_decay(step, lr_init, lr_final, max_steps, lr_delay_steps=0, lr_delay_mult=1, lr_start_steps=0):
    """rnerf/utils.py:490-528."""
    if lr_delay_steps > 0:
        delay_rate = lr_delay_mult + (1 - lr_delay_mult) * math.sin(0.5 * math.pi * min(max(step / lr_delay_steps, 0), 1))
    else:
        delay_rate = 1.0
    start_rate = min(max(step - lr_start_steps, 0), 1)
    t = min(max(max(step - lr_start_steps, 0) / (max_steps - lr_start_steps), 0), 1)
    log_lerp = math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)
    return start_rate * delay_rate * log_lerp


def render_image(render_fn: Callable, rays: Rays, rng, normalize_disp: bool, chunk: int = 8192, model=None):
    """rnerf/utils.py:331-389.  `render_fn(key_0, key_1, chunk_rays)` -> (ret, loss_sp); the fine tuple ret[-1] is kept.

    rays: Rays of [H, W, ...] tensors.  Returns (rgb [H,W,3], dist [H,W,1], acc [H,W,1]).  The same key pair is used for
    every chunk (:350).  With `model=` the chunks are software-pipelined (march of chunk k+1 beside the MLP of chunk k) and
    render_fn must accept `path=`.  There is no device padding/sharding here: one process renders on one GPU; multi-GPU eval gives
    each rank a contiguous block of rows and needs no collective (samplenerfro_amd.distributed.render_image_sharded).
    """
    height, width = rays[0].shape[:2]
    num_rays = height * width
    rays = namedtuple_map(lambda r: None if r is None else r.reshape((num_rays, -1)), rays)
    _unused, key_0, key_1 = prng.split(rng, 3)
    results = []
    if model is not None:
        # pipelined form: `render_fn(key_0, key_1, chunk_rays, path=handle)`; the march of chunk k+1 runs on the model's side
        # stream while chunk k is in its MLP phase (NerfModel.prefetch_path)
        starts = list(range(0, num_rays, chunk))
        get = lambda i: namedtuple_map(lambda r: None if r is None else r[i:i + chunk], rays)
        nxt_rays = get(starts[0]); handle = model.prefetch_path(nxt_rays, sync_inputs=True)
        for k, i in enumerate(starts):
            cur_rays, cur_handle = nxt_rays, handle
            if k + 1 < len(starts):
                nxt_rays = get(starts[k + 1]); handle = model.prefetch_path(nxt_rays, sync_inputs=False)
            results.append(render_fn(key_0, key_1, cur_rays, path=cur_handle)[0][-1])
    else:
        for i in range(0, num_rays, chunk):
            chunk_rays = namedtuple_map(lambda r: None if r is None else r[i:i + chunk], rays)
            chunk_results = render_fn(key_0, key_1, chunk_rays)[0][-1]
            results.append(chunk_results)
    rgb, distance, acc, _trans, _trans_rgb_bkgd = [torch.cat(r, dim=0) for r in zip(*results)]
    if normalize_disp:
        distance = (distance - distance.min()) / (distance.max() - distance.min())
    return (rgb.reshape((height, width, -1)), distance.reshape((height, width, -1)), acc.reshape((height, width, -1)))
