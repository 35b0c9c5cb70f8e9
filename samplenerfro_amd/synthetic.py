"""Synthetic workloads: the BASELINE.json configs restated as concrete seeded inputs.

No scene data or checkpoints are available offline, so every measured configuration is synthetic: rays on a sphere
looking into the unit ball, an analytic IoR grid, glorot-uniform weights.  numpy only (shared by bench.py, the tests
and __graft_entry__.smoke()).
"""
from __future__ import annotations

import math
from typing import Dict, Tuple

import numpy as np

SEED = 20200823   # the reference's PRNG seed (train.py:187)

NERF_MLP_SHAPES = [(63, 256), (256, 256), (256, 256), (256, 256), (256, 256), (319, 256), (256, 256), (256, 256),
                   (256, 1), (256, 256), (283, 128), (128, 3)]
BKGD_MLP_SHAPES = [(27, 128), (128, 128), (128, 128), (155, 128), (128, 3)]

# name -> workload.  ri: IoR scale chosen by scene name (train.py:220); data_in: raw voxel value inside.
CONFIGS: Dict[str, dict] = {
    "example": dict(B=512, S=64, F=128, P=12, G=128, extent=1.5, near=2.0, far=6.0, ksize=3, ksigma=1.0, ri=0.5, radius=0.5),
    "ship_straight": dict(B=4096, S=128, F=0, P=12, G=512, extent=1.5, near=2.0, far=6.0, ksize=0, ksigma=0.0, ri=0.5, radius=0.0),
    "ship_refractive": dict(B=4096, S=128, F=0, P=12, G=512, extent=1.5, near=2.0, far=6.0, ksize=9, ksigma=3.0, ri=0.5, radius=0.6),
    "dolphin_train": dict(B=4096, S=64, F=128, P=12, G=256, extent=0.2, near=0.2, far=1.2, ksize=5, ksigma=1.0, ri=0.33, radius=0.1),
    "glass_frame": dict(B=640000, S=256, F=0, P=24, G=384, extent=1.75, near=0.2, far=14.0, ksize=5, ksigma=3.0, ri=0.33, radius=0.8),
}


def sphere_rays(B: int, seed: int = SEED, radius: float = 4.0, target_radius: float = 1.0) -> Tuple[np.ndarray, np.ndarray]:
    """Origins uniform on the sphere |o| = radius, unit viewdirs toward a uniform point of the ball |x| < target_radius."""
    rng = np.random.default_rng(seed)
    o = rng.standard_normal((B, 3))
    o = radius * o / np.linalg.norm(o, axis=-1, keepdims=True)
    t = rng.standard_normal((B, 3))
    t = t / np.linalg.norm(t, axis=-1, keepdims=True) * (target_radius * rng.uniform(0, 1, (B, 1)) ** (1.0 / 3.0))
    d = t - o
    d = d / np.linalg.norm(d, axis=-1, keepdims=True)
    return o.astype(np.float32), d.astype(np.float32)


def sphere_grid(G: int, extent: float, radius: float, inside: float = 1.33) -> np.ndarray:
    """Raw voxeliser output (voxelize_mesh.py:99-107 writes mean IoR in [1, 1.33]): a solid sphere, x slowest. [G,G,G] f64."""
    if radius <= 0:
        return np.ones((G, G, G), np.float64)
    a = np.linspace(-extent, extent, G)
    x, y, z = np.meshgrid(a, a, a, indexing="ij")
    # smooth (supersampling-like) boundary one voxel wide
    h = 2.0 * extent / (G - 1)
    d = (radius - np.sqrt(x * x + y * y + z * z)) / h
    return 1.0 + (inside - 1.0) * np.clip(d + 0.5, 0.0, 1.0)


def scale_ior(data: np.ndarray, ri: float) -> np.ndarray:
    """train.py:222: (data - 1) * ri / 0.33 + 1 (float64)."""
    return (np.asarray(data, np.float64) - 1.0) * ri / 0.33 + 1.0


def init_mlp_flat(rng: np.random.Generator, shapes, bias_scale: float = 0.0) -> np.ndarray:
    """glorot/xavier-uniform kernels (rnerf/model_utils.py:62-63,124); flat fp32, flax creation order."""
    parts = []
    for fi, fo in shapes:
        lim = math.sqrt(6.0 / (fi + fo))
        parts.append(rng.uniform(-lim, lim, (fi, fo)).astype(np.float32).reshape(-1))
        parts.append((bias_scale * rng.standard_normal(fo)).astype(np.float32))
    return np.concatenate(parts)


def init_params_flat(seed: int = 0, fine: bool = True, bias_scale: float = 0.0) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(seed)
    p = {"coarse_mlp": init_mlp_flat(rng, NERF_MLP_SHAPES, bias_scale), "bkgd_mlp": init_mlp_flat(rng, BKGD_MLP_SHAPES, bias_scale)}
    if fine:
        p["fine_mlp"] = init_mlp_flat(rng, NERF_MLP_SHAPES, bias_scale)
    return p


def flat_to_np_tree(flat: np.ndarray, shapes) -> Dict[str, Dict[str, np.ndarray]]:
    tree, off = {}, 0
    for k, (i, o) in enumerate(shapes):
        tree[f"Dense_{k}"] = {"kernel": flat[off:off + i * o].reshape(i, o), "bias": flat[off + i * o:off + i * o + o]}
        off += i * o + o
    return tree


def params_tree(flat: Dict[str, np.ndarray]) -> Dict[str, dict]:
    t = {"coarse_mlp": flat_to_np_tree(flat["coarse_mlp"], NERF_MLP_SHAPES),
         "bkgd_mlp": flat_to_np_tree(flat["bkgd_mlp"], BKGD_MLP_SHAPES)}
    if "fine_mlp" in flat:
        t["fine_mlp"] = flat_to_np_tree(flat["fine_mlp"], NERF_MLP_SHAPES)
    return t
