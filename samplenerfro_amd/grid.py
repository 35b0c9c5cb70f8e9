"""IoR grid loading and preparation: the drivers' grid block (train.py:209-225, eval.py:69-85) on the device.

On-disk format (`<data_dir>/<voxel_grid>/mesh.pkl`, written by voxelize_mesh.py:109-116 / calib/make_visual_hull.py:137-145):
    {"data": float[G^3, 1] (x slowest), "extent": float, "min_point": [3], "max_point": [3], "num_voxels": G}
"""
from __future__ import annotations

import pickle
from typing import Dict, Sequence, Tuple

import numpy as np
import torch

from . import ops

# scene-name substrings that select the 0.33 IoR scale (train.py:220, eval.py:80); everything else uses 0.5
_RI_033 = ("glass", "wineglass", "pen", "torus_skydome-bkgd_cycles", "dolphin", "lighthouse", "yellow")


def refractive_index_for(config_name: str) -> float:
    return 0.33 if any(k in (config_name or "") for k in _RI_033) else 0.5


def load_mesh_pkl(path: str) -> Tuple[np.ndarray, list, list, list]:
    """-> (data float64 [G^3,1], ndim, nmin, nmax) with the extent / min_point,max_point rule of train.py:211-217."""
    with open(path, "rb") as f:
        d = pickle.load(f)
    return mesh_dict_to_grid(d)


def mesh_dict_to_grid(d: Dict) -> Tuple[np.ndarray, list, list, list]:
    if d["extent"] > 0:
        nmin = [-float(d["extent"])] * 3
        nmax = [float(d["extent"])] * 3
    else:
        nmin = [float(v) for v in d["min_point"]]
        nmax = [float(v) for v in d["max_point"]]
    g = int(d["num_voxels"])
    data = np.asarray(d["data"], np.float64).reshape(-1, 1)
    if data.shape[0] != g ** 3:
        raise ValueError(f"mesh.pkl: data has {data.shape[0]} voxels, num_voxels={g}")
    return data, [g, g, g], nmin, nmax


def prepare_grid(data, ndim: Sequence[int], config_name: str, kernel_size: int, kernel_sigma: float, device) -> torch.Tensor:
    """(data - 1) * ri / 0.33 + 1 in float64 (train.py:222), then conv3d_normal if kernel_size > 0 (:221-225). -> [Gx,Gy,Gz] f32."""
    ri = refractive_index_for(config_name)
    scaled = (np.asarray(data, np.float64) - 1.0) * ri / 0.33 + 1.0
    g = torch.from_numpy(scaled.astype(np.float32).reshape(tuple(ndim))).to(device)
    if kernel_size > 0:
        g = ops.grid_prefilter(g, kernel_size, kernel_sigma)
    return g
