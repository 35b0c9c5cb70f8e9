"""The voxeliser on the device: voxelize_mesh.py:54-116 without the G^3 Python loop and without pysdf.

    data, ndim, nmin, nmax = voxelize(verts, faces, num_voxels=128, extent=1.5, num_samples=4, device="cuda:0")
    save_mesh_pkl(path, data, extent=1.5, num_voxels=128)        # the dict train.py:209-217 / grid.load_mesh_pkl read

`data` is the raw voxel value (mean IoR in [1, 1.33]) as float32 [G,G,G], x slowest — feed it to grid.prepare / ops.grid_prefilter.
Point-in-mesh = parity of surface crossings along +z (watertight meshes), evaluated in fp64 with the top-left rule on shared
edges; pysdf's robust mode may differ on points that lie exactly on the surface (a set of measure zero).
"""
from __future__ import annotations

import ctypes as C
import pickle
from typing import Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib
from ._lib import Grid, check, current_stream, ptr


def load_obj(path: str) -> Tuple[np.ndarray, np.ndarray]:
    """Minimal Wavefront OBJ reader: v / f records, polygons fan-triangulated, 1-based (or negative) indices."""
    verts, faces = [], []
    with open(path) as f:
        for line in f:
            p = line.split()
            if not p:
                continue
            if p[0] == "v":
                verts.append([float(p[1]), float(p[2]), float(p[3])])
            elif p[0] == "f":
                idx = [int(t.split("/")[0]) for t in p[1:]]
                idx = [i - 1 if i > 0 else len(verts) + i for i in idx]
                for k in range(1, len(idx) - 1):
                    faces.append([idx[0], idx[k], idx[k + 1]])
    return np.asarray(verts, np.float64), np.asarray(faces, np.int32)


def build_xy_bins(verts: np.ndarray, faces: np.ndarray, lo: Sequence[float], hi: Sequence[float], num_bins: int):
    """CSR lists of the triangles whose xy bounding box overlaps each cell of a num_bins^2 grid over [lo, hi] (host, numpy)."""
    v = verts[faces]                                                  # [F,3,3]
    size = (np.asarray(hi[:2], np.float64) - np.asarray(lo[:2], np.float64)) / num_bins
    size = np.maximum(size, 1e-300)
    # one cell of slack on both sides: the device finds a column's cell with floor((x - x0) * (1 / size)), which may differ by one from
    # this floor((x - x0) / size) for a column exactly on a cell border (marching-cubes vertices sit on such sample coordinates)
    b0 = np.clip(np.floor((v[:, :, :2].min(1) - lo[:2]) / size).astype(np.int64) - 1, 0, num_bins - 1)
    b1 = np.clip(np.floor((v[:, :, :2].max(1) - lo[:2]) / size).astype(np.int64) + 1, 0, num_bins - 1)
    outside = (v[:, :, 0].max(1) < lo[0]) | (v[:, :, 0].min(1) > hi[0]) | (v[:, :, 1].max(1) < lo[1]) | (v[:, :, 1].min(1) > hi[1])
    cells, tris = [], []
    for f in np.nonzero(~outside)[0]:
        xs = np.arange(b0[f, 0], b1[f, 0] + 1); ys = np.arange(b0[f, 1], b1[f, 1] + 1)
        c = (xs[:, None] * num_bins + ys[None, :]).reshape(-1)
        cells.append(c); tris.append(np.full(c.shape, f, np.int32))
    if cells:
        cells = np.concatenate(cells); tris = np.concatenate(tris)
        order = np.argsort(cells, kind="stable")
        cells, tris = cells[order], tris[order]
    else:
        cells = np.zeros(0, np.int64); tris = np.zeros(0, np.int32)
    start = np.zeros(num_bins * num_bins + 1, np.int32)
    np.add.at(start, cells + 1, 1)
    return np.cumsum(start).astype(np.int32), tris.astype(np.int32), size


def voxelize(verts, faces, num_voxels: int, extent: float = 0.0, min_point=(-1, -1, -1), max_point=(1, 1, 1), num_samples: int = 4,
             ior_inside: float = 1.33, ior_outside: float = 1.0, device=None, num_bins: Optional[int] = None, robust: bool = False):
    """voxelize_mesh.py:54-106. -> (data float32 [G,G,G] on `device`, ndim, nmin, nmax).
    robust=True: three-axis majority containment (parity rays along +z, +x and +y, at least two must agree) for meshes that are not
    watertight — a single parity ray through a hole misclassifies the sample (pysdf's one ray in a random frame has the same weakness);
    identical to the default on closed meshes.  Costs three passes and (G * num_samples)^3 bytes x 3 of scratch."""
    if robust:
        return _voxelize_robust(verts, faces, num_voxels, extent, min_point, max_point, num_samples, ior_inside, ior_outside, device, num_bins)
    lib = _lib.load()
    device = torch.device(device if device is not None else ("cuda", torch.cuda.current_device()))
    if extent > 0:                                                   # voxelize_mesh.py:88-93
        nmin, nmax = [-float(extent)] * 3, [float(extent)] * 3
    else:
        nmin, nmax = [float(v) for v in min_point], [float(v) for v in max_point]
    G = int(num_voxels)
    spec = Grid.make([G] * 3, nmin, nmax)
    verts = np.ascontiguousarray(verts, np.float64); faces = np.ascontiguousarray(faces, np.int32)
    pitch = (np.asarray(nmax) - np.asarray(nmin)) / (G - 1.0)
    lo = np.asarray(nmin) - pitch; hi = np.asarray(nmax) + pitch       # the sub-samples reach one pitch beyond the grid
    nb = int(num_bins or max(8, min(512, int(np.sqrt(max(len(faces), 1)) * 2))))
    start, tris, size = build_xy_bins(verts, faces, lo, hi, nb)
    if len(tris) == 0:
        tris = np.zeros(1, np.int32)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    v_d, f_d, s_d, t_d = d(verts), d(faces), d(start), d(tris)
    count = torch.empty(G * G * G, dtype=torch.int32, device=device)
    out = torch.empty((G, G, G), dtype=torch.float32, device=device)
    overflow = torch.zeros(1, dtype=torch.int32, device=device)
    org = (C.c_double * 4)(float(lo[0]), float(lo[1]), float(size[0]), float(size[1]))
    check(lib.rnerf_voxelize(ptr(v_d), ptr(f_d), ptr(s_d), ptr(t_d), nb, C.cast(org, C.c_void_p), C.byref(spec), int(num_samples),
                             float(ior_inside), float(ior_outside), ptr(count), ptr(out), ptr(overflow), current_stream()), "rnerf_voxelize")
    if int(overflow.item()) != 0:
        raise _lib.RnerfError("rnerf_voxelize: a sample column crosses the surface more than 96 times")
    return out, [G] * 3, nmin, nmax


def _voxelize_robust(verts, faces, num_voxels, extent, min_point, max_point, num_samples, ior_inside, ior_outside, device, num_bins):
    lib = _lib.load()
    device = torch.device(device if device is not None else ("cuda", torch.cuda.current_device()))
    if extent > 0:
        nmin, nmax = [-float(extent)] * 3, [float(extent)] * 3
    else:
        nmin, nmax = [float(v) for v in min_point], [float(v) for v in max_point]
    G, K = int(num_voxels), int(num_samples)
    GK = G * K
    if GK ** 3 * 3 > 48 * 2 ** 30:
        raise _lib.RnerfError(f"voxelize(robust=True): 3 x {GK}^3 sample flags do not fit the scratch budget (48 GiB)")
    verts = np.ascontiguousarray(verts, np.float64); faces = np.ascontiguousarray(faces, np.int32)
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(device)
    f_d = d(faces)
    overflow = torch.zeros(1, dtype=torch.int32, device=device)
    flags = []
    for perm in ((0, 1, 2), (1, 2, 0), (2, 0, 1)):                   # rays along z, x, y of the original frame
        vp = np.ascontiguousarray(verts[:, perm])
        lo_g = [nmin[a] for a in perm]; hi_g = [nmax[a] for a in perm]
        spec = Grid.make([G] * 3, lo_g, hi_g)
        pitch = (np.asarray(hi_g) - np.asarray(lo_g)) / (G - 1.0)
        lo = np.asarray(lo_g) - pitch; hi = np.asarray(hi_g) + pitch
        nb = int(num_bins or max(8, min(512, int(np.sqrt(max(len(faces), 1)) * 2))))
        start, tris, size = build_xy_bins(vp, faces, lo, hi, nb)
        if len(tris) == 0:
            tris = np.zeros(1, np.int32)
        inside = torch.empty(GK ** 3, dtype=torch.uint8, device=device)
        org = (C.c_double * 4)(float(lo[0]), float(lo[1]), float(size[0]), float(size[1]))
        v_d, s_d, t_d = d(vp), d(start), d(tris)                    # (named: a temporary would be freed — and its block reused — before the launch)
        check(lib.rnerf_voxelize_samples(ptr(v_d), ptr(f_d), ptr(s_d), ptr(t_d), nb, C.cast(org, C.c_void_p), C.byref(spec), K, ptr(inside),
                                         ptr(overflow), current_stream()), "rnerf_voxelize_samples")
        if int(overflow.item()) != 0:
            raise _lib.RnerfError("rnerf_voxelize_samples: a sample column crosses the surface more than 96 times")
        flags.append(inside)
    count = torch.empty(G * G * G, dtype=torch.int32, device=device)
    out = torch.empty((G, G, G), dtype=torch.float32, device=device)
    check(lib.rnerf_voxelize_majority(ptr(flags[0]), ptr(flags[1]), ptr(flags[2]), G, K, float(ior_inside), float(ior_outside), ptr(count), ptr(out),
                                      current_stream()), "rnerf_voxelize_majority")
    return out, [G] * 3, nmin, nmax


def save_mesh_pkl(path: str, data, extent: float = 0.0, min_point=(-1, -1, -1), max_point=(1, 1, 1), num_voxels: Optional[int] = None) -> None:
    """The dict voxelize_mesh.py:109-116 pickles (data float64 [G^3, 1], x slowest)."""
    a = data.detach().cpu().numpy() if isinstance(data, torch.Tensor) else np.asarray(data)
    G = int(num_voxels or a.shape[0])
    with open(path, "wb") as f:
        pickle.dump({"data": a.astype(np.float64).reshape(-1, 1), "extent": extent, "min_point": list(min_point), "max_point": list(max_point),
                     "num_voxels": G}, f)
