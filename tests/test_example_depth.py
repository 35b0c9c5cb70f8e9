"""Geometry of the example scene against ground truth the reference ships: Blender's depth pass of its one training view
(example_data/imgs/r_0_depth_0001.exr, every 8th pixel in tests/golden/example_depth.npz) versus the first place where the rays of the
example camera (example_data/transforms_train.json through Dataset._generate_rays' model) enter the 128^3 voxelisation of
example_data/voxelize/mesh_4_128_1.5_1.165.obj.  One number checks, end to end and against data this repository did not make: the camera
model (pixel centres, focal from camera_angle_x, the camera-to-world convention), the grid's placement ([-1.5, 1.5]^3), its axis order
(x slowest), the OBJ reader and the voxeliser.  Blender's Z pass is the distance ALONG THE VIEW AXIS (the camera's -z column)."""
import math
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import cases                                       # noqa: E402
from oracle import ref_np as R                     # noqa: E402

PITCH = 3.0 / 127.0                                # voxel pitch of the 128^3 grid over [-1.5, 1.5]
T_SAMPLES = np.linspace(2.0, 6.0, 4001)            # 1 mm steps between the example config's near and far


def _depth():
    z = np.load(os.path.join(HERE, "golden", "example_depth.npz"))
    return z["z"], z["rows"], z["cols"]


def _surface_points_and_normals(origins, viewdirs):
    """World positions of Blender's first surface along the sampled rays (from its depth pass) and its world-space normals there."""
    f = np.load(os.path.join(HERE, "golden", "example_depth.npz"))
    z, n = f["z"].reshape(-1).astype(np.float64), f["normal"].reshape(-1, 3).astype(np.float64)
    axis = -cases.EXAMPLE_C2W[:3, 2].astype(np.float64)
    hit = z < 1e9
    t = z[hit] / (viewdirs[hit].astype(np.float64) @ axis)
    return origins[hit].astype(np.float64) + t[:, None] * viewdirs[hit].astype(np.float64), n[hit]


def _check_normals(grad, normals):
    """grad [n, 3]: the table's (d n / d x, y, z) at the surface points.  The index of refraction rises INTO the object: grad || -normal."""
    g = grad.astype(np.float64)
    cos = -(g / np.linalg.norm(g, axis=-1, keepdims=True) * normals / np.linalg.norm(normals, axis=-1, keepdims=True)).sum(-1)
    print(f"{len(cos)} surface points: cos(grad n, -normal) median {np.median(cos):.4f}, mean {cos.mean():.4f}, 10th percentile {np.percentile(cos, 10):.4f}")
    assert np.median(cos) > 0.999 and cos.mean() > 0.99 and np.percentile(cos, 10) > 0.97
    assert np.linalg.norm(g, axis=-1).min() > 1.0                         # every one of them sits in the boundary shell (|grad n| ~ 6.5 there)


def _check(first_hit_t, viewdirs, z):
    """first_hit_t [n]: ray parameter of the first sample whose inside fraction reaches 1/2 (inf: none); viewdirs [n, 3] unit; z [n] Blender's."""
    axis = -cases.EXAMPLE_C2W[:3, 2].astype(np.float64)                        # the camera looks along -z of its own frame
    planar = first_hit_t * (viewdirs.astype(np.float64) @ axis)
    exr_hit, vox_hit = z < 1e9, np.isfinite(first_hit_t)
    assert (exr_hit == vox_hit).mean() > 0.985                                 # silhouettes agree except on edge pixels (measured 0.993)
    both = exr_hit & vox_hit
    err = planar[both] - z[both]
    print(f"{both.sum()} pixels on the object: depth error median {np.median(err):+.4f}, mean |e| {np.abs(err).mean():.4f}, p95 {np.percentile(np.abs(err), 95):.4f}, "
          f"max {np.abs(err).max():.4f} (voxel pitch {PITCH:.4f})")
    assert abs(np.median(err)) < 0.25 * PITCH and np.percentile(np.abs(err), 95) < 0.6 * PITCH and np.abs(err).max() < 3 * PITCH
    euclid = first_hit_t[both] - z[both]                                       # the wrong reading of the pass is visibly wrong: 2.5 pitches off
    assert np.median(euclid) > 2 * PITCH


def test_oracle_rays_enter_the_voxelised_object_where_blenders_depth_pass_says():
    z, rows, cols = _depth()
    H = W = 800
    focal = 0.5 * W / math.tan(0.5 * cases.EXAMPLE_CAMERA_ANGLE_X)             # datasets.py:361
    o, _, v = R.generate_rays(cases.EXAMPLE_C2W, H, W, focal=focal)
    O = o[rows][:, cols].reshape(-1, 3).astype(np.float64); V = v[rows][:, cols].reshape(-1, 3).astype(np.float64)
    _, _, counts = cases.load_example_obj()
    frac = counts.reshape(128, 128, 128).astype(np.float64) / 64.0             # inside fraction of the 4^3 sub-samples per voxel
    first = np.full(len(O), np.inf)
    for i in range(0, len(O), 500):
        P = O[i:i + 500, None, :] + T_SAMPLES[None, :, None] * V[i:i + 500, None, :]
        idx = np.rint((P + 1.5) / PITCH).astype(int)
        inside = np.all((idx >= 0) & (idx < 128), -1)
        idx = np.clip(idx, 0, 127)
        occ = (frac[idx[..., 0], idx[..., 1], idx[..., 2]] * inside) >= 0.5
        first[i:i + 500] = np.where(occ.any(1), T_SAMPLES[occ.argmax(1)], np.inf)
    _check(first, V, z.reshape(-1))


@pytest.mark.gpu
def test_device_rays_voxeliser_and_lookup_against_blenders_depth_pass():
    """The same through the HIP entry points only: rnerf_generate_rays, rnerf_voxelize on the OBJ, rnerf_grid_build_table + rnerf_grid_query
    (trilinear inside fraction along the rays)."""
    torch = pytest.importorskip("torch")
    from samplenerfro_amd import _lib, ops, voxelize as VX
    dev = torch.device("cuda:0")
    z, rows, cols = _depth()
    H = W = 800
    focal = 0.5 * W / math.tan(0.5 * cases.EXAMPLE_CAMERA_ANGLE_X)
    o, _, v = ops.generate_rays(cases.EXAMPLE_C2W, H, W, dev, focal=focal)
    r = torch.from_numpy(rows).to(dev); c = torch.from_numpy(cols).to(dev)
    O = o[r][:, c].reshape(-1, 3); V = v[r][:, c].reshape(-1, 3)
    verts, faces, _ = cases.load_example_obj()
    data, ndim, nmin, nmax = VX.voxelize(cases.example_obj_world(verts), faces, 128, extent=1.5, num_samples=4, device=dev, ior_inside=2.0, ior_outside=1.0)
    spec = _lib.Grid.make(ndim, nmin, nmax)
    table = ops.grid_build_table(data, spec)                                   # channel 0 = 1 + inside fraction
    t = torch.from_numpy(T_SAMPLES.astype(np.float32)).to(dev)
    first = torch.full((O.shape[0],), float("inf"), device=dev)
    for i in range(0, O.shape[0], 1000):
        P = (O[i:i + 1000, None, :] + t[None, :, None] * V[i:i + 1000, None, :]).reshape(-1, 3).contiguous()
        f = ops.grid_query(table, spec, P)[:, 0].reshape(-1, t.numel()) - 1.0
        inside = ((P >= -1.5) & (P <= 1.5)).all(-1).reshape(-1, t.numel())
        occ = (f >= 0.5) & inside
        first[i:i + 1000] = torch.where(occ.any(1), t[occ.float().argmax(1)], torch.full_like(first[i:i + 1000], float("inf")))
    _check(first.cpu().numpy().astype(np.float64), V.cpu().numpy(), z.reshape(-1))


def test_oracle_gradient_table_points_along_blenders_normals():
    """Rows G1 + G2 + G3 on the example scene: prefilter (3, 1.0), gradient table, trilinear lookup — evaluated at the surface points
    Blender's depth pass gives, against Blender's normal pass (example_data/imgs/r_0_normal_0001.exr): axis order and sign of the gradient."""
    _, rows, cols = _depth()
    H = W = 800
    focal = 0.5 * W / math.tan(0.5 * cases.EXAMPLE_CAMERA_ANGLE_X)
    o, _, v = R.generate_rays(cases.EXAMPLE_C2W, H, W, focal=focal)
    P, N = _surface_points_and_normals(o[rows][:, cols].reshape(-1, 3), v[rows][:, cols].reshape(-1, 3))
    _, _, counts = cases.load_example_obj()
    table = R.build_table(cases.example_grid(counts).astype(np.float32), [128] * 3, [-1.5] * 3, [1.5] * 3)
    _check_normals(R.linear3(table, P.astype(np.float32), [128] * 3, [-1.5] * 3, [1.5] * 3)[:, 1:4], N)


@pytest.mark.gpu
def test_device_gradient_table_points_along_blenders_normals():
    """The same through rnerf_voxelize -> (ri scaling) -> rnerf_grid_prefilter -> rnerf_grid_build_table -> rnerf_grid_query."""
    torch = pytest.importorskip("torch")
    from samplenerfro_amd import _lib, ops, voxelize as VX
    dev = torch.device("cuda:0")
    _, rows, cols = _depth()
    H = W = 800
    focal = 0.5 * W / math.tan(0.5 * cases.EXAMPLE_CAMERA_ANGLE_X)
    o, _, v = ops.generate_rays(cases.EXAMPLE_C2W, H, W, dev, focal=focal)
    o, v = o.cpu().numpy(), v.cpu().numpy()
    P, N = _surface_points_and_normals(o[rows][:, cols].reshape(-1, 3), v[rows][:, cols].reshape(-1, 3))
    verts, faces, _ = cases.load_example_obj()
    data, ndim, nmin, nmax = VX.voxelize(cases.example_obj_world(verts), faces, 128, extent=1.5, num_samples=4, device=dev)
    scaled = ((data.double() - 1.0) * 0.5 / 0.33 + 1.0).float()          # train.py:222, ri = 0.5 (configs/example.yaml)
    spec = _lib.Grid.make(ndim, nmin, nmax)
    table = ops.grid_build_table(ops.grid_prefilter(scaled, 3, 1.0), spec)
    got = ops.grid_query(table, spec, torch.from_numpy(P.astype(np.float32)).to(dev)).cpu().numpy()
    _check_normals(got[:, 1:4], N)
