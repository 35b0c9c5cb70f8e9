"""Thin torch-tensor wrappers over the C ABI (one function per entry point of include/rnerf.h).

PyTorch is plumbing here: it owns the device memory and the stream; all arithmetic is in librnerf.so.
Layouts: "sample-major" arrays are [S, B, ...] tensors (sample/node index first, ray index second).
"""
from __future__ import annotations

import ctypes as C
import os
import weakref
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import Grid, check, current_stream, ptr


def _chk(t: torch.Tensor, name: str, dtype=torch.float32) -> torch.Tensor:
    if not t.is_cuda:
        raise _lib.RnerfError(f"{name}: expected a CUDA (ROCm) tensor; the hot path has no CPU implementation")
    if t.dtype != dtype:
        raise _lib.RnerfError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def grid_prefilter(grid: torch.Tensor, ksize: int, ksigma: float) -> torch.Tensor:
    """G1: ior_utils.conv3d_normal (rnerf/ior_utils.py:327-363). grid [Gx,Gy,Gz] f32 -> same shape."""
    lib = _lib.load()
    g = _chk(grid, "grid")
    dst = torch.empty_like(g); tmp = torch.empty_like(g)
    dims = (C.c_int32 * 3)(*g.shape)
    check(lib.rnerf_grid_prefilter(ptr(g), ptr(dst), ptr(tmp), C.byref(dims), int(ksize), float(ksigma), current_stream()),
          "rnerf_grid_prefilter")
    return dst


def grid_build_table(grid: torch.Tensor, spec: Grid) -> torch.Tensor:
    """G2: VoxMLP.setup/_compute_grad (rnerf/ior_utils.py:139-172). -> table [G^3, 4] = (n, grad n)."""
    lib = _lib.load()
    g = _chk(grid, "grid")
    n = spec.dims[0] * spec.dims[1] * spec.dims[2]
    if g.numel() != n:
        raise _lib.RnerfError(f"grid has {g.numel()} voxels, spec says {n}")
    nf = lib.rnerf_grid_table_floats(C.byref(spec))
    if nf == 0:
        check(-1, "rnerf_grid_table_floats")
    table = torch.empty((nf // 4, 4), dtype=torch.float32, device=g.device)      # in spec.layout (bricks pad odd dimensions)
    check(lib.rnerf_grid_build_table(ptr(g), ptr(table), C.byref(spec), current_stream()), "rnerf_grid_build_table")
    return table


def table_reference_order(table: torch.Tensor, spec: Grid) -> torch.Tensor:
    """The table as the reference indexes it: [Gx*Gy*Gz, 4], flat index x*Gy*Gz + y*Gz + z (rnerf/ior_utils.py:161,214) — the table itself
    for the reference layout, a gather out of the 2x2x2 bricks otherwise (rnerf_table_layout in include/rnerf.h).  For the oracle / tests."""
    if spec.layout == _lib.TABLE_LAYOUTS["reference"]:
        return table
    dx, dy, dz = (int(v) for v in spec.dims)
    by, bz = (dy + 1) // 2, (dz + 1) // 2
    dev = table.device
    x = torch.arange(dx, device=dev).view(dx, 1, 1); y = torch.arange(dy, device=dev).view(1, dy, 1); z = torch.arange(dz, device=dev).view(1, 1, dz)
    idx = (((x >> 1) * by + (y >> 1)) * bz + (z >> 1)) * 8 + (x & 1) * 4 + (y & 1) * 2 + (z & 1)
    return table.reshape(-1, 4)[idx.reshape(-1)]


def grid_query(table: torch.Tensor, spec: Grid, pts: torch.Tensor, want_idx: bool = False):
    """G3: VoxMLP._linear3 (rnerf/ior_utils.py:188-223). pts [n,3] -> [n,4] (+ int32 [n,6])."""
    lib = _lib.load()
    table = _chk(table, "table"); pts = _chk(pts, "pts")
    n = pts.shape[0]
    out = torch.empty((n, 4), dtype=torch.float32, device=pts.device)
    idx = torch.empty((n, 6), dtype=torch.int32, device=pts.device) if want_idx else None
    check(lib.rnerf_grid_query(ptr(table), C.byref(spec), ptr(pts), n, ptr(out), ptr(idx), current_stream()), "rnerf_grid_query")
    return (out, idx) if want_idx else out


def march(table: torch.Tensor, spec: Grid, origins: torch.Tensor, viewdirs: torch.Tensor, near: float, far: float,
          num_nodes: int, want_ior: bool = False, want_vox: bool = False, out=None):
    """E1/E2: PathSampler.__call__ (rnerf/eikonal_utils.py:100-124). -> path_pd [N,B,4], path_dr [N,B,4], ior?, vox?"""
    lib = _lib.load()
    table = _chk(table, "table"); o = _chk(origins, "origins"); d = _chk(viewdirs, "viewdirs")
    B = o.shape[0]
    if out is not None:
        path_pd, path_dr = out
    else:
        path_pd = torch.empty((num_nodes, B, 4), dtype=torch.float32, device=o.device)
        path_dr = torch.empty((num_nodes, B, 4), dtype=torch.float32, device=o.device)
    ior = torch.empty((num_nodes, B, 4), dtype=torch.float32, device=o.device) if want_ior else None
    vox = torch.empty((num_nodes, B, 6), dtype=torch.int32, device=o.device) if want_vox else None
    check(lib.rnerf_march(ptr(table), C.byref(spec), ptr(o), ptr(d), B, float(near), float(far), int(num_nodes),
                          ptr(path_pd), ptr(path_dr), ptr(ior), ptr(vox), current_stream()), "rnerf_march")
    return path_pd, path_dr, ior, vox


def nerfmlp_pack(params_flat: torch.Tensor, precision: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Pack a flat fp32 NerfMLP parameter buffer (595844 floats, flax order) into the MFMA operand stream."""
    lib = _lib.load()
    p = _chk(params_flat, "params_flat")
    if p.numel() != _lib.NERFMLP_PARAMS:
        raise _lib.RnerfError(f"NerfMLP flat params must have {_lib.NERFMLP_PARAMS} floats, got {p.numel()}")
    nbytes = lib.rnerf_nerfmlp_packed_bytes(int(precision))
    if nbytes == 0:
        check(-1, "rnerf_nerfmlp_packed_bytes")
    if out is None:
        out = torch.empty(nbytes, dtype=torch.uint8, device=p.device)
    check(lib.rnerf_nerfmlp_pack(ptr(p), int(precision), ptr(out), current_stream()), "rnerf_nerfmlp_pack")
    return out


def nerfmlp_forward(packed: torch.Tensor, precision: int, rows_pd: torch.Tensor, rows_dr: torch.Tensor,
                    node_of_sample: Optional[torch.Tensor], S: int, B: int, out: Optional[torch.Tensor] = None,
                    max_workgroups: int = 0) -> torch.Tensor:
    """P1+N1: pos_enc + NerfMLP (rnerf/model_utils.py:187-214, :30-90). -> raw [S,B,4] = (rgb raw, sigma raw).
    max_workgroups: cap of the persistent grid (0 = every CU), see include/rnerf.h."""
    lib = _lib.load()
    rows_pd = _chk(rows_pd, "rows_pd"); rows_dr = _chk(rows_dr, "rows_dr")
    if node_of_sample is not None:
        node_of_sample = _chk(node_of_sample, "node_of_sample", torch.int32)
    if out is None:
        out = torch.empty((S, B, 4), dtype=torch.float32, device=rows_pd.device)
    check(lib.rnerf_nerfmlp_forward(ptr(packed), int(precision), ptr(rows_pd), ptr(rows_dr), ptr(node_of_sample), int(S), int(B),
                                    ptr(out), int(max_workgroups), current_stream()), "rnerf_nerfmlp_forward")
    return out


def bkgd_forward(params_flat: torch.Tensor, dirs: torch.Tensor, rgb_padding: float = 0.001) -> torch.Tensor:
    """P1+N2: bkgd_mlp(pos_enc(dir,0,4)) + rgb activation (rnerf/models.py:181-191,303,336-337). dirs [n,3|4] -> [n,3]."""
    lib = _lib.load()
    p = _chk(params_flat, "params_flat"); d = _chk(dirs, "dirs")
    if p.numel() != _lib.BKGDMLP_PARAMS:
        raise _lib.RnerfError(f"bkgd MLP flat params must have {_lib.BKGDMLP_PARAMS} floats, got {p.numel()}")
    n = d.shape[0]
    out = torch.empty((n, 3), dtype=torch.float32, device=d.device)
    check(lib.rnerf_bkgd_forward(ptr(p), ptr(d), int(d.shape[-1]), n, float(rgb_padding), ptr(out), current_stream()),
          "rnerf_bkgd_forward")
    return out


def composite(raw: torch.Tensor, rows_pd: torch.Tensor, rows_dr: torch.Tensor, node_of_sample: Optional[torch.Tensor],
              S: int, B: int, bkgd: Optional[torch.Tensor], white_bkgd: bool = False, rgb_padding: float = 0.001,
              sigma_bias: float = -1.0, want_weights: bool = True, want_alpha: bool = False, mask_mode: int = 0, bbox=None):
    """V1: activations + volumetric_rendering (rnerf/models.py:334-349, rnerf/model_utils.py:247-309)."""
    lib = _lib.load()
    raw = _chk(raw, "raw"); dev = raw.device
    rgb = torch.empty((B, 3), dtype=torch.float32, device=dev)
    dist = torch.empty((B,), dtype=torch.float32, device=dev)
    acc = torch.empty((B,), dtype=torch.float32, device=dev)
    trans = torch.empty((B, 1), dtype=torch.float32, device=dev)
    trans_bkgd = torch.empty((B, 3), dtype=torch.float32, device=dev)
    weights = torch.empty((S, B), dtype=torch.float32, device=dev) if want_weights else None
    alpha = torch.empty((S, B), dtype=torch.float32, device=dev) if want_alpha else None
    if bkgd is not None:
        bkgd = _chk(bkgd, "bkgd")
    check(lib.rnerf_composite(ptr(raw), ptr(rows_pd), ptr(rows_dr), ptr(node_of_sample), int(S), int(B), ptr(bkgd),
                              int(bool(white_bkgd)), float(rgb_padding), float(sigma_bias), ptr(rgb), ptr(dist), ptr(acc),
                              ptr(trans), ptr(trans_bkgd), ptr(weights), ptr(alpha), int(mask_mode),
                              C.byref((C.c_double * 6)(*[float(v) for v in bbox])) if bbox is not None else None,
                              current_stream()), "rnerf_composite")
    return rgb, dist, acc, trans, trans_bkgd, weights, alpha


def resample(path_pd: torch.Tensor, path_dr: torch.Tensor, jitter: torch.Tensor, weights: torch.Tensor, u: torch.Tensor,
             num_fine: int, want_idx: bool = False):
    """S1+S2: sample_pdf along the bent path (rnerf/model_utils.py:312-435).

    u: [num_fine] (shared by all rays) or [num_fine, B]; non-decreasing along axis 0.
    -> rows_pd [S+F,B,4], rows_dr [S+F,B,4], node_idx int32 [S+F,B] (if want_idx).
    """
    lib = _lib.load()
    N, B = path_pd.shape[0], path_pd.shape[1]
    S = jitter.shape[0]
    jitter = _chk(jitter, "jitter", torch.int32); weights = _chk(weights, "weights"); u = _chk(u, "u")
    per_ray = 1 if u.dim() == 2 else 0
    if u.shape[0] != num_fine or (per_ray and u.shape[1] != B):
        raise _lib.RnerfError(f"u must be [F] or [F,B] with F={num_fine}, B={B}; got {tuple(u.shape)}")
    dev = path_pd.device
    T = S + num_fine
    rows_pd = torch.empty((T, B, 4), dtype=torch.float32, device=dev)
    rows_dr = torch.empty((T, B, 4), dtype=torch.float32, device=dev)
    idx = torch.empty((T, B), dtype=torch.int32, device=dev) if want_idx else None
    scratch = torch.empty((T, B), dtype=torch.float32, device=dev)
    check(lib.rnerf_resample(ptr(path_pd), ptr(path_dr), int(N), int(B), ptr(jitter), int(S), ptr(weights), ptr(u), per_ray,
                             int(num_fine), ptr(rows_pd), ptr(rows_dr), ptr(idx), ptr(scratch), current_stream()), "rnerf_resample")
    return rows_pd, rows_dr, idx


def loss_reduce(rgb_c: Optional[torch.Tensor], rgb_f: torch.Tensor, trans_f: torch.Tensor, trans_bkgd_f: torch.Tensor,
                pixels: torch.Tensor) -> torch.Tensor:
    """T1: the reductions of train_step.loss_fn (train.py:89-92,105). -> sums float[4] on the device."""
    lib = _lib.load()
    B = rgb_f.shape[0]
    sums = torch.empty(4, dtype=torch.float32, device=rgb_f.device)
    check(lib.rnerf_loss_reduce(ptr(rgb_c), ptr(_chk(rgb_f, "rgb_f")), ptr(_chk(trans_f, "trans_f")), ptr(_chk(trans_bkgd_f, "trans_bkgd_f")),
                                ptr(_chk(pixels, "pixels")), int(B), ptr(sums), current_stream()), "rnerf_loss_reduce")
    return sums


def composite_backward(raw, rows_pd, rows_dr, node_of_sample, S: int, B: int, bkgd, rgb, pixels, trans=None, trans_bkgd=None,
                       sums=None, mse_scale: float = 0.0, bg_scale: float = 0.0, d_bkgd: Optional[torch.Tensor] = None,
                       rgb_padding: float = 0.001, sigma_bias: float = -1.0, bd_cut_bbox=None, white_bkgd: bool = False,
                       accumulate_bkgd: Optional[bool] = None, mask_bbox=None):
    """T1: backward of activations + volumetric_rendering for one level. -> d_raw [S,B,4], d_bkgd [B,3] (accumulated if given).
    bd_cut_bbox: the level's trans pair is the bd_cut_dist pair (mask mode 1); mask_bbox: the level was rendered with use_mask_bbox (mode 3)."""
    if bd_cut_bbox is not None and mask_bbox is not None:
        raise ValueError("bd_cut_dist and use_mask_bbox exclude each other (rnerf/models.py:480)")
    mode, box = (1, bd_cut_bbox) if bd_cut_bbox is not None else ((3, mask_bbox) if mask_bbox is not None else (0, None))
    lib = _lib.load()
    dev = raw.device
    d_raw = torch.empty((S, B, 4), dtype=torch.float32, device=dev)
    acc = (d_bkgd is not None) if accumulate_bkgd is None else bool(accumulate_bkgd)
    if d_bkgd is None:
        d_bkgd = torch.empty((B, 3), dtype=torch.float32, device=dev)
    check(lib.rnerf_composite_backward(ptr(_chk(raw, "raw")), ptr(rows_pd), ptr(rows_dr), ptr(node_of_sample), int(S), int(B),
                                       ptr(_chk(bkgd, "bkgd")), float(rgb_padding), float(sigma_bias), ptr(_chk(rgb, "rgb")),
                                       ptr(_chk(pixels, "pixels")), ptr(trans), ptr(trans_bkgd), ptr(sums), float(mse_scale),
                                       float(bg_scale), ptr(d_raw), ptr(d_bkgd), int(acc), int(bool(white_bkgd)), int(mode),
                                       None if box is None else (C.c_double * 6)(*[float(v) for v in box]), current_stream()),
          "rnerf_composite_backward")
    return d_raw, d_bkgd


def nerfmlp_forward_train(packed, precision: int, rows_pd, rows_dr, node_of_sample, S: int, B: int, backward: int = _lib.BWD_F16X2,
                          max_workgroups: int = 0):
    """Training forward: raw [S,B,4] + the saved operands (uint8 buffer) for the backward kernels (`backward`: _lib.BWD_*)."""
    lib = _lib.load()
    dev = rows_pd.device
    out = torch.empty((S, B, 4), dtype=torch.float32, device=dev)
    save = torch.empty(lib.rnerf_nerfmlp_save_bytes(S * B, int(backward)), dtype=torch.uint8, device=dev)
    check(lib.rnerf_nerfmlp_forward_train(ptr(packed), int(precision), ptr(_chk(rows_pd, "rows_pd")), ptr(_chk(rows_dr, "rows_dr")),
                                          ptr(node_of_sample), int(S), int(B), ptr(out), ptr(save), int(backward), int(max_workgroups),
                                          current_stream()),
          "rnerf_nerfmlp_forward_train")
    return out, save


def nerfmlp_pack_bwd(params_flat: torch.Tensor, out: Optional[torch.Tensor] = None, backward: int = _lib.BWD_F16X2) -> torch.Tensor:
    lib = _lib.load()
    p = _chk(params_flat, "params_flat")
    if out is None:
        out = torch.empty(lib.rnerf_nerfmlp_bwd_packed_bytes(), dtype=torch.uint8, device=p.device)
    check(lib.rnerf_nerfmlp_pack_bwd(ptr(p), int(backward), ptr(out), current_stream()), "rnerf_nerfmlp_pack_bwd")
    return out


def nerfmlp_backward(packed_bwd, packed_fwd, precision: int, save, d_raw: torch.Tensor, rows: int,
                     grads: Optional[torch.Tensor] = None, workspace: Optional[torch.Tensor] = None, dy: Optional[torch.Tensor] = None,
                     stages: str = "dw", backward: int = _lib.BWD_F16X2, return_dy: bool = False, between=None) -> torch.Tensor:
    """d_raw [S,B,4] (d loss / d raw) -> flat fp32 gradient of the NerfMLP parameters (595844 floats).  `backward` (_lib.BWD_*) must be
    the mode the forward saved for and packed_bwd was packed for.  between: optional callable run after the dgrad launch and before the
    wgrad launch (train_step issues the next step's march there)."""
    lib = _lib.load()
    dev = d_raw.device
    dy = torch.empty(lib.rnerf_nerfmlp_dy_bytes(rows, int(backward)), dtype=torch.uint8, device=dev) if dy is None else dy
    if "d" in stages:
        check(lib.rnerf_nerfmlp_dgrad(ptr(packed_bwd), ptr(packed_fwd), int(precision), int(backward), ptr(save), ptr(_chk(d_raw, "d_raw")), int(rows),
                                      ptr(dy), current_stream()), "rnerf_nerfmlp_dgrad")
    if "w" not in stages:
        return dy
    if between is not None:
        between()
    if grads is None:
        grads = torch.empty(_lib.NERFMLP_PARAMS, dtype=torch.float32, device=dev)
    if workspace is None:
        workspace = torch.empty(lib.rnerf_nerfmlp_wgrad_workspace_bytes(), dtype=torch.uint8, device=dev)
    check(lib.rnerf_nerfmlp_wgrad(int(precision), int(backward), ptr(save), ptr(dy), int(rows), ptr(grads), ptr(workspace), current_stream()),
          "rnerf_nerfmlp_wgrad")
    return (grads, dy) if return_dy else grads


def bkgd_forward_train(params_flat: torch.Tensor, dirs: torch.Tensor, rgb_padding: float = 0.001):
    lib = _lib.load()
    p = _chk(params_flat, "params_flat"); d = _chk(dirs, "dirs")
    n = d.shape[0]
    out = torch.empty((n, 3), dtype=torch.float32, device=d.device)
    save = torch.empty(lib.rnerf_bkgd_save_bytes(n), dtype=torch.uint8, device=d.device)
    check(lib.rnerf_bkgd_forward_train(ptr(p), ptr(d), int(d.shape[-1]), n, float(rgb_padding), ptr(out), ptr(save), current_stream()),
          "rnerf_bkgd_forward_train")
    return out, save


def bkgd_backward(params_flat: torch.Tensor, save: torch.Tensor, d_out: torch.Tensor, grads: torch.Tensor, rgb_padding: float = 0.001,
                  want_d_dirs: bool = False):
    """Accumulates d loss / d params of the background MLP into `grads` (56963 floats); want_d_dirs: also returns d loss / d direction
    [n, 4] (stage "all*")."""
    lib = _lib.load()
    n = d_out.shape[0]
    dy = torch.empty(lib.rnerf_bkgd_dy_bytes(n), dtype=torch.uint8, device=d_out.device)
    d_dirs = torch.empty((n, 4), dtype=torch.float32, device=d_out.device) if want_d_dirs else None
    check(lib.rnerf_bkgd_backward(ptr(_chk(params_flat, "params_flat")), ptr(save), ptr(_chk(d_out, "d_out")), n, float(rgb_padding), ptr(dy),
                                  ptr(_chk(grads, "grads")), ptr(d_dirs), current_stream()), "rnerf_bkgd_backward")
    return (grads, d_dirs) if want_d_dirs else grads


def stratified_u(key, B: int, num_fine: int, device) -> torch.Tensor:
    """S1 randomized draws on the device (rnerf/model_utils.py:345-354) -> u [num_fine, B]."""
    import ctypes as C
    lib = _lib.load()
    k = (C.c_uint32 * 2)(int(key[0]), int(key[1]))
    u = torch.empty((num_fine, B), dtype=torch.float32, device=device)
    check(lib.rnerf_stratified_u(C.cast(k, C.c_void_p), int(B), int(num_fine), ptr(u), current_stream()), "rnerf_stratified_u")
    return u


def generate_rays(camtoworld, H: int, W: int, device, focal: Optional[float] = None, cam_mat=None, pixel_center: bool = True,
                  rows: Optional[Tuple[int, int]] = None, want_directions: bool = False):
    """Dataset._generate_rays on the device (rnerf/datasets.py:216-242 with `focal`, :486-518 with `cam_mat`).
    -> (origins, directions | None, viewdirs), each [rows, W, 3] for image rows [rows[0], rows[1])."""
    import numpy as np
    lib = _lib.load()
    c2w = np.ascontiguousarray(np.asarray(camtoworld, np.float32)[:3, :4])
    r0, r1 = rows if rows is not None else (0, H)
    n = r1 - r0
    o = torch.empty((n, W, 3), dtype=torch.float32, device=device)
    v = torch.empty_like(o)
    d = torch.empty_like(o) if want_directions else None
    pc = 0.5 if pixel_center else 0.0
    if cam_mat is None:
        args = (0, float(focal), float(focal), W * 0.5, H * 0.5)
    else:
        args = (1, float(cam_mat[0][0]), float(cam_mat[1][1]), float(cam_mat[0][2]), float(cam_mat[1][2]))
    check(lib.rnerf_generate_rays(c2w.ctypes.data_as(C.c_void_p), args[0], args[1], args[2], args[3], args[4], pc, int(W), int(r0), int(n),
                                  ptr(o), ptr(d), ptr(v), current_stream()), "rnerf_generate_rays")
    return o, d, v


def sample_batch(camtoworlds: torch.Tensor, images: Optional[torch.Tensor], ray_indices: torch.Tensor, H: int, W: int, *, focal: Optional[float] = None,
                 cam_mat=None, pixel_center: bool = True, bad_count: Optional[torch.Tensor] = None, want_directions: bool = False):
    """`images[...][ray_indices]` and `rays[...][ray_indices]` of Dataset._next_train (rnerf/datasets.py:151-176) on the device.
    camtoworlds [n, 3, 4] and images [n, H, W, C] (None: rays only) are device tensors; ray_indices int64 [B] are flat over (image, row,
    column).  -> (origins, directions | None, viewdirs, pixels | None); rays are generated for the drawn pixels only (rnerf_sample_batch)."""
    lib = _lib.load()
    c2w = _chk(camtoworlds, "camtoworlds")
    idx = ray_indices.contiguous()
    if idx.dtype != torch.int64 or not idx.is_cuda:
        raise ValueError("sample_batch: ray_indices must be an int64 device tensor")
    n, B, dev = int(c2w.shape[0]), int(idx.numel()), idx.device
    if tuple(c2w.shape[1:]) != (3, 4):
        raise ValueError("sample_batch: camtoworlds must be [n, 3, 4]")
    ch = 0
    if images is not None:
        images = _chk(images, "images")
        if tuple(images.shape[:3]) != (n, H, W):
            raise ValueError("sample_batch: images must be [n, H, W, C] for the n cameras")
        ch = int(images.shape[3])
    o = torch.empty((B, 3), dtype=torch.float32, device=dev)
    v = torch.empty_like(o)
    d = torch.empty_like(o) if want_directions else None
    pix = torch.empty((B, ch), dtype=torch.float32, device=dev) if images is not None else None
    if bad_count is None:
        bad_count = torch.zeros(1, dtype=torch.int32, device=dev)
    pc = 0.5 if pixel_center else 0.0
    if cam_mat is None:
        cam = (0, float(focal), float(focal), W * 0.5, H * 0.5)
    else:
        cam = (1, float(cam_mat[0][0]), float(cam_mat[1][1]), float(cam_mat[0][2]), float(cam_mat[1][2]))
    check(lib.rnerf_sample_batch(ptr(c2w), n, cam[0], cam[1], cam[2], cam[3], cam[4], pc, int(W), int(H), ptr(images), ch, ptr(idx), B, ptr(o), ptr(d),
                                 ptr(v), ptr(pix), ptr(bad_count), current_stream()), "rnerf_sample_batch")
    return o, d, v, pix


def ray_radii(directions: torch.Tensor) -> torch.Tensor:
    """Rays.radii of Dataset._generate_rays (rnerf/datasets.py:230-239) from the directions of a WHOLE image [H, W, 3] on the device
    (generate_rays(..., want_directions=True)): |d[r] - d[r + 1]| per pixel, the last row repeating the one before, times 2 / sqrt(12).
    Feeds integrated_pos_enc (the reference's commented mip path); plain tensor arithmetic — one pass over H x W x 3 floats."""
    diff = directions[:-1] - directions[1:]
    dx = torch.sqrt((diff * diff).sum(-1))
    dx = torch.cat([dx, dx[-2:-1]], 0)
    return dx[..., None] * (2.0 / 12.0 ** 0.5)


def so3_window(annealed_alpha: float, max_deg_point: int = 10):
    """cosine_easing_window(0, max_deg-1, max_deg, annealed_alpha * max_deg) in fp32 (rnerf/model_utils.py:218-233, ior_utils.py:283)."""
    import numpy as np
    f = np.float32
    bands = np.linspace(0, max_deg_point - 1, max_deg_point).astype(f)
    x = np.clip(f(annealed_alpha) * f(max_deg_point) - bands, f(0), f(1))
    return np.ascontiguousarray((f(0.5) * (f(1) + np.cos(f(np.pi) * x + f(np.pi)))).astype(f))


def so3_query(table: torch.Tensor, spec: Grid, so3_flat: torch.Tensor, pts: torch.Tensor, annealed_alpha: float = 1.0,
              condition: Optional[torch.Tensor] = None):
    """G4 + P2: VoxMLP.__call__ (rnerf/ior_utils.py:269-312). pts [n,3] -> (out [n,4] = (n, grad n), pred_grad [n,3]).
    condition [n,3]: rotate this vector instead of the looked-up gradient (wrapper_grad_mlp, :225-267)."""
    lib = _lib.load()
    p = _chk(pts, "pts")
    n = p.shape[0]
    out = torch.empty((n, 4), dtype=torch.float32, device=p.device)
    pred = torch.empty((n, 3), dtype=torch.float32, device=p.device)
    w = so3_window(annealed_alpha)
    check(lib.rnerf_so3_query(ptr(table), C.byref(spec), ptr(_chk(so3_flat, "so3_flat")), w.ctypes.data_as(C.c_void_p), ptr(p),
                              ptr(_chk(condition, "condition")) if condition is not None else None, n, ptr(out), ptr(pred), current_stream()),
          "rnerf_so3_query")
    return out, pred


_SHELL_CACHE: dict = {}
# Default of the `coherent` argument of march_all / march_all_train: hand the kernel a ray order in which the 16 rays of a workgroup meet
# the boundary shell over similar node ranges (a group evaluates so3_mlp whenever ANY of its rays is in the shell).  The order comes from a
# COARSE plain pre-march (num_nodes / 8 nodes of 8 x the step: ~0.1 ms at 4096 rays) + one sort; the kernel writes every record at the
# ray's own index, so nothing is permuted back.  ops.SHELL_ORDER = False marches the rays in the given order (a module attribute, set by tools /
# tests; the product reads no environment variable for it).
SHELL_ORDER = True
# the pre-march takes num_nodes / SHELL_ORDER_COARSEN nodes of SHELL_ORDER_COARSEN x the step (8: ~0.1 ms at 4096 rays; finer = a better order
# for more pre-march time — which train_step hides beside the NerfMLP wgrad when it knows the next batch)
SHELL_ORDER_COARSEN = 8


def _shell_order(table: torch.Tensor, spec: Grid, o: torch.Tensor, v: torch.Tensor, near: float, far: float, num_nodes: int):
    """int32 permutation that groups rays whose paths meet the boundary shell (|grad n| > 1e-3, where so3_mlp is evaluated) over the
    same node range.  The shell interval of a ray is read off a coarse pre-march without so3 (the so3 rotation and the step length are
    irrelevant for grouping); rays are results-independent, so the order changes no value."""
    # cached per ray batch (tensor objects + versions): evaluating repeatedly on the same rays pays the pre-march and the sort once
    key = (id(o), o._version, id(v), v._version, o.shape[0], int(num_nodes), table.data_ptr(), float(near), float(far))
    ent = _SHELL_CACHE.get(key)
    if ent is not None and ent[0]() is o and ent[1]() is v:
        if ent[3] is not None:      # computed ahead on another stream (prefetch_shell_order): this stream waits for it, once
            torch.cuda.current_stream().wait_event(ent[3])
            ent[2].record_stream(torch.cuda.current_stream())
            _SHELL_CACHE[key] = (ent[0], ent[1], ent[2], None)
        return ent[2]
    nc = max(int(num_nodes) // int(SHELL_ORDER_COARSEN), 16)
    _, _, ior, _ = march(table, spec, o, v, near, far, nc, want_ior=True)
    g = ior[..., 1:4]
    m = (g * g).sum(-1) > 1e-6                                       # [nc, B]
    hit = m.any(0)
    first = torch.argmax(m.to(torch.uint8), 0)
    last = nc - 1 - torch.argmax(torch.flip(m, [0]).to(torch.uint8), 0)
    skey = torch.where(hit, (first // 2) * nc + last, torch.full_like(first, 2 * nc * nc))
    perm = torch.argsort(skey, stable=True).to(torch.int32)
    if len(_SHELL_CACHE) >= 8:
        _SHELL_CACHE.clear()
    _SHELL_CACHE[key] = (weakref.ref(o), weakref.ref(v), perm, None)
    return perm


def prefetch_shell_order(table: torch.Tensor, spec: Grid, origins: torch.Tensor, viewdirs: torch.Tensor, near: float, far: float, num_nodes: int,
                         stream: "torch.cuda.Stream") -> None:
    """The shell-coherent ray order of a batch the stage-all* march will meet LATER (the next training batch), computed now on `stream`: the
    coarse pre-march (a 50-register kernel: it fits beside the NerfMLP kernels' tails and the weight-gradient kernel) and the sort depend on
    the rays and the grid only, not on so3_mlp.  march_all / march_all_train pick the result up from the cache and wait for its event."""
    if not (SHELL_ORDER and origins.shape[0] > 16):
        return
    o = _chk(origins, "origins"); v = _chk(viewdirs, "viewdirs")
    key = (id(o), o._version, id(v), v._version, o.shape[0], int(num_nodes), table.data_ptr(), float(near), float(far))
    if key in _SHELL_CACHE:
        return
    stream.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(stream):
        o.record_stream(stream); v.record_stream(stream)
        _shell_order(table, spec, o, v, near, far, num_nodes)
        ev = torch.cuda.Event()
        ev.record(stream)
    ent = _SHELL_CACHE[key]
    _SHELL_CACHE[key] = (ent[0], ent[1], ent[2], ev)


def march_all(table: torch.Tensor, spec: Grid, so3_flat: torch.Tensor, origins: torch.Tensor, viewdirs: torch.Tensor, near: float, far: float,
              num_nodes: int, annealed_alpha: float = 1.0, want_ior: bool = False, coherent: Optional[bool] = None):
    """E1/E2 with stage "all*" (rnerf/eikonal_utils.py:34-39). -> path_pd [N,B,4], path_dr [N,B,4], ior?"""
    lib = _lib.load()
    o = _chk(origins, "origins"); v = _chk(viewdirs, "viewdirs")
    B = o.shape[0]
    if coherent is None:
        coherent = SHELL_ORDER
    order = _shell_order(table, spec, o, v, near, far, num_nodes) if (coherent and B > 16) else None
    pd = torch.empty((num_nodes, B, 4), dtype=torch.float32, device=o.device)
    dr = torch.empty_like(pd)
    ior = torch.empty_like(pd) if want_ior else None
    w = so3_window(annealed_alpha)
    packed = torch.empty(lib.rnerf_so3_packed_bytes(), dtype=torch.uint8, device=o.device)
    check(lib.rnerf_march_all(ptr(table), C.byref(spec), ptr(_chk(so3_flat, "so3_flat")), ptr(packed), w.ctypes.data_as(C.c_void_p), ptr(o), ptr(v), B,
                              float(near), float(far), int(num_nodes), ptr(pd), ptr(dr), ptr(ior), ptr(order), current_stream()), "rnerf_march_all")
    return pd, dr, ior


# order of the compacted (ray, node) pair list of the stage-all* march: "sorted" (default) = by (node, ray), deterministic from run to run;
# "atomic" = the kernel's arrival order (ops.PAIR_ORDER = "atomic": saves a sort of n_pairs keys and a gather over [N, B] per step)
PAIR_ORDER = "sorted"


def march_all_train(table: torch.Tensor, spec: Grid, so3_flat: torch.Tensor, origins: torch.Tensor, viewdirs: torch.Tensor, near: float, far: float,
                    num_nodes: int, annealed_alpha: float = 1.0, pair_cap: Optional[int] = None, coherent: Optional[bool] = None, lazy: bool = False):
    """rnerf_march_all_train: the stage "all*" march + the record its backward needs.  Returns a dict; the pairs are trimmed to their count
    and ordered by finalize_pairs(rec), which reads the device counter (one host synchronisation per step).  lazy=True leaves that call
    to the consumer (train._all_stage_backward): the forward kernels of the step, which need the path record only, are then queued
    BEFORE the host waits for the march — the synchronisation costs no idle device time (round 5: 0.35 ms of every stage-all* step)."""
    lib = _lib.load()
    o = _chk(origins, "origins"); v = _chk(viewdirs, "viewdirs")
    B, N, dev = o.shape[0], int(num_nodes), o.device
    if coherent is None:
        coherent = SHELL_ORDER
    order = _shell_order(table, spec, o, v, near, far, N) if (coherent and B > 16) else None
    cap = int(pair_cap) if pair_cap else N * B
    pd = torch.empty((N, B, 4), dtype=torch.float32, device=dev); dr = torch.empty_like(pd); rdn = torch.empty_like(pd)
    count = torch.zeros(1, dtype=torch.int32, device=dev)
    pair_id = torch.empty((cap, 2), dtype=torch.int32, device=dev)
    pair_x = torch.empty((cap, 4), dtype=torch.float32, device=dev); pair_g = torch.empty_like(pair_x)
    pair_of_node = torch.empty((N, B), dtype=torch.int32, device=dev)
    w = so3_window(annealed_alpha)
    packed = torch.empty(lib.rnerf_so3_packed_bytes(), dtype=torch.uint8, device=dev)
    check(lib.rnerf_march_all_train(ptr(table), C.byref(spec), ptr(_chk(so3_flat, "so3_flat")), ptr(packed), w.ctypes.data_as(C.c_void_p), ptr(o), ptr(v), B,
                                    float(near), float(far), N, ptr(pd), ptr(dr), ptr(rdn), ptr(count), cap, ptr(pair_id), ptr(pair_x), ptr(pair_g),
                                    ptr(pair_of_node), ptr(order), current_stream()), "rnerf_march_all_train")
    rec = dict(path_pd=pd, path_dr=dr, path_rdn=rdn, window=w, _raw=(count, cap, pair_id, pair_x, pair_g, pair_of_node, B, N))
    return rec if lazy else finalize_pairs(rec)


def finalize_pairs(rec: dict) -> dict:
    """Trim the compacted (ray, node) pair list of a march_all_train record to its count (reads the device counter: synchronises) and put it
    into its run-to-run deterministic order.  Idempotent."""
    if "_raw" not in rec:
        return rec
    count, cap, pair_id, pair_x, pair_g, pair_of_node, B, N = rec.pop("_raw")
    dev = pair_x.device
    n = int(count.item())
    if n > cap:
        raise _lib.RnerfError(f"march_all_train: {n} boundary-shell pairs exceed pair_cap = {cap}")
    pid, px, pg = pair_id[:n], pair_x[:n], pair_g[:n]
    if n > 1 and PAIR_ORDER == "sorted":
        # The kernel hands out pair slots with one atomicAdd per pair, i.e. in arrival order: run-to-run different, and with it the summation
        # order of so3_mlp's weight gradient.  Re-order the compacted list by its (node, ray) key — unique, so the order is a function of the
        # batch alone — and re-point pair_of_node: every kernel downstream sees the same pairs in the same slots on every run.
        # (int32 keys when they fit, sort + index_select instead of argsort + advanced indexing: the same permutation in fewer, shorter
        #  launches — 319 -> 270 us for 208 k pairs, tools/r05/dbg_sort.py)
        small = int(N) * int(B) < 2 ** 31
        key = pid[:, 1] * B + pid[:, 0] if small else pid[:, 1].to(torch.int64) * B + pid[:, 0].to(torch.int64)
        perm = torch.sort(key)[1]
        inv = torch.empty(n, dtype=torch.int32, device=dev)
        inv.index_copy_(0, perm, torch.arange(n, dtype=torch.int32, device=dev))
        pid, px, pg = pid.index_select(0, perm), px.index_select(0, perm), pg.index_select(0, perm)
        flat = pair_of_node.view(-1)
        pair_of_node = torch.where(flat >= 0, inv.index_select(0, flat.clamp(min=0)), flat).view(N, B)
    rec.update(n_pairs=n, pair_id=pid, pair_x=px.contiguous(), pair_g=pg.contiguous(), pair_of_node=pair_of_node)
    return rec

def so3_forward_train(so3_flat: torch.Tensor, window, pts4: torch.Tensor):
    """so3_mlp(annealed_pos_enc(x)) on pts4 [n,4] with saved activations -> (raw [n,4] view into save, save)."""
    lib = _lib.load()
    n = pts4.shape[0]
    save = torch.empty(lib.rnerf_so3_save_bytes(n) // 4, dtype=torch.float32, device=pts4.device)
    check(lib.rnerf_so3_forward_train(ptr(_chk(so3_flat, "so3_flat")), window.ctypes.data_as(C.c_void_p), ptr(_chk(pts4, "pts4")), n, ptr(save),
                                      current_stream()), "rnerf_so3_forward_train")
    raw = save[n * (60 + 4 * 128):n * (60 + 4 * 128 + 4)].view(n, 4)        # (behind it: the ReLU sign bits the dgrad reads)
    return raw, save


def so3_backward(so3_flat: torch.Tensor, window, pts4: torch.Tensor, save: torch.Tensor, d_raw4: torch.Tensor, grads: Optional[torch.Tensor] = None,
                 want_dx: bool = True):
    """Cotangents d_raw4 [nb,4] (nb a multiple of the saved rows) -> dx4 [nb,4]; grads (65411 floats) accumulate when given (nb == n)."""
    lib = _lib.load()
    n, nb = pts4.shape[0], d_raw4.shape[0]
    dy = torch.empty(lib.rnerf_so3_dy_bytes(nb) // 4, dtype=torch.float32, device=pts4.device)
    dx = torch.empty((nb, 4), dtype=torch.float32, device=pts4.device) if want_dx else None
    check(lib.rnerf_so3_backward(ptr(_chk(so3_flat, "so3_flat")), window.ctypes.data_as(C.c_void_p), ptr(pts4), ptr(save), n, ptr(_chk(d_raw4, "d_raw4")),
                                 nb, ptr(dy), ptr(dx), ptr(grads), current_stream()), "rnerf_so3_backward")
    return dx


def so3_pair_jacobian(table: torch.Tensor, spec: Grid, pair_x: torch.Tensor, pair_g: torch.Tensor, raw4: torch.Tensor, J4: torch.Tensor):
    lib = _lib.load()
    n = pair_x.shape[0]
    A = torch.empty((n, 12), dtype=torch.float32, device=pair_x.device); P = torch.empty_like(A)
    check(lib.rnerf_so3_pair_jacobian(ptr(table), C.byref(spec), ptr(pair_x), ptr(pair_g), ptr(_chk(raw4, "raw4")), ptr(_chk(J4, "J4")), n, ptr(A), ptr(P),
                                      current_stream()), "rnerf_so3_pair_jacobian")
    return A, P


def march_adjoint(table: torch.Tensor, spec: Grid, rec: dict, A, P, a_pos: torch.Tensor, a_dir: torch.Tensor, sample_of_node: torch.Tensor,
                  near: float, far: float):
    """The reverse scan of the march -> v4 [n_pairs, 4], the cotangent of the so3 output at every pair."""
    lib = _lib.load()
    N, B = rec["path_pd"].shape[0], rec["path_pd"].shape[1]
    v = torch.zeros((max(rec["n_pairs"], 1), 4), dtype=torch.float32, device=a_pos.device)
    check(lib.rnerf_march_adjoint(ptr(table), C.byref(spec), ptr(rec["path_pd"]), ptr(rec["path_rdn"]), ptr(rec["pair_of_node"]), ptr(A), ptr(P),
                                  ptr(_chk(a_pos, "a_pos")), ptr(_chk(a_dir, "a_dir")), ptr(_chk(sample_of_node, "sample_of_node", torch.int32)), B,
                                  float(near), float(far), N, ptr(v), current_stream()), "rnerf_march_adjoint")
    return v


def nerfmlp_input_grad(params_flat: torch.Tensor, backward: int, dy: torch.Tensor, rows_pd: torch.Tensor, rows_dr: torch.Tensor,
                       node_of_sample: Optional[torch.Tensor], S: int, B: int):
    """d loss / d (position, direction) of the S x B rows of a NerfMLP level -> (d_pos [S,B,4], d_dir [S,B,4])."""
    lib = _lib.load()
    dev = rows_pd.device
    d_pos = torch.empty((S, B, 4), dtype=torch.float32, device=dev); d_dir = torch.empty_like(d_pos)
    check(lib.rnerf_nerfmlp_input_grad(ptr(_chk(params_flat, "params_flat")), int(backward), ptr(dy), ptr(rows_pd), ptr(rows_dr), ptr(node_of_sample),
                                       int(S), int(B), ptr(d_pos), ptr(d_dir), current_stream()), "rnerf_nerfmlp_input_grad")
    return d_pos, d_dir


def env_smooth_backward(rgb_env: torch.Tensor, ps: int, grad_scale: float, d_out: torch.Tensor, loss_sum: torch.Tensor) -> None:
    """train.py:127-130: gradient of the env-map smoothness term into d_out [ps*ps,3]; the un-normalised loss sum goes to loss_sum
    (rnerf_env_smooth_sum_floats(ps) floats: per-workgroup partials that train_stats adds in a fixed order)."""
    lib = _lib.load()
    check(lib.rnerf_env_smooth_backward(ptr(_chk(rgb_env, "rgb_env")), int(ps), float(grad_scale), ptr(_chk(d_out, "d_out")), ptr(loss_sum),
                                        current_stream()), "rnerf_env_smooth_backward")


def train_stats(sums, B: int, two_levels: bool, bg_on: float, env_loss_sum, ps: int, env_on: float, theta, frozen_sq: float, n_all: int,
                stats8: torch.Tensor) -> None:
    """utils.Stats scalars of one step (train.py:147-162) into stats8 (device float[8])."""
    lib = _lib.load()
    check(lib.rnerf_train_stats(ptr(sums), int(B), int(bool(two_levels)), float(bg_on), ptr(env_loss_sum), int(ps), float(env_on),
                                ptr(_chk(theta, "theta")), int(theta.numel()), float(frozen_sq), int(n_all), ptr(_chk(stats8, "stats8")),
                                current_stream()), "rnerf_train_stats")


def integrated_pos_enc(rows_pd: torch.Tensor, rows_dr: torch.Tensor, node_of_sample: Optional[torch.Tensor], S: int, B: int, radii: torch.Tensor,
                       near: float, min_deg: int = 0, max_deg: int = 10, want_gaussians: bool = False):
    """mip.cast_rays(..., "cone") + mip.integrated_pos_enc along the curved ray (rnerf/mip.py:26-175), as the reference's
    commented call sites use them (rnerf/models.py:249-254).  -> enc [S,B,6L] (+ mean4 [S,B,4], cov4 [S,B,4] if want_gaussians)."""
    lib = _lib.load()
    dev = rows_pd.device
    L = int(max_deg) - int(min_deg)
    enc = torch.empty((S, B, 6 * L), dtype=torch.float32, device=dev)
    mean = torch.empty((S, B, 4), dtype=torch.float32, device=dev) if want_gaussians else None
    cov = torch.empty((S, B, 4), dtype=torch.float32, device=dev) if want_gaussians else None
    r = _chk(radii.reshape(-1), "radii")
    if r.numel() != B:
        raise _lib.RnerfError(f"radii must hold one value per ray ({B}), got {r.numel()}")
    if node_of_sample is not None:
        node_of_sample = _chk(node_of_sample, "node_of_sample", torch.int32)
    check(lib.rnerf_integrated_pos_enc(ptr(_chk(rows_pd, "rows_pd")), ptr(_chk(rows_dr, "rows_dr")), ptr(node_of_sample), int(S), int(B), ptr(r), float(near),
                                       int(min_deg), int(max_deg), ptr(mean), ptr(cov), ptr(enc), current_stream()), "rnerf_integrated_pos_enc")
    return (enc, mean, cov) if want_gaussians else enc
