"""ctypes binding of librnerf.so (the C ABI declared in include/rnerf.h).

There is no CPU fallback: if the library is missing or a call fails, this raises.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "librnerf.so")

PREC_F32, PREC_F16X3, PREC_BF16X3, PREC_F16, PREC_BF16, PREC_F16X2, PREC_F16F8 = 0, 1, 2, 3, 4, 5, 6
PRECISIONS = {"f32": PREC_F32, "f16x3": PREC_F16X3, "bf16x3": PREC_BF16X3, "f16": PREC_F16, "bf16": PREC_BF16, "f16x2": PREC_F16X2, "f16f8": PREC_F16F8}
# enum rnerf_backward (include/rnerf.h): arithmetic of the NerfMLP dgrad + wgrad.  "f16x3" = hi + lo f16 parts (fp32-grade, the reference
# differentiates in fp32, train.py:164); "f16" = single f16 parts (11-bit significand); "bf16" = 8-bit significand.
BWD_BF16, BWD_F16, BWD_F16X2 = 0, 1, 2
BWD_F16X3 = BWD_F16X2
BWD_F16X3_LO8 = 4        # f16x3 with the lo planes of the saved operands / gradients stored as e4m3 bytes (decoded to f16 in the wgrad): 3/4 of the bytes
# named for what they compute in (gfx950 has neither an fp32 nor a TF32 matrix path worth the name): "f16x3" = hi + lo f16 planes, 3 MFMAs per
# product; "f16" = one f16 plane; "bf16".  "f32" / "tf32" are the names of rounds 2-4, kept as aliases.
BACKWARDS = {"f16x3": BWD_F16X3, "f16x3lo8": BWD_F16X3_LO8, "f16": BWD_F16, "bf16": BWD_BF16, "f32": BWD_F16X3, "tf32": BWD_F16}
BACKWARD_NAMES = {BWD_F16X3: "f16x3", BWD_F16X3_LO8: "f16x3lo8", BWD_F16: "f16", BWD_BF16: "bf16"}
TWO_PLANE_BACKWARDS = (BWD_F16X3, BWD_F16X3_LO8)
ABI_VERSION = 3          # == RNERF_VERSION of include/rnerf.h; load() refuses a library that answers anything else
NERFMLP_PARAMS = 595844
BKGDMLP_PARAMS = 56963
SO3MLP_PARAMS = 65411


class RnerfError(RuntimeError):
    pass


TABLE_LAYOUTS = {"reference": 0, "bricks": 1}      # enum rnerf_table_layout


class Grid(C.Structure):
    """rnerf_grid (include/rnerf.h): reference VoxMLP.ndim/nmin/nmax (rnerf/ior_utils.py:124-144)."""
    _fields_ = [("dims", C.c_int32 * 3), ("nmin", C.c_double * 3), ("nmax", C.c_double * 3), ("layout", C.c_int32)]

    @classmethod
    def make(cls, ndim, nmin, nmax, layout="reference") -> "Grid":
        """layout: enum rnerf_table_layout — "reference" (flat index x*Gy*Gz + y*Gz + z, rnerf/ior_utils.py:214) or "bricks" (2x2x2 bricks)."""
        g = cls()
        for i in range(3):
            g.dims[i] = int(ndim[i]); g.nmin[i] = float(nmin[i]); g.nmax[i] = float(nmax[i])
        g.layout = TABLE_LAYOUTS[layout] if isinstance(layout, str) else int(layout)
        return g


class Model(C.Structure):
    """rnerf_model (include/rnerf.h): what NerfModel closes over (rnerf/models.py:42-137), for the whole-path entry points."""
    _fields_ = [("table", C.c_void_p), ("grid", Grid), ("near", C.c_double), ("far", C.c_double), ("num_coarse", C.c_int32),
                ("num_fine", C.c_int32), ("num_path", C.c_int32), ("precision", C.c_int32), ("white_bkgd", C.c_int32), ("bd_cut", C.c_int32),
                ("rgb_padding", C.c_double), ("sigma_bias", C.c_double), ("bd_cut_bbox", C.c_double * 6), ("packed_coarse", C.c_void_p),
                ("packed_fine", C.c_void_p), ("bkgd_params", C.c_void_p)]


class TrainCfg(C.Structure):
    """rnerf_train_cfg: the loss terms of train_step.loss_fn that the shipped configs switch on (train.py:75-162)."""
    _fields_ = [("backward", C.c_int32), ("randomized", C.c_int32), ("use_random_choice", C.c_int32), ("bg_patch_size", C.c_int32),
                ("bg_weight", C.c_double), ("bg_smooth_weight", C.c_double), ("annealed_alpha", C.c_double), ("frozen_sq", C.c_double),
                ("frozen_count", C.c_int64), ("aux_stream", C.c_void_p), ("coresident_bkgd_wgrad", C.c_int32), ("grads_stream", C.c_void_p),
                ("aux2_stream", C.c_void_p)]


class AdamCfg(C.Structure):
    """rnerf_adam_cfg: optax.adam + the reference's learning-rate schedule and gradient clipping (train.py:169-183,312-317)."""
    _fields_ = [("lr_init", C.c_double), ("lr_final", C.c_double), ("lr_delay_mult", C.c_double), ("max_steps", C.c_int64),
                ("lr_delay_steps", C.c_int64), ("b1", C.c_double), ("b2", C.c_double), ("eps", C.c_double), ("weight_decay_mult", C.c_double),
                ("grad_max_val", C.c_double), ("grad_max_norm", C.c_double), ("n_all", C.c_int64), ("lr_override", C.c_double),
                ("use_lr_override", C.c_int32), ("skip_nonfinite", C.c_int32)]


class Prefetch(C.Structure):
    """rnerf_prefetch: the next batch's march on a side stream, forked behind the last wgrad of rnerf_train_forward_backward."""
    _fields_ = [("origins", C.c_void_p), ("viewdirs", C.c_void_p), ("path_pd", C.c_void_p), ("path_dr", C.c_void_p), ("side_stream", C.c_void_p),
                ("beside_wgrad", C.c_int32)]


LEVEL_FLOATS = 9
ADAM_SCRATCH_FLOATS = 3076
_vp, _i32, _i64, _dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
_GP = C.POINTER(Grid)
_MP, _TP, _AP = C.POINTER(Model), C.POINTER(TrainCfg), C.POINTER(AdamCfg)

# name -> (restype, argtypes); must list every symbol declared in include/rnerf.h
SIGNATURES = {
    "rnerf_last_error": (C.c_char_p, []),
    "rnerf_version": (C.c_int, []),
    "rnerf_device_cus": (C.c_int, []),
    "rnerf_grid_prefilter": (C.c_int, [_vp, _vp, _vp, C.POINTER(_i32 * 3), C.c_int, _dbl, _vp]),
    "rnerf_grid_table_floats": (C.c_size_t, [_GP]),
    "rnerf_grid_build_table": (C.c_int, [_vp, _vp, _GP, _vp]),
    "rnerf_grid_query": (C.c_int, [_vp, _GP, _vp, _i64, _vp, _vp, _vp]),
    "rnerf_march": (C.c_int, [_vp, _GP, _vp, _vp, _i32, _dbl, _dbl, _i32, _vp, _vp, _vp, _vp, _vp]),
    "rnerf_nerfmlp_packed_bytes": (C.c_size_t, [C.c_int]),
    "rnerf_nerfmlp_pack": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "rnerf_nerfmlp_forward": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _i32, _i32, _vp, _i32, _vp]),
    "rnerf_bkgd_forward": (C.c_int, [_vp, _vp, _i32, _i64, _dbl, _vp, _vp]),
    "rnerf_composite": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _vp, C.c_int, _dbl, _dbl,
                                  _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.c_int, C.POINTER(C.c_double * 6), _vp]),
    "rnerf_loss_reduce": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp]),
    "rnerf_env_smooth_sum_floats": (C.c_size_t, [_i32]),
    "rnerf_env_smooth_backward": (C.c_int, [_vp, _i32, _dbl, _vp, _vp, _vp]),
    "rnerf_train_stats": (C.c_int, [_vp, _i32, _i32, _dbl, _vp, _i32, _dbl, _vp, _i64, _dbl, _i64, _vp, _vp]),
    "rnerf_composite_backward": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _i32, _vp, _dbl, _dbl, _vp, _vp, _vp, _vp, _vp, _dbl, _dbl,
                                           _vp, _vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double * 6), _vp]),
    "rnerf_nerfmlp_save_bytes": (C.c_size_t, [_i64, C.c_int]),
    "rnerf_nerfmlp_dy_bytes": (C.c_size_t, [_i64, C.c_int]),
    "rnerf_nerfmlp_bwd_packed_bytes": (C.c_size_t, []),
    "rnerf_nerfmlp_wgrad_workspace_bytes": (C.c_size_t, []),
    "rnerf_nerfmlp_forward_train": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _i32, _i32, _vp, _vp, C.c_int, _i32, _vp]),
    "rnerf_nerfmlp_pack_bwd": (C.c_int, [_vp, C.c_int, _vp, _vp]),
    "rnerf_nerfmlp_dgrad": (C.c_int, [_vp, _vp, C.c_int, C.c_int, _vp, _vp, _i64, _vp, _vp]),
    "rnerf_nerfmlp_wgrad": (C.c_int, [C.c_int, C.c_int, _vp, _vp, _i64, _vp, _vp, _vp]),
    "rnerf_voxelize": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _vp, _GP, _i32, _dbl, _dbl, _vp, _vp, _vp, _vp]),
    "rnerf_voxelize_samples": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _vp, _GP, _i32, _vp, _vp, _vp]),
    "rnerf_voxelize_majority": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _dbl, _dbl, _vp, _vp, _vp]),
    "rnerf_so3_query": (C.c_int, [_vp, _GP, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "rnerf_march_all": (C.c_int, [_vp, _GP, _vp, _vp, _vp, _vp, _vp, _i32, _dbl, _dbl, _i32, _vp, _vp, _vp, _vp, _vp]),
    "rnerf_generate_rays": (C.c_int, [_vp, _i32, _dbl, _dbl, _dbl, _dbl, _dbl, _i32, _i32, _i32, _vp, _vp, _vp, _vp]),
    "rnerf_sample_batch": (C.c_int, [_vp, _i32, _i32, _dbl, _dbl, _dbl, _dbl, _dbl, _i32, _i32, _vp, _i32, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rnerf_stratified_u": (C.c_int, [_vp, _i32, _i32, _vp, _vp]),
    "rnerf_bkgd_save_bytes": (C.c_size_t, [_i64]),
    "rnerf_bkgd_dy_bytes": (C.c_size_t, [_i64]),
    "rnerf_bkgd_forward_train": (C.c_int, [_vp, _vp, _i32, _i64, _dbl, _vp, _vp, _vp]),
    "rnerf_bkgd_backward": (C.c_int, [_vp, _vp, _vp, _i64, _dbl, _vp, _vp, _vp, _vp]),
    "rnerf_bkgd_backward_dgrad": (C.c_int, [_vp, _vp, _vp, _i64, _dbl, _vp, _vp, _vp]),
    "rnerf_bkgd_backward_wgrad": (C.c_int, [_vp, _vp, _i64, _vp, C.c_int, _vp]),
    "rnerf_theta_sumsq": (C.c_int, [_vp, _i64, _vp, _vp]),
    "rnerf_march_all_train": (C.c_int, [_vp, _GP, _vp, _vp, _vp, _vp, _vp, _i32, _dbl, _dbl, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "rnerf_so3_packed_bytes": (C.c_size_t, []),
    "rnerf_so3_save_bytes": (C.c_size_t, [_i64]),
    "rnerf_so3_dy_bytes": (C.c_size_t, [_i64]),
    "rnerf_so3_forward_train": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp]),
    "rnerf_so3_backward": (C.c_int, [_vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp, _vp]),
    "rnerf_so3_pair_jacobian": (C.c_int, [_vp, _GP, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp]),
    "rnerf_march_adjoint": (C.c_int, [_vp, _GP, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _dbl, _dbl, _i32, _vp, _vp]),
    "rnerf_nerfmlp_input_grad": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp]),
    "rnerf_resample": (C.c_int, [_vp, _vp, _i32, _i32, _vp, _i32, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp]),
    "rnerf_integrated_pos_enc": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _vp, _dbl, _i32, _i32, _vp, _vp, _vp, _vp]),
    # whole-path entry points (csrc/pipeline.hip)
    "rnerf_rng_split3": (C.c_int, [_vp, _vp, _vp]),
    "rnerf_rng_forward": (C.c_int, [_vp, _i32, _i32, _i32, _vp, _vp, _vp]),
    "rnerf_stratified_u_dev": (C.c_int, [_vp, _i32, _i32, _vp, _vp]),
    "rnerf_forward_workspace_bytes": (C.c_size_t, [_MP, _i32]),
    "rnerf_forward": (C.c_int, [_MP, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp]),
    "rnerf_train_workspace_bytes": (C.c_size_t, [_MP, _TP, _i32]),
    "rnerf_train_forward_backward": (C.c_int, [_MP, _TP, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _i32, C.POINTER(Prefetch), _vp]),
    "rnerf_adam_update": (C.c_int, [_AP, _vp, _vp, _vp, _vp, _i64, _vp, _i64, _vp, _vp, _vp]),
    "rnerf_graph_begin": (C.c_int, [_vp]),
    "rnerf_graph_end": (C.c_int, [_vp, C.POINTER(C.c_void_p)]),
    "rnerf_graph_launch": (C.c_int, [_vp, _vp]),
    "rnerf_graph_destroy": (C.c_int, [_vp]),
    "rnerf_fork": (C.c_int, [_vp, _vp]),
    "rnerf_join": (C.c_int, [_vp, _vp]),
}

_lib: Optional[C.CDLL] = None


def load(path: Optional[str] = None) -> C.CDLL:
    """Load librnerf.so and bind every entry point; raises RnerfError if the HIP library is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH          # load(path) first thing in a process binds an alternative build (tools' profiling variants) for everything after
    if not os.path.exists(p):
        raise RnerfError(f"{p} not found: build it with `python -m samplenerfro_amd.build` "
                         "(there is no CPU fallback for the hot path)")
    lib = C.CDLL(p)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if a declared symbol is missing
        fn.restype = res
        fn.argtypes = args
    got = lib.rnerf_version()
    if got != ABI_VERSION:          # struct layouts (rnerf_grid / rnerf_model / rnerf_train_cfg / rnerf_adam_cfg) are part of the version
        raise RnerfError(f"{p} speaks ABI version {got}, this binding was written for {ABI_VERSION} (include/rnerf.h RNERF_VERSION): rebuild with "
                         "`python -m samplenerfro_amd.build --force`")
    if path is None or _lib is None:
        _lib = lib
    return lib


def check(status: int, what: str = "") -> None:
    if status != 0:
        msg = load().rnerf_last_error()
        raise RnerfError(f"{what} failed ({status}): {msg.decode() if msg else '?'}")


def ptr(t) -> Optional[int]:
    """Device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    return t.data_ptr()


def current_stream() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream
