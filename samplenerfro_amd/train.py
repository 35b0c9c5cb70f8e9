"""Host-side mirror of the reference's training step (train.py:58-183) for the radiance stages.

    state = TrainState.create(model, variables, flags)
    state, stats, rng = train_step(model, rng, state, batch)          # same call as train.py:58

`jax.value_and_grad(loss_fn)` is replaced by explicit backward kernels in librnerf.so, in reverse order of the forward:

    loss reductions (rnerf_loss_reduce)                     train.py:89-92,105
    compositing + activations backward, fine then coarse    rnerf_composite_backward
    NerfMLP dgrad + wgrad on the matrix cores               rnerf_nerfmlp_dgrad / rnerf_nerfmlp_wgrad
    background MLP backward (per-ray bkgd + env-map patch)  rnerf_bkgd_backward

No gradient flows through the resampling (lax.stop_gradient, rnerf/model_utils.py:407-411), the marched path (the
path_sampler parameters are labelled "zero" in the radiance stages, train.py:286-293) or trans * stop_gradient(rgb_bkgd)
(rnerf/model_utils.py:309).  The terms scaled by annealing_rate are identically zero (train.py:156 sets it to 0.0).

`jax.lax.pmean` (train.py:166-167) is ONE all-reduce (RCCL when the process group is nccl) over a single flat buffer
holding every gradient and the stats vector.  The optimiser (optax.adam behind multi_transform, train.py:312-317) is
plain torch arithmetic on that flat buffer: weights and optimiser state stay on the PyTorch side of the boundary.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
import weakref
from typing import Any, Dict, Optional

import numpy as np
import torch

from . import _lib, distributed, ops, prng
from .models import BKGD_MLP_SHAPES, NERF_MLP_SHAPES, SO3_MLP_SHAPES, NerfModel, make_variables
from .utils import Rays, Stats, learning_rate_decay

# Experiment switches of the step's stream placement: module attributes (tools set them: `train._MARCH_EARLY = True`), never environment reads.
_MARCH_EARLY = False                    # see train_step (measured slower: off)
# CUs the training forward leaves to the next step's march (issued at the START of the step, see train_step); 0 = march after the wgrad.
# Measured slower (6.8 -> 7.0 ms): the march's 256 one-wave workgroups are dispatched over ALL CUs, and every MLP workgroup (a whole CU
# each) waits for the wave on its CU; confining the march with a CU-masked stream (hipExtStreamCreateWithCUMask) serialised the two
# streams instead (7.6 ms), and packing it into 16-wave workgroups on 16 / 32 CUs makes it bound by those CUs' gather units (9.7 / 7.5 ms):
# the march wants one wave on EVERY CU, the MLP kernels want every CU whole.  Off.
_MARCH_RESERVE = 0
_AUX_STREAM = True                      # the second stream of rnerf_train_cfg (False: everything on one stream)
_AUX2_STREAM = False                    # a third stream for the background backward of small hierarchical batches (see train_cfg)
_CORESIDENT_BKGD_WGRAD = False          # the background-MLP weight gradient as a co-resident kernel beside the NerfMLP wgrad (see train_cfg)
_RANGE_RETRY_LAG = 2                    # range_retry="lag": the count of step k - 2 is read after step k has been queued (train_step)
_LAG_SLOTS = 8
_SKIP_NONFINITE_UPDATES = True          # rnerf_adam_cfg.skip_nonfinite: an update with an inf / NaN gradient entry is skipped and counted (train_step)
_ALL_CHAIN_BESIDE_WGRAD = True          # stage all*: the march's adjoint chain on the side stream beside the NerfMLP wgrad (see train_step)

_N_STATS = 8        # loss, loss_c, loss_bg, loss_bg_smooth, weight_l2, (3 spare)


class TrainState:
    """flax TrainState (step, params, opt_state) with every trained parameter in ONE flat fp32 buffer.

    Segments in order: coarse_mlp, fine_mlp (if N_f > 0), bkgd_mlp — the "adam_lr_scheduler" labels of train.py:286-293;
    path_sampler/so3_mlp is labelled "zero" and stays outside.  variables["flat"][name] are views into `theta`, so an
    optimiser step is visible to the kernels without copies (the MFMA operand streams are re-packed lazily, keyed on the
    buffer version)."""

    def __init__(self, step, theta, mu, nu, variables, segments, lr_fn, flags=None):
        self.step, self.theta, self.mu, self.nu = step, theta, mu, nu
        self.flags = flags
        self.variables, self.segments, self.lr_fn = variables, segments, lr_fn
        self.grads = torch.zeros(theta.numel() + _N_STATS, dtype=torch.float32, device=theta.device)   # + the stats vector
        self.frozen_sq: Optional[torch.Tensor] = None
        self.next_path = None          # PathHandle of the next step's rays when train_step was given next_rays
        self._lr_fn_default = lr_fn    # the reference schedule (rnerf_adam_update evaluates it on the device; a replaced lr_fn is passed by value)
        # device-resident optimiser state of rnerf_adam_update: the update count and its scratch (learning rate, bias corrections, clip)
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=theta.device) if theta.is_cuda else None
        self._step_dev_value = 0
        self.adam_scratch = torch.zeros(_lib.ADAM_SCRATCH_FLOATS, dtype=torch.float32, device=theta.device) if theta.is_cuda else None
        self._staged_bad: Optional[int] = None
        self.range_retries = 0         # steps train_step(range_retry=True) re-ran in the range-safe arithmetic
        self.range_retry_failures = 0  # re-runs whose range-safe gradient was non-finite too: that update stayed skipped (a warning tells)
        self._model_ref = None         # weakref to the model of the last lagged step: state_dict() settles the pending steps with it
        self._lag_pending = []         # range_retry="lag": the steps whose non-finite count has not been looked at yet (at most _RANGE_RETRY_LAG)
        self._lag_host = torch.zeros(_LAG_SLOTS, dtype=torch.float32).pin_memory() if theta.is_cuda else None
        self._lag_count = 0
        self.last_retry_stats = None

    def nonfinite_grads(self) -> int:
        """Non-finite gradient entries met by the last rnerf_adam_update (reads a device scalar: synchronises).  Non-zero means a row's
        f16 gradient chain left its 2^10 of headroom (backward modes "f16x3" / "f16") or the loss itself went non-finite."""
        if self._staged_bad is not None:                 # the staged sequence of a range_retry step counted on the host side
            return self._staged_bad
        return int(self.adam_scratch[3].item()) if self.adam_scratch is not None else 0

    def sync_step_counter(self) -> None:
        """Make the device-resident update count equal to self.step (they drift apart when the host sets step: restore, tests)."""
        if self.step_dev is not None and self._step_dev_value != self.step:
            self.step_dev.fill_(int(self.step))
            self._step_dev_value = self.step

    @classmethod
    def create(cls, model: NerfModel, variables: Dict[str, Any], flags) -> "TrainState":
        names = ["coarse_mlp"] + (["fine_mlp"] if model.num_fine_samples > 0 else []) + ["bkgd_mlp"]
        stage = getattr(flags, "stage", "radiance")
        if stage.startswith("all"):        # train.py:302-310: path_sampler joins the "adam_lr_scheduler" group
            names.append("so3_mlp")
        elif stage.startswith("ior"):      # train.py:294-301: only path_sampler is trained, the rest is labelled "zero"
            names = ["so3_mlp"]
        shapes = {"coarse_mlp": NERF_MLP_SHAPES, "fine_mlp": NERF_MLP_SHAPES, "bkgd_mlp": BKGD_MLP_SHAPES, "so3_mlp": SO3_MLP_SHAPES}
        parts = [model._flat(variables, n, shapes[n]).detach().reshape(-1).float() for n in names]
        theta = torch.cat(parts).contiguous()
        segments, off = {}, 0
        flat = dict(variables.get("flat", {}))
        for n, p in zip(names, parts):
            segments[n] = (off, off + p.numel())
            flat[n] = theta[off:off + p.numel()]
            off += p.numel()
        new_vars = make_variables(flat)
        lr_fn = lambda count: learning_rate_decay(count, flags.lr_init, flags.lr_final, flags.max_steps, flags.lr_delay_steps,
                                                  flags.lr_delay_mult)
        return cls(0, theta, torch.zeros_like(theta), torch.zeros_like(theta), new_vars, segments, lr_fn, flags)

    def grad_view(self, name: str) -> torch.Tensor:
        lo, hi = self.segments[name]
        return self.grads[lo:hi]

    def apply_gradients(self, grads: torch.Tensor, b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8) -> "TrainState":
        """optax.adam(learning_rate=schedule): scale_by_adam then scale_by_schedule(count) with count = updates so far."""
        count = self.step
        lr = self.lr_fn(count)
        t = count + 1
        # (seven elementwise launches, ~35 us for 1.25 M parameters; torch._fused_adam_ — a multi-tensor-apply kernel — takes 97 us
        #  on one big tensor)
        self.mu.mul_(b1).add_(grads, alpha=1 - b1)
        self.nu.mul_(b2).addcmul_(grads, grads, value=1 - b2)
        denom = (self.nu / (1 - b2 ** t)).sqrt_().add_(eps)
        self.theta.addcdiv_(self.mu, denom, value=-lr / (1 - b1 ** t))
        self.step = t
        return self

    def state_dict(self) -> Dict[str, Any]:
        """The parameters for good (checkpoint, end of training).  range_retry="lag" may still hold up to two steps whose non-finite count has
        not been looked at: they are settled first (a skipped batch is re-run) — a checkpoint never loses a batch silently.  Raises when the
        model those steps ran on is gone (flush_range_retry(model, state) was the caller's job then)."""
        if self._lag_pending:
            model = self._model_ref() if self._model_ref is not None else None
            if model is None:
                raise RuntimeError("TrainState.state_dict(): steps of range_retry='lag' are still pending and their model is gone; "
                                   "call flush_range_retry(model, state) before reading the parameters")
            flush_range_retry(model, self)
        return {"step": self.step, "theta": self.theta, "mu": self.mu, "nu": self.nu}

    def restore_flax(self, state: Dict[str, Any]) -> "TrainState":
        """Resume from a reference-format TrainState dict (checkpoint.restore_checkpoint; train.py:322): parameters, step and — when
        present — the Adam moments of the optax "adam_lr_scheduler" group."""
        from .checkpoint import find_params, flat_from_params
        flat = flat_from_params(find_params(state), self.theta.device, tuple(self.segments))
        for name, (lo, hi) in self.segments.items():
            self.theta[lo:hi].copy_(flat[name])
        self.step = int(state.get("step", 0))
        try:
            adam = state["opt_state"]["inner_states"]["adam_lr_scheduler"]["inner_state"]["0"]
            for which, buf in (("mu", self.mu), ("nu", self.nu)):
                m = flat_from_params(adam[which]["params"], self.theta.device, tuple(self.segments))
                for name, (lo, hi) in self.segments.items():
                    buf[lo:hi].copy_(m[name])
        except (KeyError, TypeError, ValueError):
            pass                                            # weights-only checkpoint: the moments restart from zero
        # parameters outside theta (the frozen path_sampler of the radiance stages) come from the checkpoint too; the cached sum of
        # their squares (weight_l2, norm clipping) is keyed on the tensor and its version, so it follows this copy
        frozen = flat_from_params(find_params(state), self.theta.device, ("so3_mlp",)) if "so3_mlp" not in self.segments else {}
        cur = self.variables.get("flat", {}).get("so3_mlp")
        if "so3_mlp" in frozen and cur is not None:
            cur.copy_(frozen["so3_mlp"])
        self.frozen_sq = None
        return self

    def load_state_dict(self, d: Dict[str, Any]) -> None:
        self._lag_pending.clear()          # pending steps belong to the parameters that are being replaced
        self.step = int(d["step"])
        for k in ("theta", "mu", "nu"):
            getattr(self, k).copy_(d[k])
        self.frozen_sq = None


def _bwd_packed(model: NerfModel, state: TrainState, name: str, backward: int) -> torch.Tensor:
    """The transposed (dgrad) weight stream of one NerfMLP, re-packed when the parameters (or the backward mode) changed."""
    cache = model.__dict__.setdefault("_packed_bwd", {})
    flat = state.variables["flat"][name]
    ent = cache.get(name)
    if ent is None or ent[0]() is not flat or ent[1] != flat._version or ent[3] != backward:
        buf = ops.nerfmlp_pack_bwd(flat, ent[2] if ent is not None else None, backward)
        cache[name] = (weakref.ref(flat), flat._version, buf, backward)
    return cache[name][2]


def backward_mode(flags, model: NerfModel) -> int:
    """flags.backward_precision: "f16x3" (default; hi + lo f16 parts, fp32-grade like the reference's jax.value_and_grad, train.py:164),
    "f16" (single f16 parts: the 11-bit class of Ampere's TF32 matmuls) or "bf16" (8-bit significand, the round-1 arithmetic).  "f32" / "tf32"
    are accepted as the older names of the first two."""
    name = getattr(flags, "backward_precision", "f16x3")
    if name not in _lib.BACKWARDS:
        raise ValueError(f"backward_precision must be one of {sorted(_lib.BACKWARDS)}")
    mode = _lib.BACKWARDS[name]
    if model.precision == _lib.PREC_BF16X3 and mode == _lib.BWD_BF16:
        return mode       # the range-safe training arithmetic (range_safe below; fp32's exponent range end to end, 8-bit gradients)
    if model.precision != _lib.PREC_F16X3 and not (model.precision == _lib.PREC_F16 and mode not in _lib.TWO_PLANE_BACKWARDS):
        raise ValueError('training is built on the f16x3 forward (NerfModel(precision="f16x3")), or — one MFMA per product, the north-star '
                         'arithmetic as a labelled leg — on the f16 forward with backward_precision "f16" / "bf16"; the other precisions are inference modes')
    return mode


def _ior_stage_step(model: NerfModel, rng, state: TrainState, batch, flags):
    """Stage ior* exactly as the reference ships it (train.py:133-146,156-162): the only data term, loss_nrm =
    compute_normal_loss_and_smooth(...)[0] (eikonal_utils.py:84-98, which returns the constant 0.0 for it), enters the objective as
    annealing_rate * loss_nrm with annealing_rate hard-wired to 0.0 (train.py:156), so what jax.value_and_grad returns for the trained
    group (path_sampler, train.py:294-301) is the weight-decay term alone: 2 * weight_decay_mult * theta / n_all.  Plain tensor
    arithmetic on the flat buffer — there is no kernel to run."""
    rng, _key_0, _key_1 = prng.split(np.asarray(rng, np.uint32), 3)
    annealed = float(np.asarray(batch["annealed_alpha"]).reshape(-1)[0])
    variables = state.variables
    others = [v for k, v in variables["flat"].items() if k not in state.segments]
    n_all = state.theta.numel() + sum(int(v.numel()) for v in others)
    sq = (state.theta.double() ** 2).sum() + sum((v.double() ** 2).sum() for v in others)
    weight_l2 = (sq / n_all).float()
    grads = state.grads[:state.theta.numel()]
    grads.copy_(state.theta).mul_(2.0 * flags.weight_decay_mult / n_all)
    distributed.allreduce_mean_([grads])
    if flags.grad_max_val > 0:
        grads.clamp_(-flags.grad_max_val, flags.grad_max_val)
    if flags.grad_max_norm > 0:
        og = [v * (2.0 * flags.weight_decay_mult / n_all) for v in others]
        if flags.grad_max_val > 0:
            og = [v.clamp(-flags.grad_max_val, flags.grad_max_val) for v in og]
        norm = torch.sqrt((grads * grads).sum() + sum((v * v).sum() for v in og))
        grads.mul_(torch.clamp(flags.grad_max_norm / (1e-7 + norm), max=1.0))
    state.apply_gradients(grads)
    zero = torch.zeros((), device=state.theta.device)
    stats = Stats(loss=zero, psnr=zero, loss_c=zero, psnr_c=zero, weight_l2=weight_l2, loss_sp=0.0, loss_nrm=0.0, annealing_rate=annealed,
                  coarse_alpha_target=0.0, fine_alpha_target=0.0, loss_bg=zero, loss_bg_c=0.0, loss_bg_smooth=0.0)
    return state, stats, rng


def _all_stage_backward(model: NerfModel, state: TrainState, variables, ctx, dy_c, d_bk_dirs, bwd: int, annealed: float, taps) -> None:
    """Stage all*: d loss / d so3_mlp through the marched path (csrc/ior_train_kernels.inc).  Only the coarse level and the background
    colour reach the path: sample_pdf stops the gradient of everything it returns (rnerf/model_utils.py:406-411), ray_dist is
    stop_gradient'ed (eikonal_utils.py:121), and |ray_dir| = 1 makes the compositing's delta independent of the direction."""
    B, Nc, N = ctx["B"], model.num_coarse_samples, model.num_samples
    rec = ops.finalize_pairs(ctx["march_rec"])      # the step's one host synchronisation: here, with the forward and the NerfMLP backward already queued
    coarse_flat = variables["flat"]["coarse_mlp"]
    so3_flat = variables["flat"]["so3_mlp"]
    a_pos, a_dir = ops.nerfmlp_input_grad(coarse_flat, bwd, dy_c, ctx["path_pd"], ctx["path_dr"], ctx["jit"], Nc, B)
    a_dir[Nc - 1] += d_bk_dirs[:B]                     # bkgd_mlp reads the direction of the LAST coarse sample (rnerf/models.py:303)
    sample_of_node = torch.full((N,), -1, dtype=torch.int32, device=a_pos.device)
    sample_of_node[ctx["jit"].long()] = torch.arange(Nc, dtype=torch.int32, device=a_pos.device)
    n = rec["n_pairs"]
    if n > 0:
        raw, save = ops.so3_forward_train(so3_flat, rec["window"], rec["pair_x"])
        eye = torch.zeros((3 * n, 4), dtype=torch.float32, device=a_pos.device)
        for j in range(3):
            eye[j * n:(j + 1) * n, j] = 1.0
        J = ops.so3_backward(so3_flat, rec["window"], rec["pair_x"], save, eye)                   # rows of d raw / d x
        A, P = ops.so3_pair_jacobian(model.table, model.spec, rec["pair_x"], rec["pair_g"], raw.contiguous(), J)
    else:
        A = P = torch.zeros((1, 12), dtype=torch.float32, device=a_pos.device)
    v = ops.march_adjoint(model.table, model.spec, rec, A, P, a_pos, a_dir, sample_of_node, model.near, model.far)
    if n > 0:
        ops.so3_backward(so3_flat, rec["window"], rec["pair_x"], save, v, grads=state.grad_view("so3_mlp"), want_dx=False)
    if taps is not None:
        taps.update(a_pos=a_pos, a_dir=a_dir, n_pairs=n, v_pairs=v)


def frozen_sq_of(state: TrainState, variables):
    """(sum of squares, count, cache key, weak reference) of the variables outside theta — the frozen path_sampler of the radiance stages —
    for weight_l2 / the norm clip (train.py:147-153,174-180); cached per (tensor object, version)."""
    so3 = variables.get("flat", {}).get("so3_mlp") if "so3_mlp" not in state.segments else None      # trained in stage all*: part of theta
    fkey = (id(so3), so3._version) if so3 is not None else None
    if state.frozen_sq is None or state.frozen_sq[2] != fkey or (so3 is not None and state.frozen_sq[3]() is not so3):
        # restore_flax / graft_pretrained / an in-place load of new so3 weights invalidate it
        state.frozen_sq = (float((so3.double() ** 2).sum()) if so3 is not None else 0.0, so3.numel() if so3 is not None else 0, fkey,
                           weakref.ref(so3) if so3 is not None else None)
    return state.frozen_sq


def train_cfg(model: NerfModel, state: TrainState, flags, annealed: float) -> "_lib.TrainCfg":
    c = _lib.TrainCfg()
    c.backward = backward_mode(flags, model)
    c.randomized, c.use_random_choice = int(bool(flags.randomized)), int(model.use_random_choice)
    c.bg_patch_size = int(flags.bg_patch_size) if flags.bg_smooth_weight > 0 else 0
    c.bg_weight, c.bg_smooth_weight, c.annealed_alpha = float(flags.bg_weight), float(flags.bg_smooth_weight), float(annealed)
    fs = frozen_sq_of(state, state.variables)
    c.frozen_sq, c.frozen_count = fs[0], fs[1]
    # the second stream of rnerf_train_cfg: what depends on the parameters only (operand packing, zeroing the gradient buffer, sum theta^2)
    # runs there beside the head of the step (_AUX_STREAM = False: everything on one stream)
    if _AUX_STREAM and hasattr(model, "tail_stream"):
        c.aux_stream = model.tail_stream().cuda_stream
        # a third stream for the background backward of small hierarchical batches: opt-in.  It pays at 256 rays (1.48 -> 1.39 ms) when it
        # gets a hardware queue of its own, and costs 30 % when it lands on the queue of the march or of the main stream — which is decided by
        # how many streams the process has created (GPU_MAX_HW_QUEUES = 4)
        if hasattr(model, "tail2_stream") and _AUX2_STREAM:
            c.aux2_stream = model.tail2_stream().cuda_stream
        # _CORESIDENT_BKGD_WGRAD (experiment, off): the background-MLP weight gradient as a co-resident kernel beside the NerfMLP wgrad.  Measured
        # neutral at 4096 x 128 (what it saves on the critical path, ~0.12 ms, the wgrad loses to the extra waves: 2.04 -> 2.2-2.4 ms),
        # +1-2 % at 1024 rays x (64 + 128)
        c.coresident_bkgd_wgrad = int(_CORESIDENT_BKGD_WGRAD)
    return c


def adam_cfg(state: TrainState, flags, lr_override: Optional[float] = None) -> "_lib.AdamCfg":
    a = _lib.AdamCfg()
    a.lr_init, a.lr_final, a.lr_delay_mult = float(flags.lr_init), float(flags.lr_final), float(flags.lr_delay_mult)
    a.max_steps, a.lr_delay_steps = int(flags.max_steps), int(flags.lr_delay_steps)
    a.b1, a.b2, a.eps = 0.9, 0.999, 1e-8
    a.weight_decay_mult, a.grad_max_val, a.grad_max_norm = float(flags.weight_decay_mult), float(flags.grad_max_val), float(flags.grad_max_norm)
    a.n_all = state.theta.numel() + frozen_sq_of(state, state.variables)[1]
    a.use_lr_override = int(lr_override is not None)      # an explicit switch: a replaced schedule may return exactly 0.0
    a.lr_override = float(lr_override) if lr_override is not None else 0.0
    # a gradient with an inf / NaN entry (a row outside the f16-based arithmetic's range) never reaches the parameters: the update is
    # skipped and counted (TrainState.nonfinite_grads()); train_step(range_retry=True) then re-runs the batch in the range-safe arithmetic
    a.skip_nonfinite = int(_SKIP_NONFINITE_UPDATES)
    return a


def _adam_update_on_device(state: TrainState, flags) -> None:
    """train.py:169-183 + optax.adam on the flat buffers as ONE C call (rnerf_adam_update): weight-decay gradient, value / norm clip, Adam
    with the reference's schedule from the device-resident step counter, and the non-finite guard (adam_cfg: an update that met an inf / NaN
    gradient entry writes nothing and is counted).  state.grads holds the (all-reduced) gradient of the data terms."""
    lib = _lib.load()
    default_lr = state._lr_fn_default is state.lr_fn
    a = adam_cfg(state, flags, None if default_lr else float(state.lr_fn(state.step)))
    state.sync_step_counter()
    fs = frozen_sq_of(state, state.variables)
    frozen = state.variables["flat"].get("so3_mlp") if fs[1] > 0 else None
    _lib.check(lib.rnerf_adam_update(C.byref(a), state.theta.data_ptr(), state.mu.data_ptr(), state.nu.data_ptr(), state.grads.data_ptr(), state.theta.numel(),
                                     _lib.ptr(frozen), fs[1], state.step_dev.data_ptr(), state.adam_scratch.data_ptr(), _lib.current_stream()), "rnerf_adam_update")
    state.step += 1
    state._step_dev_value = state.step
    state._staged_bad = None
    _bump(state.theta)


def _train_step_whole(model: NerfModel, rng, state: TrainState, batch, flags, jitter, u_fine, path, next_rays):
    """train_step for the radiance stages through the whole-path entry points (include/rnerf.h, csrc/pipeline.hip)."""
    lib = _lib.load()
    rng, key_0, key_1 = prng.split(np.asarray(rng, np.uint32), 3)
    annealed = float(np.asarray(batch["annealed_alpha"]).reshape(-1)[0])
    rays: Rays = batch["rays"]
    o, v = ops._chk(rays.origins, "origins"), ops._chk(rays.viewdirs, "viewdirs")
    pixels = ops._chk(batch["pixels"][..., :3].contiguous(), "pixels")
    B = o.shape[0]
    env = None
    if flags.bg_smooth_weight > 0:
        env = ops._chk(batch["env_rays"].viewdirs.reshape(-1, 3), "env_rays.viewdirs")
        if env.shape[0] != flags.bg_patch_size ** 2:
            raise ValueError("env_rays.viewdirs must hold bg_patch_size^2 directions")
    m = model.c_model()
    c = train_cfg(model, state, flags, annealed)
    st = _lib.current_stream()
    keys = model._keys_dev(key_0, key_1)
    jit = model._jitter_dev(jitter) if jitter is not None else None
    u, per_ray = None, 0
    if u_fine is not None:
        u = ops._chk(u_fine, "u_fine"); per_ray = 1 if u.dim() == 2 else 0
    pd = dr = None
    if path is not None:
        if path.batch != B:
            raise ValueError("path handle was marched for a different batch size")
        cur = torch.cuda.current_stream()
        cur.wait_event(path.event)
        pd, dr = path.pd, path.dr
        pd.record_stream(cur); dr.record_stream(cur)
    ws = model._workspace("train", lib.rnerf_train_workspace_bytes(C.byref(m), C.byref(c), B))
    G = state.grads
    n_theta = state.theta.numel()
    # the march of the NEXT step goes to the model's side stream, forked behind the last wgrad inside the call: it runs beside the
    # background-MLP backward, the loss tail, the all-reduce and the optimiser update
    nxt, next_path = None, None
    if next_rays is not None:
        next_path, nxt = model.prefetch_slot(next_rays)
    # jax.lax.pmean of gradients and stats (train.py:166-167).  With more than one rank the NerfMLP segments (95 % of the bytes) start their
    # all-reduce on a side stream the call orders behind the last wgrad, beside the background-MLP backward and the loss tail still queued
    # on the main stream; the background-MLP gradients and the stats follow in a small second one.
    comm = model.comm_stream() if (distributed.active() and hasattr(model, "comm_stream")) else None
    if comm is not None:
        c = _lib.TrainCfg.from_buffer_copy(c)          # (train_cfg results may be shared: the stream is this call's)
        c.grads_stream = comm.cuda_stream
    _lib.check(lib.rnerf_train_forward_backward(C.byref(m), C.byref(c), state.theta.data_ptr(), o.data_ptr(), v.data_ptr(), pixels.data_ptr(), _lib.ptr(env), B,
                                                keys.data_ptr(), _lib.ptr(jit), _lib.ptr(u), per_ray, _lib.ptr(pd), _lib.ptr(dr), G.data_ptr(), ws.data_ptr(),
                                                int(model._mlp_wg_limit), C.byref(nxt) if nxt is not None else None, st), "rnerf_train_forward_backward")
    if next_path is not None:
        next_path.event.record(model._side)
    if comm is not None:
        n_big = state.segments["bkgd_mlp"][0]
        with torch.cuda.stream(comm):
            pending = distributed.allreduce_begin(G[:n_big])
        distributed.allreduce_mean_([G[n_big:]])
        distributed.allreduce_end_mean_(pending, G[:n_big])
    else:
        distributed.allreduce_mean_([G])
    _adam_update_on_device(state, flags)
    s8 = G[n_theta:]
    two = model.num_fine_samples > 0
    stats = Stats(loss=s8[0], psnr=s8[6], loss_c=s8[1], psnr_c=(s8[7] if two else 0.0), weight_l2=s8[4], loss_sp=0.0, loss_nrm=0.0,
                  annealing_rate=annealed, coarse_alpha_target=0.0, fine_alpha_target=0.0, loss_bg=s8[2], loss_bg_c=0.0,
                  loss_bg_smooth=s8[3])
    state.next_path = next_path
    return state, stats, rng


def _bump(t: torch.Tensor) -> None:
    """librnerf.so updated `t` in place behind PyTorch's back: bump its version counter so that caches keyed on it (the packed MFMA
    operand streams of the staged path, frozen_sq) see the change."""
    try:
        torch._C._autograd._unsafe_set_version_counter([t], [t._version + 1])
    except (AttributeError, TypeError):
        t.add_(0)


@contextlib.contextmanager
def range_safe(model: NerfModel, flags):
    """Inside: the model trains in the range-safe arithmetic — NerfMLP forward in bf16x3 (fp32's exponent range, 16-bit products), its bf16 hi
    plane saved as it is, backward "bf16" — with operand-stream caches of its own; everything else (march, compositing, background MLP, loss,
    optimiser) is fp32 as always."""
    saved = (model.precision, model._packed, getattr(flags, "backward_precision", "f16x3"))
    model.precision, model._packed, flags.backward_precision = _lib.PREC_BF16X3, {}, "bf16"
    try:
        yield model
    finally:
        model.precision, model._packed, flags.backward_precision = saved


_F16_BASED = (_lib.PREC_F16X3, _lib.PREC_F16)


def train_step(model: NerfModel, rng, state: TrainState, batch: Dict[str, Any], flags=None, *, range_retry=None, **kw):
    """One optimisation step (train.py:58-183): see _train_step_once for the arguments.

    Range of the f16-based training arithmetic.  A sample whose hidden activations leave f16's range (|x| > 65504; weights >= 256) comes
    out of the training forward as NaN, never as a plausible value, and so do the gradients; rnerf_adam_update counts the non-finite entries
    and — always — SKIPS such an update (theta, mu, nu untouched; state.nonfinite_grads() tells).  The reference's fp32 step is finite on
    such a batch.  range_retry=True (or flags.range_retry) makes this one too: the count is read after the step (one host synchronisation
    per step: the price, and why it is opt-in) and a skipped step is run again on the same batch and keys in the range-safe arithmetic
    (range_safe: bf16x3 forward, bf16 backward), whose update is applied; state.range_retries counts them.  Radiance stages only (stage
    all*'s input gradients are built on the row-normalised f16 backward modes: skipped and counted there, not re-run).

    range_retry="lag" (the default of utils.default_flags): the same without the per-step read.  The count of step k - 2 is read AFTER step k
    has been queued (a pinned host word written behind that step's update; the device never idles, the host still runs two steps ahead,
    every rank looks at the same lag), and a skipped batch is re-run then, behind step k — on the parameters step k left, which is what
    costs nothing: in the rare re-run the batch moves two places down the order.
    The Stats returned for the skipped step were non-finite; the re-run's are in state.last_retry_stats.  Call
    flush_range_retry(model, state) before reading the parameters for good (checkpoint, end of training): it settles the last step."""
    flags = state.flags if flags is None else flags
    mode = getattr(flags, "range_retry", False) if range_retry is None else range_retry
    retry = bool(mode)
    if not (retry and flags.stage.startswith("radiance") and model.precision in _F16_BASED):
        return _train_step_once(model, rng, state, batch, flags, **kw)
    if mode == "lag":
        if kw.get("taps") is not None or kw.get("forward_taps") is not None or torch.cuda.is_current_stream_capturing():
            # a tapped step is looked at now, not re-run later (range_retry=True does both); inside a stream capture there is no host to decide
            return _train_step_once(model, rng, state, batch, flags, **kw)
        ctx = _dist_context()
        out = _train_step_once(model, rng, state, batch, flags, **kw)
        slot = state._lag_count % _LAG_SLOTS
        state._lag_count += 1
        state._lag_host[slot:slot + 1].copy_(state.adam_scratch[3:4], non_blocking=True)      # behind this step's update, on its stream
        ev = torch.cuda.Event(); ev.record()
        replay = dict(kw, next_rays=None, path=None)                 # (the marched path of that step may be overwritten by then: marched again)
        # the re-run happens two steps later: it must see THIS step's batch even when the caller's loader refills its staging tensors in
        # place (147 KB for 4096 rays: three small device copies per step)
        state._lag_pending.append((ev, slot, rng, _clone_batch(batch), flags, replay, ctx))
        state._model_ref = weakref.ref(model)
        while len(state._lag_pending) > _RANGE_RETRY_LAG:
            pend = state._lag_pending.pop(0)
            if _lag_flag(state, pend):
                # the Stats of THIS step are views into state.grads' tail, which the re-run is about to overwrite: hand out copies
                out = (out[0], _clone_stats(out[1]), out[2])
            _settle_lagged(model, state, pend)
        return out
    step0 = state.step
    out = _train_step_once(model, rng, state, batch, flags, _guard=True, **kw)
    if state.nonfinite_grads() == 0:
        return out
    # the update was skipped: same batch, same keys, range-safe arithmetic.  The next batch's march the first attempt started stays.
    next_path = state.next_path
    state.step = step0
    state.sync_step_counter()
    kw = dict(kw, next_rays=None)
    with range_safe(model, flags):
        out = _train_step_once(model, rng, state, batch, flags, _guard=True, **kw)
    state.next_path = next_path
    _count_rerun(state)
    return out


def _dist_context():
    """What decides which ranks a step's collectives run over (distributed.use_group) and whether they run at all (skip_allreduce): kept
    with a lagged step, so that its re-run — two steps later, possibly after the caller left that context — issues exactly the collectives
    the first attempt did, on the same group (a re-run on another group would leave one rank alone in an all-reduce: a hang)."""
    return (distributed._GROUP, bool(distributed._skip()))


@contextlib.contextmanager
def _in_dist_context(ctx):
    saved = (distributed._GROUP, distributed._SKIP)
    distributed._GROUP, distributed._SKIP = ctx
    try:
        yield
    finally:
        distributed._GROUP, distributed._SKIP = saved


def _clone_batch(batch: Dict[str, Any]) -> Dict[str, Any]:
    """A private copy of the batch's tensors (rays, pixels, env-map directions: ~350 KB at 4096 rays) in ONE device launch: every float32
    tensor gets a 256-byte aligned slice of one staging buffer, filled by a single multi-tensor copy (four separate clones were four
    serial 6 us copies behind every step — 4 % of a 128-ray step, rocprof timeline of round 6)."""
    srcs = []

    def walk(v):
        if isinstance(v, torch.Tensor):
            if v.is_cuda and v.dtype == torch.float32:
                srcs.append(v)
                return ("t", len(srcs) - 1)
            return ("c", v.clone())
        if isinstance(v, tuple) and hasattr(v, "_fields"):            # Rays
            return ("n", type(v), [walk(x) for x in v])
        return ("c", v)

    plan = {k: walk(v) for k, v in batch.items()}
    views = []
    if srcs:
        offs, total = [], 0
        for t in srcs:
            offs.append(total)
            total += (t.numel() + 63) // 64 * 64
        flat = torch.empty(total, dtype=torch.float32, device=srcs[0].device)
        views = [flat[o:o + t.numel()].view(t.shape) for o, t in zip(offs, srcs)]
        torch._foreach_copy_(views, [t if t.is_contiguous() else t.contiguous() for t in srcs])

    def build(node):
        if node[0] == "t":
            return views[node[1]]
        if node[0] == "n":
            return node[1](*[build(x) for x in node[2]])
        return node[1]
    return {k: build(v) for k, v in plan.items()}


def _clone_stats(stats: Stats) -> Stats:
    import dataclasses
    return dataclasses.replace(stats, **{f.name: getattr(stats, f.name).clone() for f in dataclasses.fields(stats)
                                         if isinstance(getattr(stats, f.name), torch.Tensor)})


def _count_rerun(state: TrainState) -> None:
    """A re-run in the range-safe arithmetic has been applied — or skipped as well when its gradient is non-finite too (a NaN pixel, a loss
    that overflowed fp32): counted apart and said aloud, never booked as a success.  (One host read, on the rare path only.)
    Note for both cases: a skipped update still advances the update count — Adam's bias-correction t and the learning-rate schedule move
    on, the moments do not: deliberately unlike optax.apply_if_finite, which the reference does not use either (its step would have
    written NaN parameters)."""
    state.range_retries += 1
    if state.nonfinite_grads() != 0:
        state.range_retry_failures += 1
        import warnings
        warnings.warn("train_step: the range-safe re-run of a skipped batch has a non-finite gradient too; its update stays skipped "
                      f"({state.range_retry_failures} so far)", RuntimeWarning, stacklevel=3)


def _lag_flag(state: TrainState, pending) -> bool:
    """True when the lagged step's update was skipped (its non-finite count, written to pinned host memory behind the update)."""
    pending[0].synchronize()                                         # that step has long finished: the device is busy with the one queued after it
    return float(state._lag_host[pending[1]]) != 0.0


def _settle_lagged(model: NerfModel, state: TrainState, pending) -> None:
    ev, slot, rng, batch, flags, replay, ctx = pending
    if not _lag_flag(state, pending):
        return
    next_path = state.next_path
    state.step -= 1                                                  # the skipped update had counted as a step: the re-run takes its place
    state.sync_step_counter()
    with range_safe(model, flags), _in_dist_context(ctx):
        _, stats, _ = _train_step_once(model, rng, state, batch, flags, **replay)
    state.next_path = next_path
    state.last_retry_stats = _clone_stats(stats)
    _count_rerun(state)


def flush_range_retry(model: NerfModel, state: TrainState) -> None:
    """range_retry="lag": settle the last queued step (read its count; re-run it if its update was skipped)."""
    while state._lag_pending:
        _settle_lagged(model, state, state._lag_pending.pop(0))


def _train_step_once(model: NerfModel, rng, state: TrainState, batch: Dict[str, Any], flags=None, *, jitter=None, u_fine=None,
                     taps: Optional[dict] = None, path=None, next_rays: Optional[Rays] = None, forward_taps: Optional[dict] = None,
                     noise_c=None, noise_f=None, _guard: bool = False):
    """One optimisation step (train.py:58-183).  batch: {"rays": Rays of [B,3], "pixels": [B,>=3], "annealed_alpha": float,
    "env_rays": Rays with viewdirs [ps,ps,3] (when bg_smooth_weight > 0)}.  path: an optional NerfModel.prefetch_path handle for
    these rays (the march carries no gradient and does not read the trained parameters, so it may overlap the previous step);
    next_rays: the rays of the NEXT step — their march is issued on the model's side stream after this step's wgrad, so that it
    runs beside the small kernels of the step's tail; the handle is left in state.next_path.  Returns (state, stats, rng); the Stats fields are
    0-dim device tensors (no host synchronisation inside the step)."""
    flags = state.flags if flags is None else flags           # the reference reads the global FLAGS (train.py:52)
    all_stage = flags.stage.startswith("all")
    if flags.stage.startswith("ior"):
        return _ior_stage_step(model, rng, state, batch, flags)
    if not (flags.stage.startswith("radiance") or all_stage):
        raise NotImplementedError(f"train_step: unknown stage {flags.stage!r}")
    if all_stage and model.stage != flags.stage:
        raise ValueError("stage all*: build the model with the same stage (the march must evaluate so3_mlp)")
    if flags.beta_weight > 0 or flags.sparsity_weight > 0:
        pass        # both are multiplied by annealing_rate = 0.0 (train.py:156): no contribution to loss or gradient
    noisy = getattr(model, "noise_std", None) is not None and bool(flags.randomized)      # the raw-sigma regulariser: drawn on the host, staged path
    if not all_stage and taps is None and forward_taps is None and getattr(model, "whole_path", False) and not noisy:
        # the product path: the whole forward + backward is ONE call into librnerf.so (rnerf_train_forward_backward), the update another
        # (rnerf_adam_update); what follows below is the same sequence stage by stage, with taps (parity tests, stage all*)
        return _train_step_whole(model, rng, state, batch, flags, jitter, u_fine, path, next_rays)
    rng, key_0, key_1 = prng.split(np.asarray(rng, np.uint32), 3)
    annealed = float(np.asarray(batch["annealed_alpha"]).reshape(-1)[0])
    rays: Rays = batch["rays"]
    pixels = batch["pixels"][..., :3].contiguous()
    variables = state.variables
    prec = model.precision
    if all_stage and next_rays is not None:
        # the NEXT batch's march needs this step's so3 update and cannot start early, but the ray order it marches in (a coarse pre-march
        # without so3 + a sort, ~0.3 ms) depends on the rays and the grid only: computed on the side stream beside this step
        if model._side is None:
            from .models import shared_stream
            model._side = shared_stream(model.device, "march")
        ops.prefetch_shell_order(model.table, model.spec, next_rays.origins, next_rays.viewdirs, model.near, model.far, model.num_samples, model._side)
        next_rays = None                  # (no path prefetch in this stage)
    Nc, Nf = model.num_coarse_samples, model.num_fine_samples
    bwd = backward_mode(flags, model)
    if all_stage and bwd == _lib.BWD_BF16:
        raise ValueError('stage all*: the input gradients are built on the row-normalised backward modes (backward_precision "f16x3" or "f16")')
    ctx: Dict[str, Any] = {"backward": bwd, "loss_sp": taps is not None}      # the sparsity term's value: only when somebody will look at it
    if flags.bg_smooth_weight > 0:
        ev = batch["env_rays"].viewdirs
        ctx["env_dirs"] = ev.reshape(-1, 3)
    hold = {}
    if _MARCH_RESERVE > 0 and next_rays is not None and not all_stage:
        # the NEXT step's march, issued now: it runs beside this step's training forward, which leaves it _MARCH_RESERVE CUs (a dependent
        # gather chain: 64 waves, as fast on a few CUs as on many)
        hold["path"] = model.prefetch_path(next_rays, sync_inputs=True, reserve_cus=_MARCH_RESERVE)
    ret, _loss_sp = model.apply(variables, key_0, key_1, rays, flags.randomized, annealed, jitter=jitter, u_fine=u_fine, ctx=ctx, path=path,
                             taps=forward_taps, noise_c=noise_c, noise_f=noise_f)
    if "path" in hold:
        model.release_reserved_cus()
    B = ctx["B"]
    rgb_f, _, _, trans_f, tb_f = ret[-1]
    rgb_c = ret[0][0] if len(ret) > 1 else None
    sums = ops.loss_reduce(rgb_c, rgb_f, trans_f, tb_f, pixels)
    bg_on = 1.0 if (flags.bg_weight > 0 and annealed > 0) else 0.0
    mse_scale = 2.0 / (3.0 * B)

    G = state.grads
    G.zero_()
    # ---- backward: last level first ---------------------------------------------------------------------------------------
    M_env = ctx["rgb_env"].shape[0] if flags.bg_smooth_weight > 0 else 0
    d_all = torch.empty((B + M_env, 3), dtype=torch.float32, device=pixels.device)     # rows [0,B): rays' bkgd, [B,B+M): env-map patch
    d_first = d_all[:B]
    if Nf > 0:
        d_raw_f, d_bkgd = ops.composite_backward(ctx["raw_f"], ctx["rows_pd"], ctx["rows_dr"], None, Nc + Nf, B, ctx["bkgd"], rgb_f, pixels,
                                                 trans_f, tb_f, sums, mse_scale, flags.bg_weight * bg_on, rgb_padding=model.rgb_padding,
                                                 sigma_bias=model.sigma_bias, bd_cut_bbox=ctx.get("bd_cut_bbox"), white_bkgd=model.white_bkgd,
                                                 d_bkgd=d_first, accumulate_bkgd=False, mask_bbox=ctx.get("mask_bbox"))
        ops.nerfmlp_backward(_bwd_packed(model, state, "fine_mlp", bwd), model._packed_weights(variables, "fine_mlp"), prec, ctx["save_f"],
                             d_raw_f, (Nc + Nf) * B, grads=state.grad_view("fine_mlp"), backward=bwd)
        d_raw_c, d_bkgd = ops.composite_backward(ctx["raw_c"], ctx["path_pd"], ctx["path_dr"], ctx["jit"], Nc, B, ctx["bkgd"], rgb_c, pixels,
                                                 None, None, None, mse_scale, 0.0, d_bkgd=d_bkgd, rgb_padding=model.rgb_padding,
                                                 sigma_bias=model.sigma_bias, white_bkgd=model.white_bkgd, mask_bbox=ctx.get("mask_bbox"))
    else:
        d_raw_c, d_bkgd = ops.composite_backward(ctx["raw_c"], ctx["path_pd"], ctx["path_dr"], ctx["jit"], Nc, B, ctx["bkgd"], rgb_f, pixels,
                                                 trans_f, tb_f, sums, mse_scale, flags.bg_weight * bg_on, rgb_padding=model.rgb_padding,
                                                 sigma_bias=model.sigma_bias, white_bkgd=model.white_bkgd, d_bkgd=d_first, accumulate_bkgd=False,
                                                 mask_bbox=ctx.get("mask_bbox"))
    # The march of the NEXT step (it reads neither the trained parameters nor anything of this step) goes to the side stream after the
    # wgrad, beside the small kernels of the step's tail (background-MLP backward, loss glue, Adam).  Those are ~0.35 ms against 0.7-0.8 ms
    # of march, so ~0.4 ms of every step still waits for it (rocprof timeline) — but issuing it between the dgrad and the
    # wgrad (RNERF_MARCH_BEFORE_WGRAD=1), where it would be hidden entirely, costs the HBM-paced wgrad more than that: 6.81 -> 7.18 ms.
    def issue_next_march():
        hold["path"] = model.prefetch_path(next_rays, sync_inputs=True, reserve_cus=0) if next_rays is not None else None
    early = _MARCH_EARLY and next_rays is not None and "path" not in hold
    chain_beside = all_stage and _ALL_CHAIN_BESIDE_WGRAD and hasattr(model, "tail_stream")
    if chain_beside:
        # stage all*: only the dgrad here; the (HBM-paced) wgrad is issued further down, beside the adjoint chain that needs dY but not dW
        dy_c = ops.nerfmlp_backward(_bwd_packed(model, state, "coarse_mlp", bwd), model._packed_weights(variables, "coarse_mlp"), prec, ctx["save_c"],
                                    d_raw_c, Nc * B, backward=bwd, stages="d")
    else:
        _, dy_c = ops.nerfmlp_backward(_bwd_packed(model, state, "coarse_mlp", bwd), model._packed_weights(variables, "coarse_mlp"), prec, ctx["save_c"],
                                       d_raw_c, Nc * B, grads=state.grad_view("coarse_mlp"), backward=bwd, return_dy=True,
                                       between=issue_next_march if early else None)
    # jax.lax.pmean of the gradients (train.py:166), first part: the NerfMLP segments are final here, their all-reduce (95 % of the
    # bytes) starts now and runs beside the rest of the step; the background-MLP gradients and the stats follow in a small second one
    n_big = state.segments["bkgd_mlp"][0]
    pending = None if chain_beside else distributed.allreduce_begin(G[:n_big])
    if "path" not in hold:
        issue_next_march()
    next_path = hold["path"]
    bk_flat = variables["flat"]["bkgd_mlp"]
    g_bk = state.grad_view("bkgd_mlp")
    # ---- env-map smoothness (train.py:127-132): its rows went through the background MLP together with the rays' rows
    ps, on, env_sum = 0, 0.0, None
    if flags.bg_smooth_weight > 0:
        ps = batch["env_rays"].viewdirs.shape[0]
        on = 1.0 if annealed > 0 else 0.0
        env_sum = torch.empty(_lib.load().rnerf_env_smooth_sum_floats(int(ps)), dtype=torch.float32, device=pixels.device)
        ops.env_smooth_backward(ctx["rgb_env"], ps, flags.bg_smooth_weight * on, d_all[B:], env_sum)
    if all_stage and chain_beside:
        _, d_bk_dirs = ops.bkgd_backward(bk_flat, ctx["save_bkgd"], d_all, g_bk, model.rgb_padding, want_d_dirs=True)
        # Two independent consumers of dY from here on: the NerfMLP weight gradient (one big kernel that streams 10.6 GB and leaves the matrix
        # pipe half idle) and the adjoint chain of the march (input gradients, so3 forward / Jacobians, the reverse scan, so3 backward: ~3.9 ms
        # of small fp32-MFMA / latency-bound kernels that stream little).  They cannot share a CU (registers), but they can share
        # the CHIP: the wgrad goes to this stream, the chain to the side stream, and the dispatcher interleaves their workgroups CU by CU.
        # The wgrad is queued first: the chain's one host synchronisation (the pair count) then idles nothing.
        main, side = torch.cuda.current_stream(), model.tail_stream()
        side.wait_stream(main)
        ops.nerfmlp_backward(_bwd_packed(model, state, "coarse_mlp", bwd), model._packed_weights(variables, "coarse_mlp"), prec, ctx["save_c"],
                             d_raw_c, Nc * B, grads=state.grad_view("coarse_mlp"), backward=bwd, dy=dy_c, stages="w")
        pending = distributed.allreduce_begin(G[:n_big])
        with torch.cuda.stream(side):
            _all_stage_backward(model, state, variables, ctx, dy_c, d_bk_dirs, bwd, annealed, taps)
        main.wait_stream(side)
    elif all_stage:
        _, d_bk_dirs = ops.bkgd_backward(bk_flat, ctx["save_bkgd"], d_all, g_bk, model.rgb_padding, want_d_dirs=True)
        _all_stage_backward(model, state, variables, ctx, dy_c, d_bk_dirs, bwd, annealed, taps)
    else:
        ops.bkgd_backward(bk_flat, ctx["save_bkgd"], d_all, g_bk, model.rgb_padding)
    # ---- weight_l2 over ALL variables, the frozen path_sampler included (train.py:147-153), and the Stats scalars: they ride in the
    #      tail of the gradient buffer, one all-reduce for both (train.py:166-167)
    n_theta = state.theta.numel()
    frozen_sq_of(state, variables)
    n_all = n_theta + state.frozen_sq[1]
    st = G[n_theta:]
    ops.train_stats(sums, B, rgb_c is not None, flags.bg_weight * bg_on, env_sum, ps, on, state.theta, state.frozen_sq[0], n_all, st)
    distributed.allreduce_mean_([G[n_big:]])
    distributed.allreduce_end_mean_(pending, G[:n_big])
    grads = G[:n_theta]
    if taps is None and state.adam_scratch is not None:
        # nobody looks at the gradient: the product form of the update — one C call with the non-finite guard — instead of the tensor
        # arithmetic below (stage all*, the noise_std steps: the sequences the whole-path entry does not cover)
        _adam_update_on_device(state, flags)
        annealing_rate = 0.0
        stats = Stats(loss=st[0], psnr=st[6], loss_c=st[1], psnr_c=(st[7] if rgb_c is not None else 0.0),
                      weight_l2=st[4], loss_sp=flags.sparsity_weight * annealing_rate * _loss_sp, loss_nrm=0.0, annealing_rate=annealed, coarse_alpha_target=0.0,
                      fine_alpha_target=0.0, loss_bg=st[2], loss_bg_c=0.0, loss_bg_smooth=st[3])
        state.next_path = next_path
        return state, stats, rng
    if flags.weight_decay_mult > 0:      # d (weight_decay_mult * weight_l2) / d theta: identical on every rank, so it is added after the mean
        grads.add_(state.theta, alpha=2.0 * flags.weight_decay_mult / n_all)
    if flags.grad_max_val > 0:                                                            # train.py:169-172
        grads.clamp_(-flags.grad_max_val, flags.grad_max_val)
    if flags.grad_max_norm > 0:                                                           # train.py:174-180
        # the reference's norm runs over the WHOLE gradient tree: the frozen path_sampler's entries are 2 wd theta / n_all (from
        # weight_l2; its optimiser label is "zero" but jax.grad still returns them), value-clipped like the rest
        sq = (grads * grads).sum()
        if flags.weight_decay_mult > 0 and state.frozen_sq[1] > 0:
            fg = variables["flat"]["so3_mlp"] * (2.0 * flags.weight_decay_mult / n_all)
            if flags.grad_max_val > 0:
                fg = fg.clamp(-flags.grad_max_val, flags.grad_max_val)
            sq = sq + (fg * fg).sum()
        norm = torch.sqrt(sq)
        grads.mul_(torch.clamp(flags.grad_max_norm / (1e-7 + norm), max=1.0))
    if taps is not None:
        taps.update(grads=grads.clone(), sums=sums, ctx=ctx, loss_sp=_loss_sp)
    state._staged_bad = None
    if _guard:                             # range_retry on the staged sequence: count on the host (it synchronises anyway), skip like rnerf_adam_update
        state._staged_bad = int((~torch.isfinite(grads)).sum().item())
    if state._staged_bad:
        state.step += 1
    else:
        state.apply_gradients(grads)
    # train.py:153-160: Stats.loss_sp = sparsity_weight * annealing_rate * loss_sp with annealing_rate = 0.0 — the reference's number is 0.0
    # whatever the term's value (models.py:351-357 keeps it finite: safe_log, a denominator >= 1); the value itself is in taps["loss_sp"]
    annealing_rate = 0.0
    stats = Stats(loss=st[0], psnr=st[6], loss_c=st[1], psnr_c=(st[7] if rgb_c is not None else 0.0),
                  weight_l2=st[4], loss_sp=flags.sparsity_weight * annealing_rate * _loss_sp, loss_nrm=0.0, annealing_rate=annealed, coarse_alpha_target=0.0, fine_alpha_target=0.0,
                  loss_bg=st[2], loss_bg_c=0.0, loss_bg_smooth=st[3])
    state.next_path = next_path
    return state, stats, rng
