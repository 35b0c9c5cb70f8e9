"""flax checkpoint import/export for the path's weights (SURVEY.md §5 "Checkpoint / resume", §8f N1).

`flax.training.checkpoints.save_checkpoint` writes `checkpoint_<step>` = `flax.serialization.msgpack_serialize(to_state_dict(state))`
where `state` is the reference's `flax.training.train_state.TrainState` (train.py:32,317):

    {"step": int, "params": {"params": {"coarse_mlp": {"Dense_k": {"kernel", "bias"}}, "fine_mlp", "bkgd_mlp",
                                        "path_sampler": {"scan": {"idx_model": {"so3_mlp": ...}}}}},
     "opt_state": {"inner_states": {label: {"inner_state": {"0": {"count", "mu", "nu"}, "1": {"count"}}}}}}      # optax.multi_transform

Encoding (flax/serialization.py `_MsgpackExtType`): numpy arrays are msgpack ExtType(1, packb((shape, dtype.name, bytes))), native
complex ExtType(2, packb((re, im))), numpy scalars ExtType(3, <the same triple as an ndarray>); tuples / NamedTuples become dicts keyed
"0", "1", ... / by field name; optax's MaskedNode and EmptyState become {}.  eval.py:124-152 and extract_mesh.py read
`pretrain["step"]` and `pretrain["params"]["params"][<sub-tree>]` and graft them into fresh variables by the gin names
`Config.radiance_weight_name / ior_weight_name / all_weight_name` (rnerf/utils.py:81-84) — `graft_pretrained` below.

Needs only `msgpack` (no flax).  No reference checkpoint is available offline: the reader is tested against a byte fixture assembled
independently of this module (tests/golden/make_flax_ckpt.py), the writer by reading its output back the way eval.py indexes it.
"""
from __future__ import annotations

import os
import re
from typing import Any, Dict, Optional

import msgpack
import numpy as np
import torch

from .models import BKGD_MLP_SHAPES, NERF_MLP_SHAPES, SO3_MLP_SHAPES, make_variables, tree_to_flat

_EXT_NDARRAY, _EXT_COMPLEX, _EXT_NPSCALAR = 1, 2, 3          # flax.serialization._MsgpackExtType


def _ndarray_from_bytes(data):
    shape, dtype_name, buf = msgpack.unpackb(data, raw=False)
    return np.frombuffer(buf, dtype=np.dtype(dtype_name)).reshape(shape)


def _ext_hook(code, data):
    if code == _EXT_NDARRAY:
        return _ndarray_from_bytes(data)
    if code == _EXT_NPSCALAR:
        return _ndarray_from_bytes(data)[()]
    if code == _EXT_COMPLEX:
        re_, im = msgpack.unpackb(data, raw=False)
        return complex(re_, im)
    return msgpack.ExtType(code, data)


def _ndarray_to_bytes(a: np.ndarray) -> bytes:
    return msgpack.packb((list(a.shape), a.dtype.name, a.tobytes("C")), use_bin_type=True)


def _default(obj):
    if isinstance(obj, np.ndarray):
        return msgpack.ExtType(_EXT_NDARRAY, _ndarray_to_bytes(obj))
    if isinstance(obj, np.generic):
        return msgpack.ExtType(_EXT_NPSCALAR, _ndarray_to_bytes(np.asarray(obj)))
    if isinstance(obj, complex):
        return msgpack.ExtType(_EXT_COMPLEX, msgpack.packb((obj.real, obj.imag)))
    raise TypeError(type(obj))


def load_state_dict(path: str) -> Dict[str, Any]:
    with open(path, "rb") as f:
        return msgpack.unpackb(f.read(), ext_hook=_ext_hook, raw=False, strict_map_key=False)


def save_state_dict(path: str, state: Dict[str, Any]) -> None:
    with open(path, "wb") as f:
        f.write(msgpack.packb(state, default=_default, use_bin_type=True))


def latest_checkpoint(ckpt_dir: str, prefix: str = "checkpoint_") -> Optional[str]:
    """flax.training.checkpoints.latest_checkpoint: the `prefix<step>` file with the largest step (natural sort), or None."""
    if not os.path.isdir(ckpt_dir):
        return None
    best = None
    for name in os.listdir(ckpt_dir):
        m = re.fullmatch(re.escape(prefix) + r"(\d+(?:\.\d+)?)", name)
        if m and (best is None or float(m.group(1)) > best[0]):
            best = (float(m.group(1)), name)
    return os.path.join(ckpt_dir, best[1]) if best else None


def restore_checkpoint(ckpt_dir_or_file: str) -> Dict[str, Any]:
    """checkpoints.restore_checkpoint(path, target=None): the raw state dict of the latest checkpoint in a directory (or of a file)."""
    p = ckpt_dir_or_file if os.path.isfile(ckpt_dir_or_file) else latest_checkpoint(ckpt_dir_or_file)
    if p is None:
        raise FileNotFoundError(f"no checkpoint_<step> file in {ckpt_dir_or_file}")
    return load_state_dict(p)


def find_params(state: Dict[str, Any]) -> Dict[str, Any]:
    """The {"coarse_mlp": ..., ...} tree inside a state dict: TrainState layout first (what the reference writes), then the legacy
    flax.optim layout and a bare params tree."""
    for path in (("params", "params"), ("optimizer", "target", "params"), ("params",), ()):
        node, ok = state, True
        for k in path:
            if isinstance(node, dict) and k in node:
                node = node[k]
            else:
                ok = False
                break
        if ok and isinstance(node, dict) and ("coarse_mlp" in node or "path_sampler" in node):
            return node
    raise KeyError("no params tree with a 'coarse_mlp' / 'path_sampler' entry found in the checkpoint")


def _so3_tree(p):
    return p.get("path_sampler", {}).get("scan", {}).get("idx_model", {}).get("so3_mlp")


def flat_from_params(p: Dict[str, Any], device, names=("coarse_mlp", "bkgd_mlp", "fine_mlp", "so3_mlp")) -> Dict[str, torch.Tensor]:
    flat = {}
    if "coarse_mlp" in p and "coarse_mlp" in names:
        flat["coarse_mlp"] = tree_to_flat(p["coarse_mlp"], NERF_MLP_SHAPES, device)
    if "bkgd_mlp" in p and "bkgd_mlp" in names:
        flat["bkgd_mlp"] = tree_to_flat(p["bkgd_mlp"], BKGD_MLP_SHAPES, device)
    if "fine_mlp" in p and "fine_mlp" in names:
        flat["fine_mlp"] = tree_to_flat(p["fine_mlp"], NERF_MLP_SHAPES, device)
    so3 = _so3_tree(p)
    if so3 is not None and "so3_mlp" in names:
        flat["so3_mlp"] = tree_to_flat(so3, SO3_MLP_SHAPES, device)
    return flat


def variables_from_checkpoint(path: str, device) -> Dict[str, Any]:
    """checkpoint_<step> (file or directory) -> variables usable by NerfModel.apply (flat fp32 buffers + flax-shaped views)."""
    return make_variables(flat_from_params(find_params(restore_checkpoint(path)), device))


def graft_pretrained(variables: Dict[str, Any], train_dir: str, stage: str, num_fine_samples: int, radiance_weight_name="radiance",
                     ior_weight_name="ior", all_weight_name="all"):
    """eval.py:124-152: overwrite sub-trees of freshly initialised `variables` with the latest checkpoint of the stage directories named
    by gin's Config.*_weight_name.  Returns (variables, step)."""
    dev = next(iter(variables["flat"].values())).device
    flat = dict(variables["flat"])

    def take(ckpt, names):
        got = flat_from_params(ckpt["params"]["params"], dev)           # indexed exactly like eval.py:128-131
        for n in names:
            if n == "fine_mlp" and num_fine_samples <= 0:
                continue
            flat[n] = got[n]

    if stage.startswith("radiance") or stage.startswith("ior"):
        pre = restore_checkpoint(os.path.join(train_dir, radiance_weight_name))
        step = int(pre["step"])
        take(pre, ["bkgd_mlp", "coarse_mlp", "fine_mlp"])
        if stage.startswith("ior"):
            pre = restore_checkpoint(os.path.join(train_dir, ior_weight_name))
            step = int(pre["step"])
            take(pre, ["so3_mlp"])
    elif stage.startswith("all"):
        pre = restore_checkpoint(os.path.join(train_dir, all_weight_name))
        step = int(pre["step"])
        take(pre, ["bkgd_mlp", "coarse_mlp", "fine_mlp", "so3_mlp"])
    else:
        raise ValueError(f"unknown stage {stage!r}")
    return make_variables(flat), step


# ---- export ---------------------------------------------------------------------------------------------------------------------
def _np_tree(t):
    if isinstance(t, dict):
        return {k: _np_tree(v) for k, v in t.items()}
    return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)


def _masked_like(tree, keep: bool):
    """optax.masked: leaves outside the group are MaskedNode() -> {} in a state dict."""
    if isinstance(tree, dict):
        return {k: _masked_like(v, keep) for k, v in tree.items()}
    return tree if keep else {}


def params_to_state_dict(variables: Dict[str, Any], step: int = 0, train_state=None, stage: str = "radiance") -> Dict[str, Any]:
    """variables (+ optionally a samplenerfro_amd.train.TrainState for the Adam moments) -> the TrainState state dict the reference's
    `checkpoints.save_checkpoint(stage_dir, state, step)` writes (train.py:424-427), readable by its restore_checkpoint / eval.py.

    opt_state follows optax.multi_transform over the four transforms of train.py:312-316 with the stage's labels (:286-310): the
    trained groups carry (ScaleByAdamState(count, mu, nu), ScaleByScheduleState(count)), the rest MaskedNode / EmptyState.  Without a
    train_state the moments are zeros and count = step (a weights-only export that still restores)."""
    if train_state is not None and getattr(train_state, "_lag_pending", None):
        train_state.state_dict()      # settles the steps range_retry="lag" still holds (re-runs a skipped batch), or raises: never a silent loss
    params = _np_tree(variables["params"])
    trained = {"radiance": {"coarse_mlp", "fine_mlp", "bkgd_mlp"}, "ior": {"path_sampler"},
               "all": {"coarse_mlp", "fine_mlp", "bkgd_mlp", "path_sampler"}}["radiance" if stage.startswith("radiance") else
                                                                              ("ior" if stage.startswith("ior") else "all")]
    moments = {"mu": {}, "nu": {}}
    for name, tree in params.items():
        for which in ("mu", "nu"):
            if name in trained:
                src = None
                # the params tree calls the IoR group "path_sampler" (rnerf/models.py:121-131); its flat segment is "so3_mlp"
                seg = "so3_mlp" if name == "path_sampler" else name
                if train_state is not None and seg in getattr(train_state, "segments", {}):
                    lo, hi = train_state.segments[seg]
                    from .models import flat_to_tree
                    shapes = {"bkgd_mlp": BKGD_MLP_SHAPES, "so3_mlp": SO3_MLP_SHAPES}.get(seg, NERF_MLP_SHAPES)
                    src = _np_tree(flat_to_tree(getattr(train_state, which)[lo:hi], shapes))
                    if name == "path_sampler":
                        src = {"scan": {"idx_model": {"so3_mlp": src}}}
                moments[which][name] = src if src is not None else _zeros_like(tree)
            else:
                moments[which][name] = _masked_like(tree, False)
    count = np.asarray(int(train_state.step) if train_state is not None else int(step), np.int32)
    adam_sched = {"0": {"count": count, "mu": {"params": moments["mu"]}, "nu": {"params": moments["nu"]}}, "1": {"count": count}}
    masked_all = {"params": {k: _masked_like(v, False) for k, v in params.items()}}
    adam_unused = {"0": {"count": np.asarray(0, np.int32), "mu": masked_all, "nu": masked_all}, "1": {}}
    inner = {"adam": {"inner_state": adam_unused}, "adam_lr_scheduler": {"inner_state": adam_sched},
             "adam_lr_scheduler1": {"inner_state": {"0": adam_unused["0"], "1": {"count": np.asarray(0, np.int32)}}},
             "zero": {"inner_state": {}}}
    return {"step": int(step), "params": {"params": params}, "opt_state": {"inner_states": inner}}


def _zeros_like(tree):
    if isinstance(tree, dict):
        return {k: _zeros_like(v) for k, v in tree.items()}
    return np.zeros_like(tree)


def save_checkpoint(ckpt_dir: str, variables: Dict[str, Any], step: int, train_state=None, stage: str = "radiance", prefix="checkpoint_") -> str:
    """checkpoints.save_checkpoint(ckpt_dir, state, step): writes `<ckpt_dir>/checkpoint_<step>`."""
    os.makedirs(ckpt_dir, exist_ok=True)
    path = os.path.join(ckpt_dir, f"{prefix}{int(step)}")
    save_state_dict(path, params_to_state_dict(variables, step, train_state, stage))
    return path
