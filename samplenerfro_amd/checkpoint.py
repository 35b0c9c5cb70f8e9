"""flax checkpoint import/export for the MLP weights (SURVEY.md §5 "Checkpoint / resume", §8f N1).

`flax.training.checkpoints` writes `checkpoint_<step>` files = msgpack of the state dict, with numpy arrays encoded as
msgpack ExtType(1, msgpack.packb((shape, dtype_name, raw_bytes))) (flax/serialization.py: _ndarray_to_bytes /
_msgpack_ext_pack; ext code 2 = numpy scalars, 3 = native complex).  The reference's TrainState is
{"optimizer": {"target": {"params": {...}}, "state": ...}} for flax.optim (utils.py:40-43) — eval.py:124-152 grafts
`coarse_mlp`, `fine_mlp`, `bkgd_mlp` and `path_sampler` sub-trees.  This module needs only `msgpack` (no flax).
Written from the published format; no reference checkpoint is available offline, so it is tested by round trip only.
"""
from __future__ import annotations

from typing import Any, Dict

import msgpack
import numpy as np
import torch

from .models import BKGD_MLP_SHAPES, NERF_MLP_SHAPES, SO3_MLP_SHAPES, make_variables, tree_to_flat


def _ext_hook(code, data):
    if code == 1:
        shape, dtype_name, buf = msgpack.unpackb(data, raw=False)
        return np.frombuffer(buf, dtype=np.dtype(dtype_name)).reshape(shape)
    if code == 2:
        dtype_name, buf = msgpack.unpackb(data, raw=False)
        return np.frombuffer(buf, dtype=np.dtype(dtype_name))[0]
    return msgpack.ExtType(code, data)


def _default(obj):
    if isinstance(obj, np.ndarray):
        return msgpack.ExtType(1, msgpack.packb((list(obj.shape), obj.dtype.name, obj.tobytes()), use_bin_type=True))
    if isinstance(obj, np.generic):
        return msgpack.ExtType(2, msgpack.packb((obj.dtype.name, obj.tobytes()), use_bin_type=True))
    raise TypeError(type(obj))


def load_state_dict(path: str) -> Dict[str, Any]:
    with open(path, "rb") as f:
        return msgpack.unpackb(f.read(), ext_hook=_ext_hook, raw=False, strict_map_key=False)


def save_state_dict(path: str, state: Dict[str, Any]) -> None:
    with open(path, "wb") as f:
        f.write(msgpack.packb(state, default=_default, use_bin_type=True))


def find_params(state: Dict[str, Any]) -> Dict[str, Any]:
    """Locate the {"coarse_mlp": ..., ...} tree inside a flax TrainState dict (flax.optim or optax layouts)."""
    cur = state
    for path in (("optimizer", "target", "params"), ("params", "params"), ("params",), ()):
        node, ok = cur, True
        for k in path:
            if isinstance(node, dict) and k in node:
                node = node[k]
            else:
                ok = False
                break
        if ok and isinstance(node, dict) and "coarse_mlp" in node:
            return node
    raise KeyError("no params tree with a 'coarse_mlp' entry found in the checkpoint")


def variables_from_checkpoint(path: str, device) -> Dict[str, Any]:
    """checkpoint_<step> -> variables usable by NerfModel.apply (flat fp32 buffers + flax-shaped views)."""
    p = find_params(load_state_dict(path))
    flat = {"coarse_mlp": tree_to_flat(p["coarse_mlp"], NERF_MLP_SHAPES, device),
            "bkgd_mlp": tree_to_flat(p["bkgd_mlp"], BKGD_MLP_SHAPES, device)}
    if "fine_mlp" in p:
        flat["fine_mlp"] = tree_to_flat(p["fine_mlp"], NERF_MLP_SHAPES, device)
    so3 = p.get("path_sampler", {}).get("scan", {}).get("idx_model", {}).get("so3_mlp")
    if so3 is not None:
        flat["so3_mlp"] = tree_to_flat(so3, SO3_MLP_SHAPES, device)
    return make_variables(flat)


def params_to_state_dict(variables: Dict[str, Any], step: int = 0) -> Dict[str, Any]:
    """variables -> a flax.optim-style TrainState dict holding numpy arrays (for save_state_dict)."""
    def conv(t):
        if isinstance(t, dict):
            return {k: conv(v) for k, v in t.items()}
        return t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
    return {"optimizer": {"target": {"params": conv(variables["params"])}, "state": {"step": np.int32(step)}}}
