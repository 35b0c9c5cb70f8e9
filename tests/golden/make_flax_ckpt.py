#!/usr/bin/env python3
"""Assembles tests/golden/flax_checkpoint_7.msgpack.gz: a `checkpoint_7` in the byte format of flax 0.3.6's
`flax.training.checkpoints.save_checkpoint` for the reference's TrainState (train.py:32,317; optax.multi_transform, :312-316),
written WITHOUT samplenerfro_amd.checkpoint (the code under test): msgpack maps with str keys, numpy arrays as
ExtType(1, packb((shape, dtype.name, tobytes))), numpy scalars as ExtType(3, <same triple>), MaskedNode / EmptyState as empty maps.

No real flax checkpoint exists offline; the layout restates flax/serialization.py and optax 0.1.0's state
NamedTuples.  Array values are a cheap arithmetic pattern (compresses to a few tens of KB).
"""
import gzip
import os

import msgpack
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
NERF = [(63, 256), (256, 256), (256, 256), (256, 256), (256, 256), (319, 256), (256, 256), (256, 256), (256, 1), (256, 256), (283, 128), (128, 3)]
SMALL = [(27, 128), (128, 128), (128, 128), (155, 128), (128, 3)]
SO3 = [(60, 128), (128, 128), (128, 128), (188, 128), (128, 3)]


def ext_array(a):
    return msgpack.ExtType(1, msgpack.packb((list(a.shape), a.dtype.name, a.tobytes("C")), use_bin_type=True))


def ext_scalar(a):
    a = np.asarray(a)
    return msgpack.ExtType(3, msgpack.packb((list(a.shape), a.dtype.name, a.tobytes("C")), use_bin_type=True))


def pattern(shape, salt):
    n = int(np.prod(shape))
    return (((np.arange(n, dtype=np.int64) * 7 + salt) % 13 - 6).astype(np.float32) / np.float32(64)).reshape(shape)


def mlp(shapes, salt, fn):
    return {f"Dense_{k}": {"kernel": fn((i, o), salt + 2 * k), "bias": fn((o,), salt + 2 * k + 1)} for k, (i, o) in enumerate(shapes)}


def tree(fn):
    return {"coarse_mlp": mlp(NERF, 100, fn), "fine_mlp": mlp(NERF, 200, fn), "bkgd_mlp": mlp(SMALL, 300, fn),
            "path_sampler": {"scan": {"idx_model": {"so3_mlp": mlp(SO3, 400, fn)}}}}


def build(step=7):
    arr = lambda shape, salt: ext_array(pattern(shape, salt))
    mom = lambda scale: (lambda shape, salt: ext_array(pattern(shape, salt + 1000) * np.float32(scale)))
    masked = lambda shape, salt: {}
    trained = lambda t: {"params": {k: (v if k != "path_sampler" else tree(masked)["path_sampler"]) for k, v in t.items()}}
    count = ext_array(np.asarray(step, np.int32))
    zero = ext_array(np.asarray(0, np.int32))
    all_masked = {"params": tree(masked)}
    adam_sched = {"0": {"count": count, "mu": trained(tree(mom(1e-3))), "nu": trained(tree(mom(1e-6)))}, "1": {"count": count}}
    inner = {"adam": {"inner_state": {"0": {"count": zero, "mu": all_masked, "nu": all_masked}, "1": {}}},
             "adam_lr_scheduler": {"inner_state": adam_sched},
             "adam_lr_scheduler1": {"inner_state": {"0": {"count": zero, "mu": all_masked, "nu": all_masked}, "1": {"count": zero}}},
             "zero": {"inner_state": {}}}
    # TrainState.step is a python int after create(); after one apply_gradients under pmap + unreplicate it is a 0-d int32 array,
    # which msgpack_serialize writes as an ndarray ext (jax arrays are converted with np.asarray): the reader must take both
    return {"step": ext_array(np.asarray(step, np.int32)), "params": {"params": tree(arr)}, "opt_state": {"inner_states": inner},
            "np_scalar_probe": ext_scalar(np.float32(1.5))}


if __name__ == "__main__":
    raw = msgpack.packb(build(), use_bin_type=True)
    out = os.path.join(HERE, "flax_checkpoint_7.msgpack.gz")
    with gzip.GzipFile(out, "wb", mtime=0) as f:
        f.write(raw)
    print("wrote", out, len(raw), "bytes raw,", os.path.getsize(out), "compressed")
