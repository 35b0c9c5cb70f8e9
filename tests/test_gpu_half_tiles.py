"""128-row tiles (one m-tile per wave) of the NerfMLP forward / dgrad kernels against the 256-row tiles: the same bits.

Launches whose 256-row tiles would leave more than half of the CUs idle run 128-row tiles instead (csrc/mlp.hip: launch_fwd_dbg,
launch_dgrad).  A row's arithmetic does not depend on the tile it sits in, so raw outputs, the saved operands the backward kernels read
(the padded rows included) and the dY planes must be IDENTICAL between the two tilings.  The switches are read once per process
(RNERF_FWD_HALF_TILES / RNERF_DGRAD_HALF_TILES; they exist only in librnerf_experiments.so, the same sources with -DRNERF_EXPERIMENTS),
hence one child process per setting; a third child runs the PRODUCT library with the switches set and must give the default's bits.
"""
import hashlib
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import hashlib, json, sys
import numpy as np, torch
sys.path.insert(0, %r)
from samplenerfro_amd import _lib, build, ops, synthetic as syn
import os
_lib.load(build.LIB if os.environ.get("RNERF_TEST_PRODUCT_LIB") else build.LIB_EXPERIMENTS)                # the switches exist only in the -DRNERF_EXPERIMENTS build; the product library reads no environment
dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
pf = syn.init_params_flat(7, bias_scale=0.1)["coarse_mlp"]
rng = np.random.default_rng(11)
B, S = 613, 9                                   # 5517 rows: 22 tiles of 256 (the last one ragged) or 44 of 128
pos = rng.uniform(-3, 3, (S, B, 3)).astype(np.float32)
d = rng.standard_normal((S, B, 3)).astype(np.float32); d /= np.linalg.norm(d, axis=-1, keepdims=True)
t = np.sort(rng.uniform(2, 6, (S, B, 1)).astype(np.float32), 0)
pd = np.concatenate([pos, t], -1); dr = np.concatenate([d, np.zeros((S, B, 1), np.float32)], -1)
out = {}
h = lambda x: hashlib.sha256(x.detach().cpu().numpy().tobytes()).hexdigest()
for prec in ("f16x3", "f16f8"):
    P = _lib.PRECISIONS[prec]
    packed = ops.nerfmlp_pack(T(pf), P)
    out["fwd_" + prec] = h(ops.nerfmlp_forward(packed, P, T(pd), T(dr), None, S, B))
P = _lib.PRECISIONS["f16x3"]
packed = ops.nerfmlp_pack(T(pf), P)
for name, BW in (("f32", _lib.BWD_F16X2), ("tf32", _lib.BWD_F16)):
    raw, save = ops.nerfmlp_forward_train(packed, P, T(pd), T(dr), None, S, B, BW)
    out["train_raw_" + name] = h(raw); out["save_" + name] = h(save[: _lib.load().rnerf_nerfmlp_save_bytes(S * B, BW) - 8192])
    pb = ops.nerfmlp_pack_bwd(T(pf), None, BW)
    d_raw = T((rng.standard_normal((S, B, 4)) * 1e-3).astype(np.float32))
    g, dy = ops.nerfmlp_backward(pb, packed, P, save, d_raw, S * B, backward=BW, return_dy=True)
    out["dy_" + name] = h(dy); out["grads_" + name] = h(g)
torch.cuda.synchronize()
print("RESULT " + json.dumps(out))
""" % ROOT


def _run(env_extra):
    env = dict(os.environ); env.update(env_extra)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    return json.loads(line[7:])


@pytest.mark.gpu
def test_half_tiles_give_the_same_bits_as_full_tiles():
    half = _run({})                                                                   # the default: 44 <= CUs -> 128-row tiles
    full = _run({"RNERF_FWD_HALF_TILES": "0", "RNERF_DGRAD_HALF_TILES": "0"})
    assert set(half) == set(full) and len(half) == 10
    for k in sorted(half):
        assert half[k] == full[k], f"{k}: 128-row tiles and 256-row tiles disagree"
    # the product library with the same variables set: deaf to them, and the same bits again
    prod = _run({"RNERF_FWD_HALF_TILES": "0", "RNERF_DGRAD_HALF_TILES": "0", "RNERF_TEST_PRODUCT_LIB": "1"})
    assert prod == half
