"""Compositing forward / backward and the resampling with 4, 16 or 64 lanes per ray: the same bits.

The kernels replay every ordered sum (optical depth, the five accumulations, the backward's suffix chains) in sample order whatever the
number of lanes that share a ray's activations (csrc/render.hip: composite_kernel<L>, composite_bwd_kernel<L>), so the width is a pure
scheduling choice of the launcher (composite_lanes).  RNERF_COMPOSITE_LANES (librnerf_experiments.so only) is read once per process: one
child process per width.
"""
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
from samplenerfro_amd import _lib, build, ops
_lib.load(build.LIB_EXPERIMENTS)                # RNERF_COMPOSITE_LANES exists only in the -DRNERF_EXPERIMENTS build of the same sources
dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
h = lambda x: hashlib.sha256(x.detach().cpu().numpy().tobytes()).hexdigest()
lib = _lib.load()
out = {}
for tag, (S, B, bd) in {"a": (67, 301, False), "b": (192, 50, True), "c": (5, 9, False)}.items():      # ragged groups for every width
    rng = np.random.default_rng(S)
    raw = rng.standard_normal((S, B, 4)).astype(np.float32)
    pos = rng.uniform(-1.5, 1.5, (S, B, 3)).astype(np.float32)
    t = np.sort(rng.uniform(2, 6, (S, B, 1)).astype(np.float32), 0)
    d = rng.standard_normal((S, B, 3)).astype(np.float32)
    pd = np.concatenate([pos, t], -1); dr = np.concatenate([d, np.zeros((S, B, 1), np.float32)], -1)
    bk = rng.uniform(0, 1, (B, 3)).astype(np.float32)
    pix = rng.uniform(0, 1, (B, 3)).astype(np.float32)
    bbox = [-1.0, -1.0, -1.0, 1.0, 1.0, 1.0] if bd else None
    for mode in ((0, 1, 2) if bd else (0,)):
        res = ops.composite(T(raw), T(pd), T(dr), None, S, B, T(bk), False, 0.001, -1.0, True, True, mode, bbox)
        out["fwd_%%s_%%d" %% (tag, mode)] = [h(x) for x in res if x is not None]
    rgb, dist, acc, trans, tb, w, a = ops.composite(T(raw), T(pd), T(dr), None, S, B, T(bk), False, 0.001, -1.0, True, True)
    sums = torch.tensor([1.0, 2.0, 3.0, float(B // 2)], device=dev)
    d_raw, d_bk = ops.composite_backward(T(raw), T(pd), T(dr), None, S, B, T(bk), rgb, T(pix), trans, tb, sums, 2.0 / (3 * B), 0.025,
                                         bd_cut_bbox=bbox)
    torch.cuda.synchronize()
    out["bwd_" + tag] = [h(d_raw), h(d_bk)]
# resampling along a path (sample_pdf): the weight-sum / cdf chains and the searches with 4 / 16 / 64 lanes per ray
for tag, (Sc, P, F, B) in {"r1": (16, 4, 24, 77), "r2": (64, 3, 128, 19)}.items():
    rng = np.random.default_rng(Sc + B)
    N = Sc * P
    tn = np.sort(rng.uniform(2, 6, (N, B)).astype(np.float32), 0)
    ppd = np.concatenate([rng.uniform(-1, 1, (N, B, 3)).astype(np.float32), tn[..., None]], -1)
    pdr = np.concatenate([rng.standard_normal((N, B, 3)).astype(np.float32), np.zeros((N, B, 1), np.float32)], -1)
    jit = (np.arange(Sc) * P + rng.integers(0, P, Sc)).astype(np.int32)
    wts = rng.uniform(0, 1, (Sc, B)).astype(np.float32); wts[:, 3] = 0.0                      # one ray without any weight (the padding branch)
    u = np.sort(rng.uniform(0, 1 - 1e-6, (F, B)).astype(np.float32), 0)
    rp, rd, idx = ops.resample(T(ppd), T(pdr), T(jit), T(wts), T(u), F, want_idx=True)
    out["resample_" + tag] = [h(rp), h(rd), h(idx)]
torch.cuda.synchronize()
print("RESULT " + json.dumps(out))
""" % ROOT


def _run(lanes):
    env = dict(os.environ); env["RNERF_COMPOSITE_LANES"] = str(lanes)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1][7:])


@pytest.mark.gpu
def test_compositing_gives_the_same_bits_for_every_lane_count():
    ref = _run(4)
    assert len(ref) >= 10
    for lanes in (16, 64):
        got = _run(lanes)
        for k in sorted(ref):
            assert got[k] == ref[k], f"{k}: {lanes} lanes per ray differ from 4"
