#!/usr/bin/env python3
"""Feasibility probe (timing only, results garbage): the dgrad on a capped grid and the wgrad on the remaining CUs AT THE SAME TIME (two
streams; the wgrad reads the dY of an earlier call), against the two in sequence on the whole chip.
usage: RNERF_DGRAD_WG=<n> RNERF_WGRAD_WGS=<m> python3 tools/r04/concurrent_bwd.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from samplenerfro_amd import _lib, ops, synthetic as syn
rows = 4096 * 128
B, S = 4096, rows // 4096
dev = "cuda:0"
P, BW = _lib.PRECISIONS["f16x3"], _lib.BACKWARDS["f32"]
pf = torch.from_numpy(syn.init_params_flat(0, fine=False)["coarse_mlp"]).to(dev)
packed = ops.nerfmlp_pack(pf, P); pbwd = ops.nerfmlp_pack_bwd(pf, None, BW)
g = torch.Generator(device=dev).manual_seed(0)
pd = torch.rand((S, B, 4), device=dev, generator=g) * 2 - 1
dr = torch.nn.functional.normalize(torch.randn((S, B, 4), device=dev, generator=g), dim=-1)
d_raw = torch.randn((S, B, 4), device=dev, generator=g) * 1e-3
raw, save = ops.nerfmlp_forward_train(packed, P, pd, dr, None, S, B, BW)
lib = _lib.load()
dy = torch.empty(lib.rnerf_nerfmlp_dy_bytes(rows, BW), dtype=torch.uint8, device=dev)
dy2 = torch.empty_like(dy)
ws = torch.empty(lib.rnerf_nerfmlp_wgrad_workspace_bytes(), dtype=torch.uint8, device=dev)
grads = torch.empty(_lib.NERFMLP_PARAMS, device=dev)
ops.nerfmlp_backward(pbwd, packed, P, save, d_raw, rows, dy=dy2, stages="d", backward=BW)      # a valid dY for the wgrad to read
torch.cuda.synchronize()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def both():
    with torch.cuda.stream(s1):
        ops.nerfmlp_backward(pbwd, packed, P, save, d_raw, rows, dy=dy, stages="d", backward=BW)
    with torch.cuda.stream(s2):
        ops.nerfmlp_backward(pbwd, packed, P, save, d_raw, rows, grads=grads, workspace=ws, dy=dy2, stages="w", backward=BW)
def seq():
    ops.nerfmlp_backward(pbwd, packed, P, save, d_raw, rows, dy=dy, stages="d", backward=BW)
    ops.nerfmlp_backward(pbwd, packed, P, save, d_raw, rows, grads=grads, workspace=ws, dy=dy2, stages="w", backward=BW)
def wall(fn, n=5):
    import time
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(n):
        torch.cuda.synchronize(); t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(1e3 * (time.perf_counter() - t))
    return min(ts), float(np.median(ts))
mode = "concurrent" if os.environ.get("RNERF_DGRAD_WG") else "sequential"
print(mode, "dgrad wgs", os.environ.get("RNERF_DGRAD_WG", "all"), "wgrad wgs", os.environ.get("RNERF_WGRAD_WGS", "2 x CUs"),
      "ms min/med = %.3f / %.3f" % wall(both if mode == "concurrent" else seq))
