#!/usr/bin/env python3
"""Compositing forward / backward alone: time per call for the lanes-per-ray setting of this process (RNERF_COMPOSITE_LANES = 4 | 16 | 64 | unset).
usage: python tools/r04/composite_time.py [B S ...]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from samplenerfro_amd import ops
dev = torch.device("cuda:0")
args = [int(a) for a in sys.argv[1:]] or [128, 192, 512, 192, 4096, 128, 4096, 192, 32768, 128]
for B, S in zip(args[::2], args[1::2]):
    rng = np.random.default_rng(1)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    raw = T(rng.standard_normal((S, B, 4)).astype(np.float32))
    t = np.sort(rng.uniform(2, 6, (S, B, 1)).astype(np.float32), 0)
    pd = T(np.concatenate([rng.uniform(-1, 1, (S, B, 3)).astype(np.float32), t], -1))
    dr = T(np.concatenate([rng.standard_normal((S, B, 3)).astype(np.float32), np.zeros((S, B, 1), np.float32)], -1))
    bk = T(rng.uniform(0, 1, (B, 3)).astype(np.float32)); pix = T(rng.uniform(0, 1, (B, 3)).astype(np.float32))
    rgb, dist, acc, trans, tb, w, a = ops.composite(raw, pd, dr, None, S, B, bk)
    sums = torch.tensor([1.0, 2.0, 3.0, float(B // 2)], device=dev)
    def timed(f, reps=30):
        for _ in range(5): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    tf = timed(lambda: ops.composite(raw, pd, dr, None, S, B, bk))
    tb_ = timed(lambda: ops.composite_backward(raw, pd, dr, None, S, B, bk, rgb, pix, trans, tb, sums, 2.0 / (3 * B), 0.025))
    print(f"lanes {os.environ.get('RNERF_COMPOSITE_LANES', 'auto'):>4s}  B {B:6d} S {S:4d}: forward {tf:7.1f} us  backward {tb_:7.1f} us  (incl. host launch + allocations)")
