#!/usr/bin/env python3
"""Background MLP forward (train) / backward: time per call and difference between the f16 hi+lo kernels and the exact-fp32 ones.

usage: RNERF_BKGD_EXACT=0|1 python tools/r04/bkgd_time.py [rows] [out.npy]
Writes the outputs (rgb, save head, grads) of this mode to out.npy so that the two modes (the switch is read once per process) can be
compared by tools/r04/bkgd_time.sh.
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from samplenerfro_amd import ops, synthetic as syn          # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20480
    out_path = sys.argv[2] if len(sys.argv) > 2 else None
    dev = torch.device("cuda:0")
    pf = syn.init_params_flat(3, bias_scale=0.1)["bkgd_mlp"]
    rng = np.random.default_rng(4)
    d = rng.standard_normal((n, 3)).astype(np.float32)
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    P = torch.from_numpy(pf).to(dev); D = torch.from_numpy(d).to(dev)
    dout = torch.from_numpy(rng.standard_normal((n, 3)).astype(np.float32) * 1e-3).to(dev)
    grads = torch.zeros_like(P)

    def timed(f, reps=50):
        for _ in range(5):
            f()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            f()
        b.record(); torch.cuda.synchronize()
        return a.elapsed_time(b) / reps * 1e3

    t_inf = timed(lambda: ops.bkgd_forward(P, D))
    t_fwd = timed(lambda: ops.bkgd_forward_train(P, D))
    rgb, save = ops.bkgd_forward_train(P, D)
    t_bwd = timed(lambda: ops.bkgd_backward(P, save, dout, grads))
    grads.zero_()
    ops.bkgd_backward(P, save, dout, grads)
    mode = "exact" if os.environ.get("RNERF_BKGD_EXACT") == "1" else "f16x3"
    print(f"{mode:6s} rows {n}: forward {t_inf:.1f} us, forward(train) {t_fwd:.1f} us, backward {t_bwd:.1f} us (incl. host launch)")
    if out_path:
        np.save(out_path, {"rgb": rgb.cpu().numpy(), "save": save.view(torch.float32).cpu().numpy(), "grads": grads.cpu().numpy()}, allow_pickle=True)


if __name__ == "__main__":
    main()
