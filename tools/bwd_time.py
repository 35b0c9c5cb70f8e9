#!/usr/bin/env python3
"""Time the NerfMLP training kernels alone (forward_train, dgrad, wgrad) on synthetic operands: python tools/bwd_time.py [rows]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from samplenerfro_amd import _lib, ops, synthetic as syn

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4096 * 128
B, S = 4096, rows // 4096
dev = "cuda:0"
P = _lib.PRECISIONS[os.environ.get("PREC", "f16x3")]
BW = _lib.BACKWARDS[os.environ.get("BWD", "f32")]
pf = torch.from_numpy(syn.init_params_flat(0, fine=False)["coarse_mlp"]).to(dev)
packed = ops.nerfmlp_pack(pf, P); pbwd = ops.nerfmlp_pack_bwd(pf, None, BW)
g = torch.Generator(device=dev).manual_seed(0)
pd = torch.rand((S, B, 4), device=dev, generator=g) * 2 - 1
dr = torch.nn.functional.normalize(torch.randn((S, B, 4), device=dev, generator=g), dim=-1)
d_raw = torch.randn((S, B, 4), device=dev, generator=g) * 1e-3
raw, save = ops.nerfmlp_forward_train(packed, P, pd, dr, None, S, B, BW)
lib = _lib.load()
dy = torch.empty(lib.rnerf_nerfmlp_dy_bytes(rows, BW), dtype=torch.uint8, device=dev)
ws = torch.empty(lib.rnerf_nerfmlp_wgrad_workspace_bytes(), dtype=torch.uint8, device=dev)
grads = torch.empty(_lib.NERFMLP_PARAMS, device=dev)

def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    return min(ev[i].elapsed_time(ev[i + 1]) for i in range(n))

print("backward", os.environ.get("BWD", "f32"), "rows", rows, "save MB", save.numel() / 1e6, "dy MB", dy.numel() / 1e6)
print("forward        %.3f ms" % timeit(lambda: ops.nerfmlp_forward(packed, P, pd, dr, None, S, B, out=raw)))
print("forward_train  %.3f ms" % timeit(lambda: lib.rnerf_nerfmlp_forward_train(packed.data_ptr(), P, pd.data_ptr(), dr.data_ptr(), None, S, B, raw.data_ptr(), save.data_ptr(), BW, 0, None)))
print("dgrad          %.3f ms" % timeit(lambda: ops.nerfmlp_backward(pbwd, packed, P, save, d_raw, rows, dy=dy, stages="d", backward=BW)))
print("wgrad+reduce   %.3f ms" % timeit(lambda: ops.nerfmlp_backward(pbwd, packed, P, save, d_raw, rows, grads=grads, workspace=ws, dy=dy, stages="w", backward=BW)))

if os.environ.get("DGRAD_PROFILE"):       # library built with -DRNERF_DGRAD_PROFILE (tools/r02/build_ablate.sh)
    ops.nerfmlp_backward(pbwd, packed, P, save, d_raw, rows, dy=dy, stages="d", backward=BW); torch.cuda.synchronize()
    pr = dy[:256 * 4 * 32].view(torch.float32).reshape(-1, 8).cpu().numpy()[:, :5]
    per_tile = pr.sum(0) / pr.shape[0] / (rows / 256 / 256)
    for n, v in zip(["rows + head gradients", "first operands of a layer (grad_ops(0), masks)", "k-steps", "layer end", "dY_0 record"], per_tile):
        print(f"  {n:48s} {v:9.0f} clk/tile {100 * v / per_tile.sum():5.1f} %")
    print(f"  total {per_tile.sum():.0f} clk/tile")
