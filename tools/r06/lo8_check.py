#!/usr/bin/env python3
"""Backward mode f16x3lo8 (e4m3 lo planes) against f16x3 and float64 autograd: gradient error at 581 and 524 288 rows, kernel times.
python tools/r06/lo8_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import ref_np as R, torch_ref as TR
from samplenerfro_amd import _lib, ops, synthetic as syn

dev = "cuda:0"
F32 = np.float32


def case(B, S, seed, scale_w=1.0, bias=0.1):
    rng = np.random.default_rng(seed)
    pf = syn.init_params_flat(12, fine=False, bias_scale=bias)["coarse_mlp"].copy()
    if scale_w != 1.0:
        off = 0
        for k, (i, o) in enumerate(TR.NERF_MLP_SHAPES):
            if 1 <= k <= 7:
                pf[off:off + i * o] *= scale_w
            off += i * o + o
    pos = rng.uniform(-3, 3, (B, S, 3)).astype(F32)
    dirs = R.safe_l2_normalize(rng.standard_normal((B, S, 3)).astype(F32))
    pd = np.concatenate([pos, np.zeros((B, S, 1), F32)], -1).transpose(1, 0, 2)
    dr = np.concatenate([dirs, np.zeros((B, S, 1), F32)], -1).transpose(1, 0, 2)
    cot = (rng.standard_normal((S, B, 4)) * np.array([1e-3, 1e-3, 1e-3, 3e-4])).astype(F32)
    cot[:, 5] = 0.0
    cot[:, 6] *= 1e-4; cot[:, 7] *= 1e3
    return pf, pos, dirs, pd.astype(F32), dr.astype(F32), cot


def run(B, S, seed=9, scale_w=1.0):
    pf, pos, dirs, pd, dr, cot = case(B, S, seed, scale_w)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    flat_d = T(pf)
    P = _lib.PREC_F16X3
    packed = ops.nerfmlp_pack(flat_d, P)
    # float64 reference on the device
    flat = torch.tensor(pf, dtype=torch.float64, device=dev, requires_grad=True)
    enc = torch.tensor(R.pos_enc(pos.transpose(1, 0, 2).reshape(-1, 3), 0, 10), dtype=torch.float64, device=dev)
    venc = torch.tensor(R.pos_enc(dirs.transpose(1, 0, 2).reshape(-1, 3), 0, 4), dtype=torch.float64, device=dev)
    out = TR.nerf_mlp(flat, enc, venc)
    (out * torch.tensor(cot.reshape(-1, 4), dtype=torch.float64, device=dev)).sum().backward()
    ref = flat.grad.cpu().numpy()
    res = {}
    for name in ("f16x3", "f16x3lo8", "f16"):
        BW = _lib.BACKWARDS[name]
        raw, save = ops.nerfmlp_forward_train(packed, P, T(pd), T(dr), None, S, B, BW)
        grads = ops.nerfmlp_backward(ops.nerfmlp_pack_bwd(flat_d, None, BW), packed, P, save, T(cot), S * B, backward=BW).cpu().numpy().astype(np.float64)
        assert np.isfinite(grads).all(), name
        off = 0; worst = 0.0; wname = ""
        for k, (i, o) in enumerate(TR.NERF_MLP_SHAPES):
            for nm, n in (("kernel", i * o), ("bias", o)):
                g, r = grads[off:off + n], ref[off:off + n]
                off += n
                err = np.abs(g - r).max() / np.abs(r).max()
                if err > worst: worst, wname = err, f"Dense_{k} {nm}"
        res[name] = (worst, wname, np.abs(grads - ref).max() / np.abs(ref).max())
        raw_err = float(np.abs(out.detach().cpu().numpy() - raw.cpu().numpy().reshape(-1, 4)).max())
    print(f"rows {B * S:7d} hidden-kernel scale {scale_w}: raw err {raw_err:.1e}; worst per-tensor gradient error vs float64 (of the tensor's max):")
    for k, v in res.items():
        print(f"    {k:9s} {v[0]:.2e} ({v[1]})   whole-vector {v[2]:.2e}")
    return res


run(83, 7)
run(83, 7, scale_w=1.5)
run(83, 7, seed=10, scale_w=3.0)
run(4096, 16, seed=11)
run(4096, 128, seed=12)
