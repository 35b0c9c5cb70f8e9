"""What limits the weight gradient at 524 288 rows: fp32 accumulation, or the per-row scale m_row / m_ref applied to the saved activations as
packed f16 (rows far below the largest lose significand bits to f16's subnormal range)?  The same NerfMLP backward on cotangents whose
per-row magnitudes are (a) all alike, (b) log-uniform over 3 / 6 decades, (c) alike but for ONE ray at 1000 x.  python tools/r06/grad_rowscale.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import ref_np as R, torch_ref as TR
from samplenerfro_amd import _lib, ops, synthetic as syn
dev = "cuda:0"
F32 = np.float32
B, S = 4096, 128
rng = np.random.default_rng(12)
pf = syn.init_params_flat(12, fine=False, bias_scale=0.1)["coarse_mlp"].copy()
pos = rng.uniform(-3, 3, (B, S, 3)).astype(F32)
dirs = R.safe_l2_normalize(rng.standard_normal((B, S, 3)).astype(F32))
pd = np.concatenate([pos, np.zeros((B, S, 1), F32)], -1).transpose(1, 0, 2).astype(F32)
dr = np.concatenate([dirs, np.zeros((B, S, 1), F32)], -1).transpose(1, 0, 2).astype(F32)
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
flat_d = T(pf)
P = _lib.PREC_F16X3
packed = ops.nerfmlp_pack(flat_d, P)
enc = torch.tensor(R.pos_enc(pos.transpose(1, 0, 2).reshape(-1, 3), 0, 10), dtype=torch.float64, device=dev)
venc = torch.tensor(R.pos_enc(dirs.transpose(1, 0, 2).reshape(-1, 3), 0, 4), dtype=torch.float64, device=dev)
base = (rng.standard_normal((S, B, 4)) * np.array([1e-3, 1e-3, 1e-3, 3e-4])).astype(F32)
cases = {"all rows alike": np.ones((S, B, 1), F32),
         "log-uniform over 3 decades": (10.0 ** rng.uniform(-3, 0, (S, B, 1))).astype(F32),
         "log-uniform over 6 decades": (10.0 ** rng.uniform(-6, 0, (S, B, 1))).astype(F32),
         "alike, one ray x 1000": np.ones((S, B, 1), F32)}
cases["alike, one ray x 1000"][:, 7] = 1000.0
for name, scale in cases.items():
    cot = (base * scale).astype(F32)
    flat = torch.tensor(pf, dtype=torch.float64, device=dev, requires_grad=True)
    out = TR.nerf_mlp(flat, enc, venc)
    (out * torch.tensor(cot.reshape(-1, 4), dtype=torch.float64, device=dev)).sum().backward()
    ref = flat.grad
    line = []
    for bw in ("f16x3", "f16"):
        BW = _lib.BACKWARDS[bw]
        raw, save = ops.nerfmlp_forward_train(packed, P, T(pd), T(dr), None, S, B, BW)
        g = ops.nerfmlp_backward(ops.nerfmlp_pack_bwd(flat_d, None, BW), packed, P, save, T(cot), S * B, backward=BW).double()
        off, worst, wn = 0, 0.0, ""
        for k, (i, o) in enumerate(TR.NERF_MLP_SHAPES):
            for nm, n in (("kernel", i * o), ("bias", o)):
                e = float((g[off:off + n] - ref[off:off + n]).abs().max() / ref[off:off + n].abs().max())
                if e > worst: worst, wn = e, f"Dense_{k} {nm}"
                off += n
        line.append(f"{bw}: worst tensor {worst:.2e} ({wn}), whole {float((g - ref).abs().max() / ref.abs().max()):.2e}")
        del save
    print(f"{name:28s} " + "   ".join(line), flush=True)
