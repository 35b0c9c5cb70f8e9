import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from samplenerfro_amd import ops, _lib, synthetic as syn
from oracle import ref_np as R
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
pf = syn.init_params_flat(7, bias_scale=0.1)
rng = np.random.default_rng(5)
B, S = 37, 11
pos = rng.uniform(-3, 3, (B, S, 3)).astype(np.float32)
dirs = R.safe_l2_normalize(rng.standard_normal((B, S, 3)).astype(np.float32))
pd = np.concatenate([pos.transpose(1, 0, 2), np.zeros((S, B, 1), np.float32)], -1)
dr = np.concatenate([dirs.transpose(1, 0, 2), np.zeros((S, B, 1), np.float32)], -1)
run = lambda flat, prec: ops.nerfmlp_forward(ops.nerfmlp_pack(T(flat), prec), prec, T(pd), T(dr), None, S, B).cpu().numpy()
for val in (1e2, 1e4, 7e4, 3e5):
    hot = pf["coarse_mlp"].copy()
    off = 63 * 256
    hot[off:off + 256] = val
    a, b = run(hot, _lib.PREC_F16X3), run(hot, _lib.PREC_F32)
    print(val, "f16x3", a.reshape(-1)[:4], np.isfinite(a).all(), "f32", b.reshape(-1)[:4])
