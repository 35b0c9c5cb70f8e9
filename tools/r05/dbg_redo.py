import sys, numpy as np, torch
sys.path.insert(0, ".")
from oracle import ref_np as R
from samplenerfro_amd import _lib, ops, synthetic as syn
F32=np.float32
T=lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
pf = syn.init_params_flat(7, bias_scale=0.1)
rng = np.random.default_rng(5)
B, S = 37, 11
pos = rng.uniform(-3, 3, (B, S, 3)).astype(F32)
dirs = R.safe_l2_normalize(rng.standard_normal((B, S, 3)).astype(F32))
t = np.sort(rng.uniform(2, 6, (B, S)).astype(F32), -1)
pd = np.concatenate([pos, t[..., None]], -1).transpose(1, 0, 2).copy(); dr = np.concatenate([dirs, np.zeros((B, S, 1), F32)], -1).transpose(1, 0, 2).copy()
hot = pf["coarse_mlp"].copy(); off = 63 * 256; hot[off:off + 256] = 3.0e5
P=_lib.PREC_F16X3
raw_t,_ = ops.nerfmlp_forward_train(ops.nerfmlp_pack(T(hot), P), P, T(pd), T(dr), None, S, B, _lib.BWD_F16X3)
print("train fwd f16x3 (no second pass):", raw_t.flatten()[:8].cpu().numpy())
for name in ("f16x3","bf16x3","f32","f16f8"):
    p=_lib.PRECISIONS[name]
    out = ops.nerfmlp_forward(ops.nerfmlp_pack(T(hot), p), p, T(pd), T(dr), None, S, B)
    print(name, out.flatten()[:8].cpu().numpy())
