"""Per-tensor gradient error of the NerfMLP backward modes vs torch float64 (debug helper): python tools/r02/bwd_err.py [B] [S]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import ref_np as R, torch_ref as TR
from samplenerfro_amd import _lib, ops, synthetic as syn
F32 = np.float32
B = int(sys.argv[1]) if len(sys.argv) > 1 else 83
S = int(sys.argv[2]) if len(sys.argv) > 2 else 7
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")
rng = np.random.default_rng(9)
pf = syn.init_params_flat(12, fine=False, bias_scale=0.1)["coarse_mlp"]
pos = rng.uniform(-3, 3, (B, S, 3)).astype(F32)
dirs = R.safe_l2_normalize(rng.standard_normal((B, S, 3)).astype(F32))
pd = np.concatenate([pos, np.zeros((B, S, 1), F32)], -1).transpose(1, 0, 2)
dr = np.concatenate([dirs, np.zeros((B, S, 1), F32)], -1).transpose(1, 0, 2)
cot = (rng.standard_normal((S, B, 4)) * np.array([1e-3, 1e-3, 1e-3, 3e-4])).astype(F32)
SP = int(os.environ.get("SPREAD", "7"))
if SP & 1: cot[:, 5] = 0.0
if SP & 2: cot[:, 6] *= 1e-4
if SP & 4: cot[:, 7] *= float(os.environ.get("BIG", "1e3"))
flat = torch.tensor(pf, dtype=torch.float64, requires_grad=True)
enc = torch.tensor(R.pos_enc(pos.transpose(1, 0, 2).reshape(-1, 3), 0, 10), dtype=torch.float64)
venc = torch.tensor(R.pos_enc(dirs.transpose(1, 0, 2).reshape(-1, 3), 0, 4), dtype=torch.float64)
out = TR.nerf_mlp(flat, enc, venc)
(out * torch.tensor(cot.reshape(-1, 4), dtype=torch.float64)).sum().backward()
ref = flat.grad.numpy()
PRECN = os.environ.get("PREC", "f16x3")
P = _lib.PRECISIONS[PRECN]
flat_d = T(pf)
packed = ops.nerfmlp_pack(flat_d, P)
for bwd in (("bf16", "tf32", "f32") if PRECN == "f16x3" else ("bf16",)):
    BW = _lib.BACKWARDS[bwd]
    raw, save = ops.nerfmlp_forward_train(packed, P, T(pd.astype(F32)), T(dr.astype(F32)), None, S, B, BW)
    g = ops.nerfmlp_backward(ops.nerfmlp_pack_bwd(flat_d, None, BW), packed, P, save, T(cot), S * B, backward=BW).cpu().numpy().astype(np.float64)
    off = 0; line = []
    for k, (i, o) in enumerate(TR.NERF_MLP_SHAPES):
        for name, n in (("k", i * o), ("b", o)):
            a, r = g[off:off + n], ref[off:off + n]; off += n
            line.append(f"D{k}{name} {np.abs(a - r).max() / np.abs(r).max():.1e}")
    print(f"[{bwd}] rows {B*S}:", " ".join(line))
