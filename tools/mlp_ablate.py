"""Profiling helper (not part of the product): time the PE+NerfMLP kernel alone, optionally under RNERF_MLP_DEBUG ablations."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from samplenerfro_amd import ops, _lib, synthetic as syn
if os.environ.get("RNERF_LIB"):          # an ablation / experiments build (tools/r03/build_variant.sh): bound before anything else loads the product library
    _lib.load(os.environ["RNERF_LIB"])
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
S, B = 128, 4096
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
pd = torch.from_numpy(rng.uniform(-2, 2, (S, B, 4)).astype(np.float32)).to(dev)
dr = torch.from_numpy(rng.uniform(-1, 1, (S, B, 4)).astype(np.float32)).to(dev)
pf = syn.init_params_flat(0, fine=False)
packed = ops.nerfmlp_pack(torch.from_numpy(pf["coarse_mlp"]).to(dev), _lib.PRECISIONS[prec])
out = ops.nerfmlp_forward(packed, _lib.PRECISIONS[prec], pd, dr, None, S, B)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
ev[0].record()
for i in range(10):
    ops.nerfmlp_forward(packed, _lib.PRECISIONS[prec], pd, dr, None, S, B, out=out)
    ev[i + 1].record()
torch.cuda.synchronize()
ms = np.array([ev[i].elapsed_time(ev[i + 1]) for i in range(10)])
fl = 2 * 593408 * S * B
print(f"prec={prec} dbg={os.environ.get('RNERF_MLP_DEBUG','0')} ms(min/med)={ms.min():.3f}/{np.median(ms):.3f}  algTF/s={fl/np.median(ms)/1e9:.1f}")

if os.environ.get('RNERF_MLP_DEBUG') == '64':
    prof = out.reshape(-1, 4)[:1024].cpu().numpy()
    n = prof[:, 3].mean()
    print(f"slabs/wave={n:.0f}  cycles/slab total={prof[:,0].sum()/prof[:,3].sum():.0f}  dma-wait={prof[:,1].sum()/prof[:,3].sum():.0f}  barrier-wait={prof[:,2].sum()/prof[:,3].sum():.0f}")
    w = prof.reshape(-1, 4, 4)
    print("per-wave-slot barrier wait:", [round(float(w[:, k, 2].sum()/w[:, k, 3].sum())) for k in range(4)], " dma:", [round(float(w[:, k, 1].sum()/w[:, k, 3].sum())) for k in range(4)])

if os.environ.get('RNERF_MLP_DEBUG') == '256':
    if os.environ.get('TRAIN'):           # the training forward (hi + lo saves) instead of the evaluation forward
        out, _save = ops.nerfmlp_forward_train(packed, _lib.PRECISIONS[prec], pd, dr, None, S, B, _lib.BWD_F16X2)
        torch.cuda.synchronize()
    prof = out.reshape(-1, 4)[:256 * 4 * 3].cpu().numpy().reshape(-1, 12)[:, :9]
    names = ["row loads", "layer 0 (PE slabs)", "hidden k-step 0 (+conversion)", "hidden k-steps 1..15", "layer end", "skip slabs", "sigma head", "view layer", "rgb head + store"]
    tot = prof.sum()
    per_tile = prof.sum(0) / prof.shape[0] / 8          # 8 tiles per workgroup on this workload
    for n, v in zip(names, per_tile):
        print(f"  {n:32s} {v:9.0f} clk/tile  {100 * v / per_tile.sum():5.1f} %")
    print(f"  total {per_tile.sum():.0f} clk/tile; MFMA floor 145 slabs x 1536 = {145 * 1536}")
