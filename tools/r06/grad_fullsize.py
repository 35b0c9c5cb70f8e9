"""The gradient of ONE train step at bench size (ship_straight, 4096 rays x 128 samples = 524 288 rows, the real loss: d raw spans the orders of
magnitude compositing weights span) against float64 autograd of the same loss on the same rows — per backward mode, on glorot-initialised
weights and on trained-like ones (hidden kernels x 1.5, N(0, 0.3) biases: sharper densities, a wider spread of per-row gradients).
python tools/r06/grad_fullsize.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from oracle import ref_np as R, torch_ref as TR
from samplenerfro_amd import synthetic as syn, prng, utils as U, _lib
from samplenerfro_amd.utils import Rays
from samplenerfro_amd.train import TrainState, train_step
dev = torch.device("cuda:0")
cfg = dict(syn.CONFIGS["ship_straight"])
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
model, variables, pf0 = bench.build_scene(cfg, dev, "f16x3", 0, "radiance", None)
o, d = syn.sphere_rays(B, seed=syn.SEED)
rays = Rays(torch.from_numpy(o).to(dev), None, torch.from_numpy(d).to(dev), None)
key = prng.PRNGKey(syn.SEED)
gen = np.random.default_rng(5)
pix = torch.from_numpy(gen.uniform(0, 1, (B, 3)).astype(np.float32)).to(dev)


def scaled(pf, scale, bias):
    out = dict(pf)
    flat = pf["coarse_mlp"].copy()
    off, rng = 0, np.random.default_rng(99)
    for k, (fi, fo) in enumerate(syn.NERF_MLP_SHAPES):
        if 1 <= k <= 7:
            flat[off:off + fi * fo] *= scale
        if bias > 0:
            flat[off + fi * fo:off + fi * fo + fo] = (bias * rng.standard_normal(fo)).astype(np.float32)
        off += fi * fo + fo
    out["coarse_mlp"] = flat
    return out


for tag, pf in (("glorot", pf0), ("hidden kernels x 1.5, biases N(0, 0.3)", scaled(pf0, 1.5, 0.3))):
    ref = None
    for bw in ("f16x3", "f16x3lo8", "f16"):
        vv = bench.models_fresh_variables(pf, dev)
        fl = U.default_flags(num_coarse_samples=cfg["S"], num_fine_samples=0, num_path_samples=cfg["P"], white_bkgd=False, bg_weight=0.025, bg_smooth_weight=0.0,
                             use_online_sparsity=False, randomized=True, near=cfg["near"], far=cfg["far"], batch_size=B, backward_precision=bw, stage="radiance")
        ts = TrainState.create(model, vv, fl)
        tp = {}
        train_step(model, key, ts, {"rays": rays, "pixels": pix, "annealed_alpha": 0.5}, fl, taps=tp)
        g = tp["grads"].double()
        if ref is None:                       # float64 autograd of loss_fn on the rows the device used, on the device
            ctx = tp["ctx"]
            jit = ctx["jit"].long()
            pd, dr = ctx["path_pd"][jit], ctx["path_dr"][jit]
            S = pd.shape[0]
            pos = pd[..., :3].permute(1, 0, 2).reshape(-1, 3).cpu().numpy(); dirs = dr[..., :3].permute(1, 0, 2).reshape(-1, 3).cpu().numpy()
            enc = torch.tensor(R.pos_enc(pos, 0, 10), dtype=torch.float64, device=dev); venc = torch.tensor(R.pos_enc(dirs, 0, 4), dtype=torch.float64, device=dev)
            t = pd[..., 3].permute(1, 0).double(); dirs_t = torch.tensor(dirs, dtype=torch.float64, device=dev).reshape(B, S, 3)
            last = torch.tensor(R.pos_enc(ctx["path_dr"][int(jit[-1])][:, :3].cpu().numpy(), 0, 4), dtype=torch.float64, device=dev)
            th = ts.theta.detach().double().clone().requires_grad_(True)
            seg = ts.segments
            bk = TR.bkgd_mlp(th[seg["bkgd_mlp"][0]:seg["bkgd_mlp"][1]], last, model.rgb_padding)
            raw = TR.nerf_mlp(th[seg["coarse_mlp"][0]:seg["coarse_mlp"][1]], enc, venc).reshape(B, S, 4)
            rgb, sigma = TR.activations(raw, model.rgb_padding, model.sigma_bias)
            comp, acc, w, trans, tb = TR.volumetric_rendering(rgb, sigma, t, dirs_t, bk)
            total, parts = TR.radiance_loss([(comp, trans, tb)], pix.double(), fl.bg_weight, 0.5)
            total.backward()
            ref = th.grad.detach()
            wmax = w.detach().max(1).values
            print(f"[{tag}] {B} rays x {S} samples: loss {float(parts['loss']):.4f}; compositing weights: median of the rays' largest {float(wmax.median()):.3f}, acc mean {float(acc.mean()):.3f}; "
                  f"max |raw| {float(raw.abs().max()):.1f}")
            del enc, venc, raw, rgb, sigma, comp, acc, w, trans, tb, total
        lo, hi = ts.segments["coarse_mlp"]
        worst, wname, off = 0.0, "", lo
        for k, (fi, fo) in enumerate(syn.NERF_MLP_SHAPES):
            for nm, n in (("kernel", fi * fo), ("bias", fo)):
                a, r = g[off:off + n], ref[off:off + n]
                off += n
                e = float((a - r).abs().max() / r.abs().max())
                if e > worst: worst, wname = e, f"Dense_{k} {nm}"
        whole = float((g[lo:hi] - ref[lo:hi]).abs().max() / ref[lo:hi].abs().max())
        bko, bkh = ts.segments["bkgd_mlp"]
        print(f"    backward {bw:9s}: NerfMLP gradient vs float64: worst tensor {worst:.2e} ({wname}), whole vector {whole:.2e}; background MLP {float((g[bko:bkh] - ref[bko:bkh]).abs().max() / ref[bko:bkh].abs().max()):.2e}", flush=True)
        del ts, vv, tp
        torch.cuda.empty_cache()
