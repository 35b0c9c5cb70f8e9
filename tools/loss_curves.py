#!/usr/bin/env python3
"""Loss curves of the three backward arithmetics against a torch fp32 autograd loop (not a test; writes a JSON under gpurun_out/).

    python tools/loss_curves.py [steps=1000] [out.json]

Teacher/student set-up at 4096 rays x 64 samples (flat, fixed quadrature nodes, so the sampled rows do not depend on the parameters):
a teacher parameter set renders the target pixels of a fixed pool of 16 ray batches; the SAME student initialisation is then trained
  * with train_step in each backward mode ("f32" = f16 hi+lo parts, "tf32" = f16 parts, "bf16"), and
  * by a plain PyTorch fp32 loop on the GPU (oracle/torch_ref.py's restatement of train.py's loss_fn + torch.autograd + the optax Adam
    formula) on the rows the HIP march produced — the stand-in for the reference's fp32 jax.value_and_grad (train.py:164).
All four use the same schedule, jitter and batches; the curves differ only through the gradient arithmetic.
"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from oracle import torch_ref as TR
from samplenerfro_amd import models, ops, prng, synthetic as syn, utils as U
from samplenerfro_amd.train import TrainState, train_step
from samplenerfro_amd.utils import Rays, learning_rate_decay

dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
out_path = sys.argv[2] if len(sys.argv) > 2 else "gpurun_out/r02/loss_curves.json"
G, B, S, P, POOL = 128, 4096, 64, 8, 16
a = torch.linspace(-1.5, 1.5, G, dtype=torch.float64, device=dev)
r = torch.sqrt(a[:, None, None] ** 2 + a[None, :, None] ** 2 + a[None, None, :] ** 2)
grid = (1.0 + 0.5 * torch.clamp((0.6 - r) / (3.0 / (G - 1)) + 0.5, 0.0, 1.0)).float()


def make_flags(bwd):
    return U.default_flags(num_coarse_samples=S, num_fine_samples=0, num_path_samples=P, white_bkgd=False, bg_weight=0.025, bg_smooth_weight=1.0,
                           bg_patch_size=32, use_online_sparsity=False, randomized=True, lr_init=1e-3, lr_final=1e-4, lr_delay_steps=0, max_steps=steps,
                           backward_precision=bwd)


flags = make_flags("f32")
model, variables0 = models.construct_nerf(np.array([0, 1], np.uint32), None, flags, [G] * 3, [-1.5] * 3, [1.5] * 3, grid)
init = {k: v.clone() for k, v in variables0["flat"].items()}
teacher = models.make_variables({k: torch.from_numpy(v).to(dev) for k, v in syn.init_params_flat(123, fine=False, bias_scale=0.3).items()})
gen = np.random.default_rng(0)
ev = gen.standard_normal((32, 32, 3)).astype(np.float32); ev /= np.linalg.norm(ev, axis=-1, keepdims=True)
env = Rays(None, None, torch.from_numpy(ev).to(dev), None)
key = np.array([9, 9], np.uint32)
fixed = np.arange(0, S * P, P) + P // 2
pool = []
for i in range(POOL):
    o, d = syn.sphere_rays(B, seed=1000 + i)
    rays = Rays(torch.from_numpy(o).to(dev), None, torch.from_numpy(d).to(dev), None)
    with torch.no_grad():
        pix = model.apply(teacher, key, key, rays, False, jitter=fixed)[0][-1][0].clone()
    pool.append((rays, pix))
curves = {}
for bwd in ("f32", "tf32", "bf16"):
    fl = make_flags(bwd)
    variables = models.make_variables({k: v.clone() for k, v in init.items()})
    state = TrainState.create(model, variables, fl)
    rng = prng.PRNGKey(5)
    losses = []
    t0 = time.perf_counter()
    for step in range(steps):
        rays, pix = pool[step % POOL]
        state, stats, rng = train_step(model, rng, state, {"rays": rays, "pixels": pix, "annealed_alpha": 0.5, "env_rays": env}, fl, jitter=fixed)
        losses.append(stats.loss.clone())              # (the Stats fields are views into the step's buffer)
    curves[bwd] = [float(x) for x in torch.stack([l.reshape(()) for l in losses]).cpu()]
    print(f"[{bwd}] {steps} steps in {time.perf_counter() - t0:.1f} s; loss {curves[bwd][0]:.5f} -> {curves[bwd][-1]:.6f}", flush=True)

# ---- torch fp32 autograd on the GPU, on the rows of the HIP march ---------------------------------------------------------------
names = ["coarse_mlp", "bkgd_mlp"]
th = {k: init[k].clone().float().requires_grad_(True) for k in names}
mu = {k: torch.zeros_like(th[k]) for k in names}; nu = {k: torch.zeros_like(th[k]) for k in names}
jit = torch.from_numpy(fixed.astype(np.int64)).to(dev)


def pos_enc(x, L):
    scales = (2.0 ** torch.arange(L, device=x.device, dtype=x.dtype))
    xb = (x[..., None, :] * scales[:, None]).reshape(*x.shape[:-1], -1)
    return torch.cat([x, torch.sin(torch.cat([xb, xb + 0.5 * np.pi], -1))], -1)


rows = []
for rays, pix in pool:
    pd, dr, _, _ = ops.march(model.table, model.spec, rays.origins, rays.viewdirs, model.near, model.far, S * P)
    pd, dr = pd[jit], dr[jit]                                        # [S, B, 4]
    pos = pd[..., :3].permute(1, 0, 2).reshape(-1, 3); dirs = dr[..., :3].permute(1, 0, 2).reshape(-1, 3)
    rows.append((pos_enc(pos, 10), pos_enc(dirs, 4), pd[..., 3].permute(1, 0).contiguous(), dirs.reshape(B, S, 3), pos_enc(dr[-1][:, :3], 4)))
env_enc = pos_enc(torch.from_numpy(ev.reshape(-1, 3)).to(dev), 4)
torch_losses = []
t0 = time.perf_counter()
for step in range(steps):
    enc, venc, t, dirs_t, last = rows[step % POOL]
    pix = pool[step % POOL][1]
    bk = TR.bkgd_mlp(th["bkgd_mlp"], last)
    raw = TR.nerf_mlp(th["coarse_mlp"], enc, venc).reshape(B, S, 4)
    rgb, sigma = TR.activations(raw)
    comp, acc, w, trans, tb = TR.volumetric_rendering(rgb, sigma, t, dirs_t, bk)
    total, parts = TR.radiance_loss([(comp, trans, tb)], pix, 0.025, 0.5)
    envc = TR.bkgd_mlp(th["bkgd_mlp"], env_enc).reshape(32, 32, 3)
    smooth = (0.5 * ((envc[1:, :] - envc[:-1, :]) ** 2).reshape(-1) + 0.5 * ((envc[:, 1:] - envc[:, :-1]) ** 2).reshape(-1)).mean()
    (total + smooth).backward()
    torch_losses.append(parts["loss"].detach())
    lr = learning_rate_decay(step, 1e-3, 1e-4, steps, 0, 0.01)
    with torch.no_grad():
        k1 = step + 1
        for k in names:
            g = th[k].grad
            mu[k].mul_(0.9).add_(g, alpha=0.1); nu[k].mul_(0.999).addcmul_(g, g, value=0.001)
            th[k].addcdiv_(mu[k] / (1 - 0.9 ** k1), (nu[k] / (1 - 0.999 ** k1)).sqrt_().add_(1e-8), value=-lr)
            th[k].grad = None
curves["torch_fp32"] = [float(x) for x in torch.stack(torch_losses).cpu()]
print(f"[torch fp32 autograd on the GPU] {steps} steps in {time.perf_counter() - t0:.1f} s; loss {curves['torch_fp32'][0]:.5f} -> {curves['torch_fp32'][-1]:.6f}")
ref = np.array(curves["torch_fp32"])
summary = {}
for k in ("f32", "tf32", "bf16"):
    c = np.array(curves[k])
    rel = np.abs(c - ref) / ref
    summary[k] = {"final_loss": float(c[-1]), "mean_last_100": float(c[-100:].mean()), "max_rel_dev_first_100_steps": float(rel[:100].max()),
                  "median_rel_dev_all_steps": float(np.median(rel))}
summary["torch_fp32"] = {"final_loss": float(ref[-1]), "mean_last_100": float(ref[-100:].mean())}
for k, v in summary.items():
    print(k, v)
os.makedirs(os.path.dirname(out_path) or ".", exist_ok=True)
json.dump({"note": __doc__.strip().split("\n\n")[1], "steps": steps, "config": {"rays": B, "samples": S, "P": P, "grid": G, "pool": POOL},
           "summary": summary, "every_10th_step": {k: v[::10] for k, v in curves.items()}}, open(out_path, "w"), indent=0)
print("wrote", out_path)
