#!/usr/bin/env python3
"""Teacher/student convergence check of the whole training stack at bench size (not a test: prints the PSNR curve).
A 'teacher' parameter set renders target pixels for random rays; a student with a different initialisation is trained on them."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from samplenerfro_amd import models, synthetic as syn, utils as U, prng
from samplenerfro_amd.train import TrainState, train_step
from samplenerfro_amd.utils import Rays

dev = torch.device("cuda:0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 400
G, B, S, P = 128, 4096, 64, 8
a = torch.linspace(-1.5, 1.5, G, dtype=torch.float64, device=dev)
r = torch.sqrt(a[:, None, None] ** 2 + a[None, :, None] ** 2 + a[None, None, :] ** 2)
grid = (1.0 + 0.5 * torch.clamp((0.6 - r) / (3.0 / (G - 1)) + 0.5, 0.0, 1.0)).float()
flags = U.default_flags(num_coarse_samples=S, num_fine_samples=0, num_path_samples=P, white_bkgd=False, bg_weight=0.025, bg_smooth_weight=float(os.environ.get("BG_SMOOTH", "1.0")),
                        bg_patch_size=128, use_online_sparsity=False, randomized=True, lr_init=1e-3, lr_final=1e-4, lr_delay_steps=0, max_steps=steps)
model, variables = models.construct_nerf(np.array([0, 1], np.uint32), None, flags, [G] * 3, [-1.5] * 3, [1.5] * 3, grid)
teacher = models.make_variables({k: torch.from_numpy(v).to(dev) for k, v in syn.init_params_flat(123, fine=False, bias_scale=0.3).items()})
state = TrainState.create(model, variables, flags)
gen = np.random.default_rng(0)
ev = gen.standard_normal((128, 128, 3)).astype(np.float32); ev /= np.linalg.norm(ev, axis=-1, keepdims=True)
env = Rays(None, None, torch.from_numpy(ev).to(dev), None)
rng = prng.PRNGKey(5)
key = np.array([9, 9], np.uint32)
fixed = (np.arange(0, S * P, P) + P // 2) if os.environ.get("FIXED_JITTER") else None     # same quadrature nodes for teacher and student
t0 = time.perf_counter()
for step in range(steps):
    o, d = syn.sphere_rays(B, seed=1000 + step % 64)
    rays = Rays(torch.from_numpy(o).to(dev), None, torch.from_numpy(d).to(dev), None)
    with torch.no_grad():
        pix = model.apply(teacher, key, key, rays, False, jitter=fixed)[0][-1][0].clone()
    state, stats, rng = train_step(model, rng, state, {"rays": rays, "pixels": pix, "annealed_alpha": 0.5, "env_rays": env}, jitter=fixed)
    if step % 50 == 0 or step == steps - 1:
        print(f"step {step:4d}  loss {float(stats.loss):.6f}  psnr {float(stats.psnr):6.2f} dB  loss_bg {float(stats.loss_bg):.5f}  weight_l2 {float(stats.weight_l2):.5f}")
torch.cuda.synchronize()
print(f"{steps} steps in {time.perf_counter() - t0:.1f} s (incl. teacher renders and host ray generation)")
