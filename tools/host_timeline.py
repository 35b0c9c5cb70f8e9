#!/usr/bin/env python3
"""Host-side issue timeline of one train step (which call blocks the host?): wraps every ops.* entry and prints when it was entered
and how long the host spent inside.  No device synchronisation is added."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from samplenerfro_amd import ops, synthetic as syn, utils as U, prng, distributed as D
from samplenerfro_amd.train import TrainState, train_step
from samplenerfro_amd.utils import Rays

dev = torch.device("cuda:0")
cfg = dict(syn.CONFIGS["ship_straight"]); B = 4096
model, variables, pf = bench.build_scene(cfg, dev, "f16x3", 0)
o, d = syn.sphere_rays(B)
rays = Rays(torch.from_numpy(o).to(dev), None, torch.from_numpy(d).to(dev), None)
flags = U.default_flags(num_coarse_samples=cfg["S"], num_fine_samples=0, num_path_samples=cfg["P"], white_bkgd=False, bg_weight=0.025,
                        bg_smooth_weight=1.0, bg_patch_size=128, use_online_sparsity=False, randomized=True, near=cfg["near"], far=cfg["far"])
state = TrainState.create(model, variables, flags)
gen = np.random.default_rng(0)
ev = gen.standard_normal((128, 128, 3)).astype(np.float32); ev /= np.linalg.norm(ev, axis=-1, keepdims=True)
batch = {"rays": rays, "pixels": torch.rand((B, 3), device=dev), "annealed_alpha": 0.5, "env_rays": Rays(None, None, torch.from_numpy(ev).to(dev), None)}
rng = prng.PRNGKey(1)
log = []
def wrap(mod, name):
    f = getattr(mod, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); log.append((name, t, time.perf_counter() - t)); return r
    setattr(mod, name, g)
for n in dir(ops):
    if callable(getattr(ops, n)) and not n.startswith("_") and n not in ("check", "ptr", "current_stream", "Optional", "Grid"):
        wrap(ops, n)
wrap(prng, "split"); wrap(prng, "randint")
if os.environ.get("SYNC_DEBUG"):
    torch.cuda.set_sync_debug_mode("warn")
for i in range(6):
    log.clear()
    t0 = time.perf_counter()
    state, stats, rng = train_step(model, rng, state, batch)
    t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host issue time of the last step: {(t1-t0)*1e3:.2f} ms; drain after it: {(t2-t1)*1e3:.2f} ms")
for n, t, dt in log:
    print(f"  +{(t-t0)*1e3:7.3f} ms  {dt*1e3:7.3f} ms  {n}")
