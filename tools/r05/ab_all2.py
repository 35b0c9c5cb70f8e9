"""A/B of the stage-all* step: the next batch's shell-order pre-march on the tail stream (train._ALL_PREFETCH_ON_TAIL) and the NerfMLP input
gradients on a stream of their own beside the pair chain (train._ALL_INPUT_GRAD_OWN_STREAM) — one process, alternating, same model and batch.
usage (GPU box): python tools/r05/ab_all2.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from samplenerfro_amd import synthetic as syn, prng, train, distributed as D
from samplenerfro_amd.utils import Rays
dev = torch.device("cuda:0")
cfg = dict(syn.CONFIGS["ship_refractive"])
model, variables, pf = bench.build_scene(cfg, dev, "f16x3", 0, "all", None)
o, d = syn.sphere_rays(4096, seed=syn.SEED)
rays = Rays(torch.from_numpy(o).to(dev), None, torch.from_numpy(d).to(dev), None)
key = prng.PRNGKey(syn.SEED)
class A: pass
args = A(); args.reserve_cus = 0
def barrier(): torch.cuda.synchronize()
for rep in range(3):
    for pre, own in ((False, False), (True, False), (False, True), (True, True)):
        train._ALL_PREFETCH_ON_TAIL, train._ALL_INPUT_GRAD_OWN_STREAM = pre, own
        st = bench.Stepper(args, cfg, model, variables, rays, key, 4096, 1, 0, 0, dev, "f16x3", "train", "all", False, False)
        dt = bench.timed_steps(st, 3, 20, barrier, D, dev)
        st.close()
        print(f"prefetch on tail = {pre}, input grads on own stream = {own}: {1e3 * dt / 20:.3f} ms per step", flush=True)
