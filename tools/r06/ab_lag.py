"""What the lagged range retry costs per step, and which part of it: the bench's Stepper with range_retry = "lag" (with / without the per-step
clones of the batch tensors), True (decided in place) and False, alternating in ONE process.  python tools/r06/ab_lag.py"""
import gc, os, sys, time, types
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from samplenerfro_amd import synthetic as syn, prng, train as TRN
from samplenerfro_amd.utils import Rays
dev = torch.device("cuda:0")
cfg = dict(syn.CONFIGS["ship_straight"])
model, variables, pf = bench.build_scene(cfg, dev, "f16x3", 0, "radiance", None)
o, d = syn.sphere_rays(4096, seed=syn.SEED)
rays = Rays(torch.from_numpy(o).to(dev), None, torch.from_numpy(d).to(dev), None)
key = prng.PRNGKey(syn.SEED)
args = types.SimpleNamespace(reserve_cus=32)
orig_clone = TRN._clone_batch
gc.collect(); gc.freeze()
def run(mode, clones=True):
    TRN._clone_batch = orig_clone if clones else (lambda b: b)
    v = bench.models_fresh_variables(pf, dev)
    st = bench.Stepper(args, cfg, model, v, rays, key, 4096, 1, 0, 0, dev, "f16x3", "train", "radiance", True, False)
    st.flags.range_retry = mode
    for _ in range(5): st.step()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(40): st.step()
    torch.cuda.synchronize(); ms = 1e3 * (time.perf_counter() - t) / 40
    st.close()
    return ms
for rep in range(3):
    print("lag %.3f   lag without clones %.3f   in place %.3f   off %.3f" % (run("lag"), run("lag", False), run(True), run(False)), flush=True)
