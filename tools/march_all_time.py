"""Profiling helper (not part of the product): the stage-`all*` march (rnerf_march_all) at bench size, and the statistics that bound it:
how many (ray, node) pairs lie in the boundary shell, and how many (wave, node) MLP evaluations a W-ray wave needs in the shell-coherent
ray order (a wave evaluates so3_mlp at a node whenever ANY of its rays is in the shell there).

usage: python tools/march_all_time.py [workload] [rays]       (default ship_refractive 4096)
"""
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from samplenerfro_amd import ops, models, synthetic as syn

wl = sys.argv[1] if len(sys.argv) > 1 else "ship_refractive"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = torch.device("cuda:0")
cfg = dict(syn.CONFIGS[wl])
model, variables, pf = bench.build_scene(cfg, dev, "f16x3", 0, stage="all")
o, d = syn.sphere_rays(B)
o = torch.from_numpy(o).to(dev); d = torch.from_numpy(d).to(dev)
N = cfg["S"] * cfg["P"]
so3 = model._flat(variables, "so3_mlp", models.SO3_MLP_SHAPES).detach()


def timed(fn, n=5):
    fn(); torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    ev[0].record()
    for i in range(n):
        fn(); ev[i + 1].record()
    torch.cuda.synchronize()
    return float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(n)]))


# shell statistics from the plain march (the paths differ from the all* paths only by the so3 rotation, ~1e-5 at initialisation)
_, _, ior, _ = ops.march(model.table, model.spec, o, d, cfg["near"], cfg["far"], N, want_ior=True)
g = ior[..., 1:4]
m = (g * g).sum(-1) > 1e-6            # [N, B]
pairs = int(m.sum())
print(f"{wl}: {B} rays x {N} nodes, shell pairs {pairs} = {pairs / (B * N):.3%} of the nodes; rays touching the shell {int(m.any(0).sum())}")
perm = ops._shell_order(model.table, model.spec, o, d, cfg["near"], cfg["far"], N).long()     # from the COARSE pre-march (N / 8 nodes)
for name, mm in (("given order", m), ("shell order", m[:, perm])):
    for W in (64, 32, 16, 8):
        ev = int(mm.reshape(N, B // W, W).any(-1).sum())
        print(f"  {name}: {W:2d}-ray groups: {ev} group evaluations ({ev * W / max(pairs, 1):.2f} x the pairs), worst group {int(mm.reshape(N, B // W, W).any(-1).sum(0).max())} of {N} nodes")

t_plain = timed(lambda: ops.march(model.table, model.spec, o, d, cfg["near"], cfg["far"], N))
t_given = timed(lambda: ops.march_all(model.table, model.spec, so3, o, d, cfg["near"], cfg["far"], N, 1.0, False, False))
t_cached = timed(lambda: ops.march_all(model.table, model.spec, so3, o, d, cfg["near"], cfg["far"], N, 1.0, False, True))


def fresh():
    ops._SHELL_CACHE.clear()
    return ops.march_all(model.table, model.spec, so3, o, d, cfg["near"], cfg["far"], N, 1.0, False, True)


t_fresh = timed(fresh)
print(f"plain march {t_plain:.3f} ms; all* march: given ray order {t_given:.3f} ms, shell order handed to the kernel {t_cached:.3f} ms (order cached), "
      f"{t_fresh:.3f} ms with the coarse pre-march + sort of a new batch")
