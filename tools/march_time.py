"""Profiling helper (not part of the product): time the march kernel for several grid sizes."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from samplenerfro_amd import ops, _lib, synthetic as syn
dev = torch.device("cuda:0")
B, N = 4096, 1536
o, d = syn.sphere_rays(B)
o = torch.from_numpy(o).to(dev); d = torch.from_numpy(d).to(dev)
LAYOUTS = sys.argv[1:] or ["reference", "bricks"]
for G, layout in [(G, l) for G in (64, 256, 512) for l in LAYOUTS]:
    spec = _lib.Grid.make([G] * 3, [-1.5] * 3, [1.5] * 3, layout)
    grid = torch.ones((G, G, G), device=dev)
    table = ops.grid_build_table(grid, spec)
    del grid
    pd, dr, _, _ = ops.march(table, spec, o, d, 2.0, 6.0, N)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    ev[0].record()
    for i in range(5):
        ops.march(table, spec, o, d, 2.0, 6.0, N, out=(pd, dr))
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(5)]
    print(f"G={G} {layout:9s} march ms min/med = {min(ms):.3f}/{np.median(ms):.3f}  per-step us = {np.median(ms)*1e3/N:.3f}")
    del table
# the refractive scene of BASELINE configs[2] (glass sphere after the (9, 3.0) prefilter): the rays bend, the speculative gathers mispredict
for G, layout in [(G, l) for G in (256, 512) for l in LAYOUTS]:
    spec = _lib.Grid.make([G] * 3, [-1.5] * 3, [1.5] * 3, layout)
    grid = torch.from_numpy(syn.scale_ior(syn.sphere_grid(G, 1.5, 0.6), 0.5).astype(np.float32)).to(dev)
    grid = ops.grid_prefilter(grid, 9, 3.0)
    table = ops.grid_build_table(grid, spec)
    del grid
    pd, dr, _, _ = ops.march(table, spec, o, d, 2.0, 6.0, N)
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(6)]
    ev[0].record()
    for i in range(5):
        ops.march(table, spec, o, d, 2.0, 6.0, N, out=(pd, dr))
        ev[i + 1].record()
    torch.cuda.synchronize()
    ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(5)]
    print(f"refractive sphere G={G} {layout:9s} march ms min/med = {min(ms):.3f}/{np.median(ms):.3f}")
    del table
