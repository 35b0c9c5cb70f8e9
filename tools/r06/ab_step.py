#!/usr/bin/env python3
"""The default train step (bench's Stepper: march of the next batch co-resident with the wgrad) for every library given, alternating on ONE box,
one process per library: python tools/r06/ab_step.py [lib.so ...] (no argument: the product library).  [--backward f16x3lo8]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    from samplenerfro_amd import _lib
    _lib.load(sys.argv[2] if sys.argv[2] != "product" else None)
    import gc, time, types
    import numpy as np, torch
    import bench
    from samplenerfro_amd import synthetic as syn, prng
    from samplenerfro_amd.utils import Rays
    dev = torch.device("cuda:0")
    cfg = dict(syn.CONFIGS["ship_straight"])
    model, variables, pf = bench.build_scene(cfg, dev, "f16x3", 0, "radiance", None)
    o, d = syn.sphere_rays(4096, seed=syn.SEED)
    rays = Rays(torch.from_numpy(o).to(dev), None, torch.from_numpy(d).to(dev), None)
    st = bench.Stepper(types.SimpleNamespace(reserve_cus=32), cfg, model, variables, rays, prng.PRNGKey(syn.SEED), 4096, 1, 0, 0, dev, sys.argv[3], "train", "radiance", True, False)
    gc.collect(); gc.freeze()
    out = []
    for w in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(40): st.step()
        torch.cuda.synchronize(); out.append(1e3 * (time.perf_counter() - t) / 40)
    print(f"{os.path.basename(sys.argv[2]):28s} backward {sys.argv[3]}: " + " ".join(f"{x:.3f}" for x in out) + " ms per step (3 windows of 40)", flush=True)
    sys.exit(0)
args = sys.argv[1:]
bw = "f16x3"
if "--backward" in args:
    i = args.index("--backward"); bw = args[i + 1]; del args[i:i + 2]
libs = args or ["product"]
for rep in range(3):
    for l in libs:
        subprocess.run([sys.executable, os.path.abspath(__file__), "--one", l, bw], check=False)
