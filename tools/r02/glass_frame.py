#!/usr/bin/env python3
"""ms per 800x800 frame of BASELINE configs[4] (glass: 256 samples/ray, P = 24 -> 6144 eikonal steps, 384^3 grid) on ONE GPU
(the config shards the rays over 8; distributed.render_image_sharded is the 8-GPU form).  python tools/r02/glass_frame.py [precision]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from samplenerfro_amd import ops, prng, synthetic as syn, utils as U
from samplenerfro_amd.utils import Rays
dev = torch.device("cuda:0")
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
cfg = dict(syn.CONFIGS["glass_frame"])
model, variables, pf = bench.build_scene(cfg, dev, prec, 0)
H = W = 800
focal = 0.5 * W / np.tan(0.5 * 0.6911112070083618)
c2w = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0]], np.float32)
o_w, _, v_w = ops.generate_rays(c2w, H, W, dev, focal=focal)
fr = Rays(o_w, None, v_w, None)
key = prng.PRNGKey(0)
fn = lambda k0, k1, r, path=None: model.apply(variables, k0, k1, r, False, path=path)
chunk = 16384
U.render_image(fn, fr, key, False, chunk=chunk); torch.cuda.synchronize()
t = time.perf_counter()
rgb, _, _ = U.render_image(fn, fr, key, False, chunk=chunk); torch.cuda.synchronize()
print(f"glass_frame {prec}: {1e3 * (time.perf_counter() - t):.1f} ms per 800x800x{cfg['S']} frame on 1 GPU (chunk {chunk}), finite={bool(torch.isfinite(rgb).all())}")
