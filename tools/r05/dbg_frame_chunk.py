"""ms per 800 x 800 x 128 frame against the chunk size of render_image (the march's occupancy grows with the chunk; the path record of a chunk
is chunk x 1536 x 32 B).  usage (GPU box): python tools/r05/dbg_frame_chunk.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from samplenerfro_amd import synthetic as syn, prng, ops, utils as U
from samplenerfro_amd.utils import Rays
dev = torch.device("cuda:0")
cfg = dict(syn.CONFIGS["ship_straight"])
model, variables, pf = bench.build_scene(cfg, dev, "f16x3", 0, "radiance", None)
H = W = 800
focal = 0.5 * W / np.tan(0.5 * 0.6911112070083618)
c2w = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0]], np.float32)
o_w, _, v_w = ops.generate_rays(c2w, H, W, dev, focal=focal)
fr = Rays(o_w, None, v_w, None)
key = prng.PRNGKey(syn.SEED)
fn = lambda k0, k1, r, path=None: model.apply(variables, k0, k1, r, False, path=path)
ref = None
for chunk in (8192, 32768, 65536, 131072, 320000, 640000):
    U.render_image(fn, fr, key, False, chunk=chunk)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(3):
        rgb, _, _ = U.render_image(fn, fr, key, False, chunk=chunk)
    torch.cuda.synchronize(); ms = 1e3 * (time.perf_counter() - t) / 3
    same = True if ref is None else bool(torch.equal(rgb, ref))
    ref = rgb if ref is None else ref
    print(f"chunk {chunk:7d}: {ms:7.1f} ms per frame, peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB, same bits as the first: {same}", flush=True)
    model._ws.clear(); torch.cuda.empty_cache()
