"""ms per 800 x 800 x 128 frame: strict sequence against the software pipeline of render_image(model=...) — march of chunk k + 1 beside the MLP
of chunk k — with (a) the MLP grid capped (reserve_cus) and (b) CU-masked streams (hipExtStreamCreateWithCUMask: the march's stream owns
`n` CUs, the main stream the others).  usage (GPU box): python tools/r06/frame_pipe.py [eval_precision]"""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from samplenerfro_amd import synthetic as syn, prng, ops, utils as U, models as M
from samplenerfro_amd.utils import Rays
dev = torch.device("cuda:0")
evalp = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
cfg = dict(syn.CONFIGS["ship_straight"])
model, variables, pf = bench.build_scene(cfg, dev, "f16x3", 0, "radiance", None if evalp == "f16x3" else evalp)
H = W = 800
focal = 0.5 * W / np.tan(0.5 * 0.6911112070083618)
c2w = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0]], np.float32)
o_w, _, v_w = ops.generate_rays(c2w, H, W, dev, focal=focal)
fr = Rays(o_w, None, v_w, None)
key = prng.PRNGKey(syn.SEED)
fn = lambda k0, k1, r, path=None: model.apply(variables, k0, k1, r, False, path=path)


def frame(chunk, pipelined, reps=3):
    m = model if pipelined else None
    U.render_image(fn, fr, key, False, chunk=chunk, model=m)
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps):
        rgb, _, _ = U.render_image(fn, fr, key, False, chunk=chunk, model=m)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t) / reps, rgb


ref = None
for chunk in (8192, 32768, 65536):
    model.release_reserved_cus()
    ms, rgb = frame(chunk, False)
    if ref is None: ref = rgb
    print(f"chunk {chunk:6d} strict sequence: {ms:7.1f} ms  same bits {bool(torch.equal(rgb, ref))}", flush=True)
    for res in (16, 32, 48, 64):
        orig = model.prefetch_path
        model.prefetch_path = lambda rays, sync_inputs=True, reserve_cus=res, _o=orig: _o(rays, sync_inputs=sync_inputs, reserve_cus=reserve_cus)
        ms, rgb = frame(chunk, True)
        model.prefetch_path = orig
        print(f"chunk {chunk:6d} pipelined, MLP grid capped at 256 - {res}: {ms:7.1f} ms  same bits {bool(torch.equal(rgb, ref))}", flush=True)
    model.release_reserved_cus()

# ---- CU-masked streams
hip = C.CDLL("libamdhip64.so")
def masked_stream(lo, hi, total=256):
    words = (total + 31) // 32
    mask = (C.c_uint32 * words)()
    for b in range(lo, hi):
        mask[b // 32] |= 1 << (b % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), C.c_uint32(words), mask)
    if rc != 0:
        raise RuntimeError(f"hipExtStreamCreateWithCUMask -> {rc}")
    return torch.cuda.ExternalStream(s.value, device=dev)

try:
    for res in (16, 24, 32, 48):
        main_s, side_s = masked_stream(0, 256 - res), masked_stream(256 - res, 256)
        M._STREAMS[(str(dev), "march")] = side_s
        model._side = side_s
        for chunk in (32768, 65536):
            with torch.cuda.stream(main_s):
                orig = model.prefetch_path
                model.prefetch_path = lambda rays, sync_inputs=True, reserve_cus=res, _o=orig: _o(rays, sync_inputs=sync_inputs, reserve_cus=reserve_cus)
                ms, rgb = frame(chunk, True)
                model.prefetch_path = orig
            print(f"chunk {chunk:6d} pipelined, CU masks {256 - res} | {res}: {ms:7.1f} ms  same bits {bool(torch.equal(rgb, ref))}", flush=True)
except Exception as e:
    print("CU-masked streams failed:", repr(e))
