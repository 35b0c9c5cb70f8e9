#!/usr/bin/env python3
"""End-to-end RGB error and kernel time of a two-pass NerfMLP forward against the shipped three-pass f16x3 forward.

    python tools/r02/fwd_passes.py gen out.npz              # precision f16x3: weight sets + their RGB / depth
    PREC=f16x2 python tools/r02/fwd_passes.py cmp out.npz    # the same weights through another forward precision, max |delta|

(profiles/r02/fwd_two_pass_error.jsonl was taken with two variant BUILDS of the kernel: "22" = exact weights x f16(activations), which
became the precision f16x2, and "2" = f16(weights) x exact activations, which was dropped.)
Workload = bench.py's default (ship_straight: 4096 rays x 128 samples, flat).  Weight sets:
  init     bench.py's initial weights
  trained  the student of a 1000-step teacher/student run from those (train_step, f32 backward)
  teacher  the teacher itself (biases ~ 0.3: sharper densities)
  stress   the teacher with every hidden weight matrix x 1.5 (pre-activations ~ 1.5^8 larger spread: worst case for a rounded operand)
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import bench
from samplenerfro_amd import _lib, models, ops, prng, synthetic as syn, utils as U
from samplenerfro_amd.utils import Rays

dev = torch.device("cuda:0")
mode, path = sys.argv[1], sys.argv[2]
steps = int(os.environ.get("STEPS", "1000"))
cfg = dict(syn.CONFIGS["ship_straight"])
B = 4096
PREC = "f16x3" if mode == "gen" else os.environ.get("PREC", "f16x2")
model, variables, pf = bench.build_scene(cfg, dev, PREC, 0)
o, d = syn.sphere_rays(B, seed=syn.SEED + 7)
rays = Rays(torch.from_numpy(o).to(dev), None, torch.from_numpy(d).to(dev), None)
key = np.array([0, 1], np.uint32)
jitter = np.arange(0, cfg["S"] * cfg["P"], cfg["P"]) + cfg["P"] // 2


def render(flat):
    v = models.make_variables({k: torch.from_numpy(x).to(dev) for k, x in flat.items()})
    with torch.no_grad():
        ret, _ = model.apply(v, key, key, rays, False, jitter=jitter)
    return ret[-1][0].cpu().numpy(), ret[-1][1].cpu().numpy()


def fwd_ms():
    pd, dr, _, _ = ops.march(model.table, model.spec, rays.origins, rays.viewdirs, model.near, model.far, cfg["S"] * cfg["P"])
    jit = torch.from_numpy(jitter.astype(np.int32)).to(dev)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    pid = _lib.PRECISIONS[PREC]
    packed = ops.nerfmlp_pack(variables["flat"]["coarse_mlp"], pid)
    for it in range(3):
        if it == 1:
            ev[0].record()
        for _ in range(10):
            ops.nerfmlp_forward(packed, pid, pd, dr, jit, cfg["S"], B)
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / 20


if mode == "gen":
    from samplenerfro_amd.train import TrainState, train_step
    teacher = syn.init_params_flat(123, fine=False, bias_scale=0.3)
    flags = U.default_flags(num_coarse_samples=cfg["S"], num_fine_samples=0, num_path_samples=cfg["P"], white_bkgd=False, bg_weight=0.025,
                            bg_smooth_weight=1.0, bg_patch_size=32, use_online_sparsity=False, randomized=True, near=cfg["near"], far=cfg["far"],
                            lr_init=1e-3, lr_final=1e-4, lr_delay_steps=0, max_steps=steps, backward_precision="f32")
    tv = models.make_variables({k: torch.from_numpy(v).to(dev) for k, v in teacher.items()})
    gen = np.random.default_rng(0)
    ev = gen.standard_normal((32, 32, 3)).astype(np.float32); ev /= np.linalg.norm(ev, axis=-1, keepdims=True)
    env = Rays(None, None, torch.from_numpy(ev).to(dev), None)
    pool = []
    for i in range(8):
        oo, dd = syn.sphere_rays(B, seed=2000 + i)
        rr = Rays(torch.from_numpy(oo).to(dev), None, torch.from_numpy(dd).to(dev), None)
        with torch.no_grad():
            pix = model.apply(tv, key, key, rr, False, jitter=jitter)[0][-1][0].clone()
        pool.append((rr, pix))
    state = TrainState.create(model, variables, flags)
    rng = prng.PRNGKey(5)
    first = last = None
    for s in range(steps):
        rr, pix = pool[s % 8]
        state, stats, rng = train_step(model, rng, state, {"rays": rr, "pixels": pix, "annealed_alpha": 0.5, "env_rays": env}, flags, jitter=jitter)
        if s == 0:
            first = float(stats.loss)
    last = float(stats.loss)
    print(f"trained {steps} steps: loss {first:.5f} -> {last:.6f}")
    trained = {k: v.detach().cpu().numpy().copy() for k, v in state.variables["flat"].items()} if hasattr(state, "variables") else None
    if trained is None:
        trained = {k: v.detach().cpu().numpy().copy() for k, v in state.params["flat"].items()}
    stress = {k: v.copy() for k, v in teacher.items()}
    off = 0
    w = stress["coarse_mlp"]
    for (fi, fo) in models.NERF_MLP_SHAPES if hasattr(models, "NERF_MLP_SHAPES") else []:
        w[off:off + fi * fo] *= 1.5 if fo == 256 else 1.0
        off += fi * fo + fo
    sets = {"init": pf, "trained": trained, "teacher": teacher, "stress": stress}
    out = {}
    for name, flat in sets.items():
        rgb, dist = render(flat)
        out[f"{name}/rgb"], out[f"{name}/dist"] = rgb, dist
        for k, v in flat.items():
            out[f"{name}/w/{k}"] = v
    np.savez(path, **out)
    print("three-pass forward kernel ms:", round(fwd_ms(), 4))
else:
    z = np.load(path)
    res = {"precision": PREC, "fwd_kernel_ms": round(fwd_ms(), 4)}
    for name in ("init", "trained", "teacher", "stress"):
        flat = {k.split("/w/")[1]: z[k] for k in z.files if k.startswith(name + "/w/")}
        rgb, dist = render(flat)
        res[name] = {"max_abs_rgb": float(np.abs(rgb - z[f"{name}/rgb"]).max()), "rms_rgb": float(np.sqrt(((rgb - z[f"{name}/rgb"]) ** 2).mean())),
                     "max_abs_dist": float(np.abs(dist - z[f"{name}/dist"]).max())}
    print(json.dumps(res))
