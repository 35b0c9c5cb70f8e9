#!/usr/bin/env python3
"""How exposed is the "bit-exact integer index" contract to XLA's x / const -> x * (1 / const) rewrite?

The oracle (and the HIP march, which is bit-identical to it) forms the grid coordinate as the IEEE quotient (p - nmin) / ndelta, like the
reference source (rnerf/ior_utils.py:189-191).  XLA is allowed to turn a division by a compile-time constant into a multiplication by the
rounded reciprocal; if jax 0.2.22's XLA did that, a fraction of the coordinates would differ by one ulp and floor() would flip for the
ones within an ulp of a cell face.  No JAX is available to observe which form the reference runs; this tool MEASURES what would change:
it evaluates BASELINE configs 1 (example, full size), 3 (ship refractive, oracle-sized grid) and 5 (glass flat) twice with the numpy
oracle — division vs reciprocal form (oracle.ref_np.CONST_DIV_AS_RECIPROCAL) — and reports, per config, the fraction of (ray, node) voxel
index 6-tuples that differ, the fraction of resample node indices that differ, the largest position / depth difference along the paths and
the effect on the rendered RGB.

usage: python tools/xla_rcp_exposure.py [out.json]        (CPU only, ~2 minutes)
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from oracle import ref_np as R                       # noqa: E402
from samplenerfro_amd import synthetic as syn        # noqa: E402


def run_case(c, reciprocal):
    R.CONST_DIV_AS_RECIPROCAL = bool(reciprocal)
    try:
        table = R.build_table(c["grid"], c["ndim"], c["nmin"], c["nmax"])
        N = c["S"] * c["P"]
        out = R.path_sampler(c["origins"], c["viewdirs"], table, c["ndim"], c["nmin"], c["nmax"], c["near"], c["far"], N, np.float32, return_idx=True)
        cfg = R.ModelConfig(c["ndim"], c["nmin"], c["nmax"], near=c["near"], far=c["far"], num_coarse_samples=c["S"], num_fine_samples=c["F"],
                            num_path_samples=c["P"])
        taps = {}
        ret, _ = R.nerf_forward(cfg, syn.params_tree(c["params"]), table, c["origins"], c["viewdirs"], c["jitter"], taps=taps)
        return dict(pos=out[0], dist=out[2], vox=out[5], rgb=ret[-1][0], depth=ret[-1][1], idx_f=taps.get("idx_f"))
    finally:
        R.CONST_DIV_AS_RECIPROCAL = False


def compare(name, c):
    t0 = time.time()
    a, b = run_case(c, False), run_case(c, True)
    vox_diff = np.any(a["vox"] != b["vox"], axis=-1)
    res = {"config": name, "rays": int(a["pos"].shape[0]), "nodes_per_ray": int(a["pos"].shape[1]),
           "voxel_index_tuples_changed": int(vox_diff.sum()), "voxel_index_fraction_changed": float(vox_diff.mean()),
           "rays_with_any_voxel_index_change": int(vox_diff.any(axis=1).sum()),
           "max_abs_position_diff": float(np.abs(a["pos"] - b["pos"]).max()), "max_abs_depth_diff_along_path": float(np.abs(a["dist"] - b["dist"]).max()),
           "max_abs_rgb_diff": float(np.abs(a["rgb"] - b["rgb"]).max()), "max_abs_rendered_depth_diff": float(np.abs(a["depth"] - b["depth"]).max())}
    if a["idx_f"] is not None:
        d = a["idx_f"] != b["idx_f"]
        res.update(resample_indices_changed=int(d.sum()), resample_index_fraction_changed=float(d.mean()))
    res["seconds"] = round(time.time() - t0, 1)
    return res


def ship_refractive_small(G=128, B=256):
    """BASELINE config 3 at an oracle-sized grid: sphere of radius 0.6, ri 0.5, prefilter (9, 3.0), 128 samples x P = 12, flat."""
    cfg = syn.CONFIGS["ship_refractive"]
    grid = R.conv3d_normal(syn.scale_ior(syn.sphere_grid(G, cfg["extent"], cfg["radius"]), cfg["ri"]).reshape(-1, 1), [G] * 3, cfg["ksize"],
                           cfg["ksigma"]).reshape(G, G, G).astype(np.float32)
    o, d = syn.sphere_rays(B, seed=syn.SEED)
    S, P = cfg["S"], cfg["P"]
    return dict(grid=grid, ndim=[G] * 3, nmin=[-cfg["extent"]] * 3, nmax=[cfg["extent"]] * 3, origins=o, viewdirs=d, S=S, F=0, P=P,
                near=cfg["near"], far=cfg["far"], jitter=np.arange(0, S * P, P) + P // 2, params=syn.init_params_flat(0, fine=False))


def main():
    import cases
    results = [compare("1 example_full (512 rays, 64 + 128, N = 768, the OBJ grid 128^3)", cases.inputs_example()),
               compare("3 ship_refractive (256 rays, 128 flat, N = 1536, sphere grid 128^3 after (9, 3.0))", ship_refractive_small()),
               compare("5 glass_flat (128 rays, 256 flat, N = 6144, 96^3 anisotropic bbox)", cases.inputs_glass("flat"))]
    out = {"what": "oracle with (p - nmin) / ndelta [reference source form] vs (p - nmin) * RN(1 / ndelta) [what XLA may emit]; both in float32",
           "results": results}
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
