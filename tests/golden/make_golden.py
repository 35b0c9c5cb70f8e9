#!/usr/bin/env python3
"""Regenerates tests/golden/*.npz:  python tests/golden/make_golden.py [small] [obj] [cases | example_full dolphin_train glass_flat glass_hier]

These vectors come from THIS repository's oracle (oracle/ref_np.py), not from the reference implementation: the reference needs
jax/flax, which cannot be imported offline, and ships no fixtures for the path — parity therefore stays
"unpinned" in the sense of the brief.  What the fixtures do pin is the oracle itself (a change of its arithmetic shows up as a
diff against committed numbers) and, on the GPU box, the HIP path against numbers that do not depend on the oracle code at test time.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import ref_np as R                       # noqa: E402
from samplenerfro_amd import synthetic as syn        # noqa: E402


def small_case():
    G, ext, B, S, F, P = 16, 1.5, 48, 8, 12, 4
    ndim, nmin, nmax = [G] * 3, [-ext] * 3, [ext] * 3
    grid = R.conv3d_normal(syn.scale_ior(syn.sphere_grid(G, ext, 0.6), 0.5).reshape(-1, 1), ndim, 3, 1.0).reshape(ndim)
    o, d = syn.sphere_rays(B, seed=77)
    jitter = (np.arange(0, S * P, P) + np.random.default_rng(77).integers(0, P, S)).astype(np.int32)
    # the 1.25 M network weights are NOT stored: they are numpy's PCG64 stream for `param_seed` (synthetic.init_params_flat), whose
    # first values are pinned below
    pf = params_of(77)
    return dict(G=G, ext=ext, B=B, S=S, F=F, P=P, grid=grid.astype(np.float32), origins=o, viewdirs=d, jitter=jitter, param_seed=77,
                param_probe=np.concatenate([pf[k][:8] for k in sorted(pf)]))


def params_of(seed):
    return syn.init_params_flat(int(seed), fine=True, bias_scale=0.05)


def run_oracle(c):
    G, ext = int(c["G"]), float(c["ext"])
    ndim, nmin, nmax = [G] * 3, [-ext] * 3, [ext] * 3
    table = R.build_table(c["grid"], ndim, nmin, nmax)
    cfg = R.ModelConfig(ndim, nmin, nmax, num_coarse_samples=int(c["S"]), num_fine_samples=int(c["F"]), num_path_samples=int(c["P"]))
    pf = params_of(c["param_seed"])
    assert np.array_equal(np.concatenate([pf[k][:8] for k in sorted(pf)]), c["param_probe"]), "numpy RNG stream changed"
    taps = {}
    ret, _ = R.nerf_forward(cfg, syn.params_tree(pf), table, c["origins"], c["viewdirs"], c["jitter"], taps=taps)
    out = {}
    for lvl, name in enumerate(("coarse", "fine")):
        rgb, dist, acc, trans, tb = ret[lvl]
        out.update({f"{name}_rgb": rgb, f"{name}_dist": dist, f"{name}_acc": acc, f"{name}_trans": trans, f"{name}_trans_bkgd": tb})
    out["ray_pos"] = taps["ray_pos"]; out["ray_dist"] = taps["ray_dist"]
    out["table"] = table
    return out


def make_example_obj(ref_obj="/root/reference/example_data/voxelize/mesh_4_128_1.5_1.165.obj"):
    """example_obj.npz: the one real artefact of the missing example grid that the reference ships (the marching-cubes OBJ written by
    voxelize_mesh.py:134-135), stored as DATA (vertex / face arrays), plus the oracle voxeliser's per-voxel inside counts for it.
    Needs the reference checkout (this container only); the GPU box uses the committed file."""
    from samplenerfro_amd.voxelize import load_obj
    import cases
    verts, faces = load_obj(ref_obj)
    counts = R.voxelize_counts(cases.example_obj_world(verts), faces, 128, [-1.5] * 3, [1.5] * 3, 4)
    np.savez_compressed(os.path.join(HERE, "example_obj.npz"), verts=verts, faces=faces.astype(np.int32), counts=counts.astype(np.uint8))
    print("wrote example_obj.npz", verts.shape, faces.shape, "occupied fraction", float((counts > 0).mean()))


def make_config_cases(names=None):
    import cases
    for name, (make_inputs, run) in cases.CASES.items():
        if names and name not in names:
            continue
        out = run(make_inputs())
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print("wrote", name + ".npz", {k: getattr(v, "shape", v) for k, v in list(out.items())[:4]})


if __name__ == "__main__":
    what = sys.argv[1:] or ["small", "obj", "cases"]
    if "small" in what:
        c = small_case()
        o = run_oracle(c)
        np.savez_compressed(os.path.join(HERE, "example_small.npz"), **c, **{f"out_{k}": v for k, v in o.items() if k != "table"})
        print("wrote", os.path.join(HERE, "example_small.npz"))
    if "obj" in what:
        make_example_obj()
    if "cases" in what or any(w in ("example_full", "dolphin_train", "glass_flat", "glass_hier") for w in what):
        make_config_cases([w for w in what if w not in ("small", "obj", "cases")])
