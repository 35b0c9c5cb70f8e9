"""Golden vectors computed BY THE REFERENCE: the four pure-numpy methods it ships on this path.

rnerf/datasets.py cannot be imported here (its module-level `import jax` fails offline), but four of its methods use numpy only:
  Dataset._generate_rays (:216-242)   Blender pinhole rays          -> SURVEY 8f N4 (rnerf_generate_rays, focal form)
  OpenCV._generate_rays  (:486-518)   OpenCV pinhole rays           -> SURVEY 8f N4 (cam_mat form)
  Grid._linear3          (:278-313)   trilinear lookup, clamp to edge -> row G3 (the same index arithmetic as ior_utils' jax _linear3)
  Grid._compute_grad     (:315-322)   central-difference gradients  -> row G2
  Dataset._next_train    (:151-205)   the training batch sampler (numpy's global generator + fancy indexing of images / rays)
                                      -> SURVEY 8f N4, second half (rnerf_sample_batch, datasets.DeviceBatcher); it calls
                                      utils.namedtuple_map (rnerf/utils.py:70-72, pure Python), read out of the reference the same way
This script reads those FunctionDefs out of the reference's source with `ast` (nothing is imported from the reference, nothing of it is
copied into this repository), compiles each one as it stands, and calls it on seeded inputs with a plain attribute holder as `self` and a
namedtuple with the reference's field names as `utils.Rays`.  Inputs and outputs go to tests/golden/reference_numpy.npz — data, not source.
tests/test_reference_numpy_pin.py holds the oracle and (on the GPU box) the HIP entry points to them, and re-runs this script wherever
/root/reference exists to check that the committed file is still what the reference computes.

usage: python tests/golden/make_from_reference_numpy.py [out.npz]"""
import ast
import collections
import hashlib
import os
import sys
import types

import numpy as np

REF = os.environ.get("RNERF_REFERENCE_ROOT", "/root/reference")
SRC = os.path.join(REF, "rnerf", "datasets.py")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_numpy.npz")
SRC_UTILS = os.path.join(REF, "rnerf", "utils.py")
WANTED = {("Dataset", "_generate_rays"), ("OpenCV", "_generate_rays"), ("Grid", "_linear3"), ("Grid", "_compute_grad"), ("Dataset", "_next_train")}
Rays = collections.namedtuple("Rays", ("origins", "directions", "viewdirs", "radii"))        # field names of rnerf/utils.py's Rays


# what the four methods may name besides `np` and `utils`: nothing that opens files, imports, or evaluates text
SAFE_BUILTINS = {k: __builtins__[k] if isinstance(__builtins__, dict) else getattr(__builtins__, k)
                 for k in ("range", "len", "int", "float", "min", "max", "abs", "list", "tuple", "zip", "enumerate", "isinstance", "print", "sum")}


def _numpy_only_import(name, *args, **kwargs):
    """numpy's C code imports its own submodules lazily through the calling frame's builtins: allow exactly that."""
    if name.split(".")[0] != "numpy":
        raise ImportError(f"the reference's methods may import numpy only, not {name!r}")
    import builtins
    return builtins.__import__(name, *args, **kwargs)


SAFE_BUILTINS["__import__"] = _numpy_only_import


def source_sha256():
    """sha256 of the reference files the methods are read from (None when they are not on this machine) — nothing is parsed or executed.
    One digest over rnerf/datasets.py and rnerf/utils.py (the second only contributes namedtuple_map)."""
    if not (os.path.exists(SRC) and os.path.exists(SRC_UTILS)):
        return None
    return hashlib.sha256(open(SRC, "rb").read() + b"\0" + open(SRC_UTILS, "rb").read()).hexdigest()


def _reference_namedtuple_map():
    """rnerf/utils.py:70-72 compiled from the reference's text (the module itself imports jax: not importable here)."""
    tree = ast.parse(open(SRC_UTILS).read(), SRC_UTILS)
    for fn in (n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "namedtuple_map"):
        if fn.decorator_list:
            raise RuntimeError(f"{SRC_UTILS}: namedtuple_map carries a decorator: refusing to execute it")
        b = dict(SAFE_BUILTINS); b.update(type=type, map=map)
        ns = {"__builtins__": b}
        exec(compile(ast.Module(body=[fn], type_ignores=[]), SRC_UTILS, "exec"), ns)
        return ns["namedtuple_map"]
    raise RuntimeError(f"{SRC_UTILS}: namedtuple_map not found")


def reference_methods(expect_sha256=None):
    """{(class, method): function} compiled from the reference's own text; {} when the reference is not on this machine.

    The reference tree is untrusted content: the file's hash is taken BEFORE anything of it is compiled, and with `expect_sha256` (the test
    passes the hash committed in reference_numpy.npz) a file that is not the one the fixture was made from is refused unexecuted —
    decorators and default-argument expressions of a FunctionDef run at exec time.  The methods run with a whitelist of builtins."""
    if not (os.path.exists(SRC) and os.path.exists(SRC_UTILS)):
        return {}, None
    raw = open(SRC, "rb").read()
    sha = source_sha256()
    if expect_sha256 is not None and sha != expect_sha256:
        raise RuntimeError(f"{SRC}: sha256 {sha[:16]} is not the source the committed vectors were made from ({expect_sha256[:16]}): "
                           "nothing of it was executed; re-run tests/golden/make_from_reference_numpy.py after reading the diff")
    text = raw.decode()
    tree = ast.parse(text, SRC)
    ntmap = _reference_namedtuple_map()
    out = {}
    for cls in (n for n in tree.body if isinstance(n, ast.ClassDef)):
        for fn in (n for n in cls.body if isinstance(n, ast.FunctionDef)):
            if (cls.name, fn.name) in WANTED:
                if fn.decorator_list:
                    raise RuntimeError(f"{SRC}: {cls.name}.{fn.name} carries a decorator: refusing to execute it")
                mod = ast.Module(body=[fn], type_ignores=[])
                ns = {"__builtins__": dict(SAFE_BUILTINS), "np": np, "utils": types.SimpleNamespace(Rays=Rays, namedtuple_map=ntmap)}
                exec(compile(mod, SRC, "exec"), ns)
                out[(cls.name, fn.name)] = ns[fn.name]
    missing = WANTED - set(out)
    if missing:
        raise RuntimeError(f"{SRC}: methods not found: {sorted(missing)}")
    return out, sha


def inputs():
    rng = np.random.default_rng(20261003)
    q0, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    q1, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    c2w = np.stack([np.concatenate([q, rng.uniform(-3, 3, (3, 1))], -1) for q in (q0, q1)]).astype(np.float32)       # [2, 3, 4]
    H, W = 11, 14
    focal = np.float32(0.5 * W / np.tan(0.5 * 0.6911112070083618))             # example_data/transforms_train.json camera_angle_x
    cam_mat = np.array([[612.3, 0, 6.6], [0, 609.8, 5.2], [0, 0, 1]], np.float64)
    ndim = [9, 7, 8]                                                             # anisotropic on purpose
    nmin, nmax = [-1.5, -1.0, -0.75], [1.5, 1.25, 0.5]
    ior = (1.0 + 0.5 * rng.random(ndim)).astype(np.float32)
    ndelta = [(nmax[a] - nmin[a]) / (ndim[a] - 1.0) for a in range(3)]
    pts = rng.uniform([-1.8, -1.3, -1.0], [1.8, 1.5, 0.8], (256, 3))             # inside, outside (clamp to edge) ...
    k = rng.integers(0, [ndim[0], ndim[1], ndim[2]], (64, 3))
    on_nodes = np.array(nmin) + k * np.array(ndelta)                             # ... and exactly on nodes / cell faces
    pts = np.concatenate([pts, on_nodes]).astype(np.float32)
    bat_images = rng.uniform(0, 1, (2, H, W, 3)).astype(np.float32)              # two decoded views for the batch sampler (after every other draw: the older vectors keep their bits)
    return dict(c2w=c2w, H=H, W=W, focal=focal, cam_mat=cam_mat, ndim=np.array(ndim), nmin=np.array(nmin), nmax=np.array(nmax), ior=ior, pts=pts,
                ndelta=np.array(ndelta), bat_images=bat_images, bat_seed=np.array(7), bat_batch_size=np.array(16), bat_patch_size=np.array(2),
                bat_precrop_iters=np.array(2), bat_precrop_frac=np.array(0.5), bat_steps=np.array(4))


def compute(meth, x):
    H, W = int(x["H"]), int(x["W"])
    out = {}
    # focal and cam_mat as PYTHON floats: the reference holds cam_mat as json lists (datasets.py:463) and focal as a numpy float64 scalar
    # (:369), which its numpy (1.x, value-based casting) keeps out of the result type like a Python float — float32 arithmetic throughout;
    # NumPy >= 2 would promote a float64 SCALAR to float64 results, which is not what the reference computed
    blender = dict(focal=float(x["focal"]))
    opencv = dict(cam_mat=[[float(v) for v in row] for row in x["cam_mat"]])
    for tag, key, extra in (("blender", ("Dataset", "_generate_rays"), blender), ("opencv", ("OpenCV", "_generate_rays"), opencv)):
        for pc in (True, False):
            me = types.SimpleNamespace(w=W, h=H, use_pixel_centers=pc, camtoworlds=x["c2w"], **extra)
            meth[key](me)
            for f in Rays._fields:
                out[f"{tag}_pc{int(pc)}_{f}"] = np.asarray(getattr(me.rays, f))
    ndim = [int(v) for v in x["ndim"]]
    g = types.SimpleNamespace(ndim=ndim, nmin=[float(v) for v in x["nmin"]], nmax=[float(v) for v in x["nmax"]], ndelta=[float(v) for v in x["ndelta"]])
    grad = meth[("Grid", "_compute_grad")](g, x["ior"])                          # [Nx, Ny, Nz, 3]
    out["grad"] = np.asarray(grad)
    data = np.concatenate([x["ior"].reshape(-1, 1), grad.reshape(-1, 3)], -1)    # (n, dn/dx, dn/dy, dn/dz) per node: what the path's table holds
    out["lookup"] = np.asarray(meth[("Grid", "_linear3")](g, data, x["pts"]))
    # Dataset._next_train as the training loop calls it (datasets.py:113-116), on the state _train_init leaves (:123-143): the rays of
    # _generate_rays and the images, [n, H*W, .] ("single_image") or flattened over the views ("all_images"); numpy's GLOBAL generator,
    # seeded, is what the reference draws from — its state is put back afterwards
    me = types.SimpleNamespace(w=W, h=H, use_pixel_centers=True, camtoworlds=x["c2w"], **blender)
    meth[("Dataset", "_generate_rays")](me)
    n = x["c2w"].shape[0]
    keep = np.random.get_state()
    try:
        for batching in ("single_image", "all_images"):
            if batching == "single_image":
                images = x["bat_images"].reshape([-1, H * W, 3])
                rays = Rays(*[None if r is None else r.reshape([-1, H * W, r.shape[-1]]) for r in me.rays])
            else:
                images = x["bat_images"].reshape([-1, 3])
                rays = Rays(*[None if r is None else r.reshape([-1, r.shape[-1]]) for r in me.rays])
            ds = types.SimpleNamespace(batching=batching, rays=rays, images=images, batch_size=int(x["bat_batch_size"]), n_examples=n, train_it=0,
                                       precrop_iters=int(x["bat_precrop_iters"]), precrop_frac=float(x["bat_precrop_frac"]), h=H, w=W,
                                       patch_size=int(x["bat_patch_size"]) if batching == "single_image" else 0)      # (the reference's patch branch indexes rays[0][0].shape[0]: per-view layout only)
            np.random.seed(int(x["bat_seed"]))
            for it in range(int(x["bat_steps"])):
                b = meth[("Dataset", "_next_train")](ds)
                out[f"bat_{batching}_{it}_pixels"] = np.asarray(b["pixels"])
                for f in ("origins", "directions", "viewdirs"):
                    out[f"bat_{batching}_{it}_{f}"] = np.asarray(getattr(b["rays"], f))
                    if batching == "single_image":
                        out[f"bat_{batching}_{it}_env_{f}"] = np.asarray(getattr(b["env_rays"], f))
    finally:
        np.random.set_state(keep)
    return out


def main(path=OUT):
    meth, sha = reference_methods()
    if not meth:
        print(f"SKIPPED: {SRC} is not on this machine")
        return None
    x = inputs()
    y = compute(meth, x)
    np.savez_compressed(path, source_sha256=np.array(sha), **{f"in_{k}": v for k, v in x.items()}, **{f"out_{k}": v for k, v in y.items()})
    print(f"wrote {path}: {len(y)} arrays computed by {SRC} (sha256 {sha[:16]})")
    return path


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else OUT)
