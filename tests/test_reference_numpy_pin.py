"""Rows G2, G3 and 8f N4 against vectors computed BY THE REFERENCE (tests/golden/reference_numpy.npz, made by
tests/golden/make_from_reference_numpy.py: the reference's four numpy-only methods — Dataset._generate_rays, OpenCV._generate_rays,
Grid._linear3, Grid._compute_grad, rnerf/datasets.py:216-242,278-322,486-518 — executed from its own source).  Unlike every other golden
file of this repository these numbers do not come from the restatement under oracle/: they pin it, and through it the HIP entry points."""
import importlib.util
import os

import numpy as np
import pytest

from oracle import ref_np as R

HERE = os.path.dirname(os.path.abspath(__file__))
NPZ = os.path.join(HERE, "golden", "reference_numpy.npz")
F32 = np.float32


def _load():
    d = np.load(NPZ)
    x = {k[3:]: d[k] for k in d.files if k.startswith("in_")}
    y = {k[4:]: d[k] for k in d.files if k.startswith("out_")}
    return x, y, str(d["source_sha256"])


def _grid(x):
    return [int(v) for v in x["ndim"]], [float(v) for v in x["nmin"]], [float(v) for v in x["nmax"]]


def test_the_committed_vectors_are_what_the_reference_computes():
    """Wherever the reference is on the machine (the build container; not the GPU box) the generator is run again: same source hash, same bits."""
    spec = importlib.util.spec_from_file_location("make_from_reference_numpy", os.path.join(HERE, "golden", "make_from_reference_numpy.py"))
    gen = importlib.util.module_from_spec(spec); spec.loader.exec_module(gen)
    if gen.source_sha256() is None:
        pytest.skip("the reference is not on this machine")
    x, y, committed_sha = _load()
    # the hash is compared BEFORE anything of the (untrusted) reference file is compiled or executed
    assert gen.source_sha256() == committed_sha, "rnerf/datasets.py changed: read the diff, then re-run tests/golden/make_from_reference_numpy.py"
    meth, sha = gen.reference_methods(expect_sha256=committed_sha)
    again = gen.compute(meth, gen.inputs())
    assert set(again) == set(y)
    for k in y:
        assert again[k].dtype == y[k].dtype and np.array_equal(again[k], y[k]), k


def test_oracle_ray_generation_equals_the_references_bit_for_bit():
    x, y, _ = _load()
    H, W = int(x["H"]), int(x["W"])
    K = [[float(t) for t in r] for r in x["cam_mat"]]
    for cam in range(x["c2w"].shape[0]):
        for pc in (True, False):
            for tag, kw in (("blender", dict(focal=float(x["focal"]))), ("opencv", dict(cam_mat=K))):
                o, d, v = R.generate_rays(x["c2w"][cam], H, W, pixel_center=pc, **kw)
                pre = f"{tag}_pc{int(pc)}_"
                assert o.dtype == d.dtype == v.dtype == F32
                assert np.array_equal(o, y[pre + "origins"][cam]) and np.array_equal(d, y[pre + "directions"][cam]) and np.array_equal(v, y[pre + "viewdirs"][cam]), (tag, pc, cam)


def test_oracle_radii_agree_with_the_references():
    """The reference's radii come out float64 under this NumPy (np.sqrt(12) is a float64 SCALAR, which NumPy >= 2 no longer demotes) and
    float32 under the NumPy it pins: the oracle's float32 values agree to float32 rounding."""
    x, y, _ = _load()
    for cam in range(x["c2w"].shape[0]):
        for pre in ("blender_pc1_", "opencv_pc0_"):
            r = R.ray_radii(y[pre + "directions"][cam])
            want = y[pre + "radii"][cam]
            assert r.shape == want.shape and r.dtype == F32 and np.abs(r / want - 1).max() < 3e-7


def test_oracle_gradient_table_equals_the_references_bit_for_bit():
    """Grid._compute_grad: edge padding + central differences / (2 ndelta), per axis — the gradient columns of the path's table (row G2)."""
    x, y, _ = _load()
    ndim, nmin, nmax = _grid(x)
    table = R.build_table(x["ior"].reshape(-1), ndim, nmin, nmax)
    assert np.array_equal(table[:, 0], x["ior"].reshape(-1))
    assert np.array_equal(table[:, 1:4].reshape(*ndim, 3), y["grad"])


def test_oracle_trilinear_lookup_agrees_with_the_references():
    """Grid._linear3 (floor, the 8 corners, clamp to edge, 7 lerps) on 320 points: inside, outside the box, exactly on nodes and faces.  The
    reference's numpy twin promotes to float64 where an int array meets a float32 one; the path's lookup (ior_utils' jax _linear3, restated
    by the oracle) is float32 throughout: 2e-7 apart on values of order 1."""
    x, y, _ = _load()
    ndim, nmin, nmax = _grid(x)
    table = R.build_table(x["ior"].reshape(-1), ndim, nmin, nmax)
    got = R.linear3(table, x["pts"], ndim, nmin, nmax)
    assert got.dtype == F32 and np.abs(got.astype(np.float64) - y["lookup"]).max() < 5e-7
    outside = np.any((x["pts"] < np.array(nmin, F32)) | (x["pts"] > np.array(nmax, F32)), -1)
    assert 40 < outside.sum() < 280            # both regimes are in the sample


@pytest.mark.gpu
def test_hip_entry_points_against_the_references_vectors():
    """rnerf_generate_rays (both camera models, bit for bit), rnerf_grid_build_table (bit for bit) and rnerf_grid_query (2e-7: see above;
    bit-equal to the oracle's float32) against what the reference's own numpy methods return on the same inputs."""
    torch = pytest.importorskip("torch")
    from samplenerfro_amd import _lib, ops
    dev = torch.device("cuda:0")
    x, y, _ = _load()
    H, W = int(x["H"]), int(x["W"])
    K = [[float(t) for t in r] for r in x["cam_mat"]]
    for cam in range(x["c2w"].shape[0]):
        for pc in (True, False):
            for tag, kw in (("blender", dict(focal=float(x["focal"]))), ("opencv", dict(cam_mat=K))):
                o, d, v = ops.generate_rays(x["c2w"][cam], H, W, dev, pixel_center=pc, want_directions=True, **kw)
                pre = f"{tag}_pc{int(pc)}_"
                assert np.array_equal(o.cpu().numpy(), y[pre + "origins"][cam]) and np.array_equal(d.cpu().numpy(), y[pre + "directions"][cam])
                assert np.array_equal(v.cpu().numpy(), y[pre + "viewdirs"][cam]), (tag, pc, cam)
                assert np.abs(ops.ray_radii(d).cpu().numpy() / y[pre + "radii"][cam] - 1).max() < 1e-6
    ndim, nmin, nmax = _grid(x)
    for layout in ("reference", "bricks"):
        spec = _lib.Grid.make(ndim, nmin, nmax, layout)
        table = ops.grid_build_table(torch.from_numpy(x["ior"]).to(dev), spec)
        flat = ops.table_reference_order(table, spec).cpu().numpy()
        assert np.array_equal(flat[:, 0], x["ior"].reshape(-1)) and np.array_equal(flat[:, 1:4].reshape(*ndim, 3), y["grad"]), layout
        got = ops.grid_query(table, spec, torch.from_numpy(x["pts"]).to(dev)).cpu().numpy()
        assert np.abs(got.astype(np.float64) - y["lookup"]).max() < 5e-7, layout
        assert np.array_equal(got, R.linear3(R.build_table(x["ior"].reshape(-1), ndim, nmin, nmax), x["pts"], ndim, nmin, nmax))
