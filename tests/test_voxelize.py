"""the voxeliser.  CPU: the oracle's containment test on analytic cases; GPU: rnerf_voxelize vs the oracle."""
import numpy as np
import pytest

from oracle import ref_np as R

CUBE_V = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], np.float64)
CUBE_F = np.array([[0, 1, 3], [0, 3, 2], [4, 6, 7], [4, 7, 5], [0, 4, 5], [0, 5, 1], [2, 3, 7], [2, 7, 6], [0, 2, 6], [0, 6, 4],
                   [1, 5, 7], [1, 7, 3]], np.int32)


def _rotated_box(seed=3, half=(0.55, 0.35, 0.45), centre=(0.05, -0.1, 0.08)):
    rng = np.random.default_rng(seed)
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    return (CUBE_V * np.asarray(half)) @ q.T + np.asarray(centre), CUBE_F, q, np.asarray(half), np.asarray(centre)


def test_oracle_containment_cube_and_rotated_box():
    rng = np.random.default_rng(0)
    pts = rng.uniform(-1.5, 1.5, (400, 3))
    assert np.array_equal(R.mesh_contains(CUBE_V * 0.7, CUBE_F, pts), np.all(np.abs(pts) < 0.7, axis=-1))
    v, f, q, half, c = _rotated_box()
    local = (pts - c) @ q
    assert np.array_equal(R.mesh_contains(v, f, pts), np.all(np.abs(local) < half, axis=-1))
    # points exactly on a shared edge / vertex of the projection are counted once (top-left rule): a column through the cube's vertex
    edge_pts = np.array([[0.7, 0.7, 0.0], [0.7, 0.0, 0.0], [-0.7, -0.7, 0.0], [0.0, 0.0, 0.0]])
    got = R.mesh_contains(CUBE_V * 0.7, CUBE_F, edge_pts)
    assert got[3] and got.sum() in (1, 2, 3, 4)          # interior point inside; boundary points consistently in or out, never NaN/garbage


def test_oracle_voxelize_axis_aligned_cube_fractions():
    """A cube whose faces sit strictly between sub-sample planes: every voxel value is (n_in*1.33 + n_out)/K^3 with n_in separable."""
    G, K = 5, 4
    out = R.voxelize(CUBE_V * 0.6, CUBE_F, G, [-1] * 3, [1] * 3, K)
    pitch = 2.0 / (G - 1)
    lin = np.linspace(-1, 1, G)
    frac = np.array([np.mean(np.abs(c + np.linspace(-1, 1, K) * pitch) < 0.6) for c in lin])
    want = 1.0 + 0.33 * frac[:, None, None] * frac[None, :, None] * frac[None, None, :]
    assert np.abs(out - want).max() < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["cube", "rotated_box", "two_shells"])
def test_device_voxeliser_matches_the_oracle(case):
    torch = pytest.importorskip("torch")
    from samplenerfro_amd import voxelize as V
    if case == "cube":
        v, f = CUBE_V * 0.6, CUBE_F
    elif case == "rotated_box":
        v, f = _rotated_box()[:2]
    else:                                               # nested boxes: inside the outer, outside the inner (4 crossings per column)
        v2, f2 = _rotated_box(seed=5, half=(0.25, 0.2, 0.3))[:2]
        v = np.concatenate([CUBE_V * 0.8, v2]); f = np.concatenate([CUBE_F, f2 + 8])
    G, K = 9, 3
    out, ndim, nmin, nmax = V.voxelize(v, f, G, extent=1.0, num_samples=K, device="cuda:0", num_bins=5)
    ref = R.voxelize(v, f, G, [-1] * 3, [1] * 3, K)
    assert ndim == [G] * 3 and nmin == [-1.0] * 3
    assert np.abs(out.cpu().numpy() - ref).max() < 1e-6
    assert ref.max() > 1.1 and ref.min() < 1.05


@pytest.mark.gpu
def test_device_voxeliser_sphere_volume_and_pkl_round_trip(tmp_path):
    torch = pytest.importorskip("torch")
    from samplenerfro_amd import voxelize as V, grid as Gd
    # UV sphere of radius 0.5, written to and read back from an OBJ file
    nu, nv, r = 48, 24, 0.5
    vs = [[0, 0, r]] + [[r * np.sin(np.pi * b / nv) * np.cos(2 * np.pi * a / nu), r * np.sin(np.pi * b / nv) * np.sin(2 * np.pi * a / nu),
                         r * np.cos(np.pi * b / nv)] for b in range(1, nv) for a in range(nu)] + [[0, 0, -r]]
    fs = []
    ring = lambda b, a: 1 + (b - 1) * nu + (a % nu)
    for a in range(nu):
        fs.append([0, ring(1, a), ring(1, a + 1)])
        fs.append([len(vs) - 1, ring(nv - 1, a + 1), ring(nv - 1, a)])
        for b in range(1, nv - 1):
            fs += [[ring(b, a), ring(b + 1, a), ring(b + 1, a + 1)], [ring(b, a), ring(b + 1, a + 1), ring(b, a + 1)]]
    obj = tmp_path / "mesh.obj"
    obj.write_text("".join(f"v {x} {y} {z}\n" for x, y, z in vs) + "".join(f"f {a + 1} {b + 1} {c + 1}\n" for a, b, c in fs))
    verts, faces = V.load_obj(str(obj))
    G = 64
    out, ndim, nmin, nmax = V.voxelize(verts, faces, G, extent=1.0, num_samples=4, device="cuda:0")
    vol = float(((out.double() - 1.0) / 0.33).sum()) * (2.0 / (G - 1)) ** 3
    assert abs(vol - 4 / 3 * np.pi * r ** 3) / (4 / 3 * np.pi * r ** 3) < 0.02        # polyhedral sphere, box-filtered: within 2 %
    V.save_mesh_pkl(str(tmp_path / "mesh.pkl"), out, extent=1.0, num_voxels=G)
    data, nd, mn, mx = Gd.load_mesh_pkl(str(tmp_path / "mesh.pkl"))
    assert nd == [G] * 3 and mn == [-1.0] * 3 and np.array_equal(data.reshape(G, G, G).astype(np.float32), out.cpu().numpy())


# ---- three-axis majority containment (meshes that are not watertight) ----------------------------------------------------------------
OPEN_TOP_F = np.array([t for t in CUBE_F.tolist() if not all(CUBE_V[i][2] > 0 for i in t)], np.int32)      # the cube without its +z face


def test_oracle_majority_containment_survives_a_hole():
    """A box with its +z face removed: a parity ray along +z escapes through the hole (every interior sample votes 'outside', and the
    samples BELOW the box, whose ray crosses the bottom face once, vote 'inside'); the rays along +x and +y still cross a wall each — the
    majority restores the closed box; on closed meshes it changes nothing.  (Half-width 0.55: no face coincides with a sample plane.)"""
    G, K = 6, 3
    lo, hi = [-1.0] * 3, [1.0] * 3
    box = CUBE_V * 0.55
    closed = R.voxelize_counts(box, CUBE_F, G, lo, hi, K)
    assert closed.sum() > 0 and np.array_equal(R.voxelize_counts_robust(box, CUBE_F, G, lo, hi, K), closed)
    assert len(OPEN_TOP_F) == 10
    single = R.voxelize_counts(box, OPEN_TOP_F, G, lo, hi, K)
    robust = R.voxelize_counts_robust(box, OPEN_TOP_F, G, lo, hi, K)
    assert not np.array_equal(single, closed) and single[3, 3, 3] == 0 and closed[3, 3, 3] == 8     # the single ray misses the interior
    assert np.array_equal(robust, closed)                                              # the majority sees the box
    # a rotated box: still identical on the watertight mesh
    v, f = _rotated_box()[:2]
    assert np.array_equal(R.voxelize_counts_robust(v, f, G, lo, hi, K), R.voxelize_counts(v, f, G, lo, hi, K))


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["open_box", "rotated_box", "anisotropic"])
def test_device_majority_voxeliser_matches_the_oracle(case):
    torch = pytest.importorskip("torch")
    from samplenerfro_amd import voxelize as V
    G, K = 9, 3
    kw = dict(extent=1.0)
    lo, hi = [-1.0] * 3, [1.0] * 3
    if case == "open_box":
        v, f = CUBE_V * 0.55, OPEN_TOP_F
    elif case == "rotated_box":
        v, f = _rotated_box()[:2]
    else:                                           # different extents per axis: the axis rotation must carry nmin / nmax along
        v, f = _rotated_box(seed=7, half=(0.5, 0.3, 0.2))[:2]
        lo, hi = [-1.0, -0.8, -0.6], [1.2, 0.9, 0.7]
        kw = dict(min_point=lo, max_point=hi)
    out, ndim, nmin, nmax = V.voxelize(v, f, G, num_samples=K, device="cuda:0", num_bins=5, robust=True, **kw)
    want = R.counts_to_ior(R.voxelize_counts_robust(v, f, G, lo, hi, K), K)
    assert np.abs(out.cpu().numpy() - want).max() < 1e-6 and want.max() > 1.1
    if case == "open_box":                          # and the default single-ray test gets the open box wrong
        plain, _, _, _ = V.voxelize(v, f, G, num_samples=K, device="cuda:0", num_bins=5, **kw)
        assert not torch.equal(plain, out) and float(plain[4, 4, 5]) == 1.0 and float(out[4, 4, 5]) > 1.0
    else:
        plain, _, _, _ = V.voxelize(v, f, G, num_samples=K, device="cuda:0", num_bins=5, **kw)
        assert torch.equal(plain, out)
