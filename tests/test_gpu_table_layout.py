"""SURVEY §8f N2 (remainder): the 2x2x2-brick order of the IoR table (include/rnerf.h: rnerf_table_layout).  A layout only moves entries in
memory: the table's VALUES, every gathered voxel index, every marched position, every rendered colour must be the reference layout's bits
(reference index formula: rnerf/ior_utils.py:214-217)."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import ref_np as R
from samplenerfro_amd import _lib, models, ops, prng, synthetic as syn
from samplenerfro_amd.utils import Rays

DEV = "cuda:0"
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _grid(dims, seed=3):
    rng = np.random.default_rng(seed)
    g = 1.0 + 0.3 * rng.uniform(0, 1, dims)
    for ax in range(3):                      # smooth it a little: rays bend, no wild gradients
        g = 0.5 * (g + np.roll(g, 1, ax))
    return g.astype(np.float32)


@pytest.mark.parametrize("dims", [(24, 24, 24), (23, 26, 21), (2, 3, 5)])
def test_bricked_table_holds_the_reference_values(dims):
    nmin, nmax = [-1.5, -1.0, -0.5], [1.5, 1.2, 0.9]
    grid = _grid(dims)
    ref = ops.grid_build_table(T(grid), _lib.Grid.make(dims, nmin, nmax, "reference"))
    spec = _lib.Grid.make(dims, nmin, nmax, "bricks")
    brk = ops.grid_build_table(T(grid), spec)
    bx, by, bz = [(d + 1) // 2 for d in dims]
    assert ref.shape == (dims[0] * dims[1] * dims[2], 4) and brk.shape == (bx * by * bz * 8, 4)
    assert torch.equal(ops.table_reference_order(brk, spec), ref)
    np.testing.assert_array_equal(ref.cpu().numpy(), R.build_table(grid, list(dims), nmin, nmax))          # and both are the oracle's table
    # an even-aligned cell's 8 corners are one 128-byte line
    flat = brk.reshape(-1)
    cell = ref.reshape(*dims, 4)[0:2, 0:2, 0:2].reshape(8, 4)
    assert torch.equal(flat[:32].reshape(8, 4), cell)


@pytest.mark.parametrize("dims", [(24, 24, 24), (23, 26, 21)])
def test_query_and_march_are_bit_identical_in_both_layouts(dims):
    nmin, nmax = [-1.5] * 3, [1.5] * 3
    grid = _grid(dims, seed=5)
    specs = {k: _lib.Grid.make(dims, nmin, nmax, k) for k in ("reference", "bricks")}
    tabs = {k: ops.grid_build_table(T(grid), s) for k, s in specs.items()}
    rng = np.random.default_rng(1)
    pts = T(rng.uniform(-1.9, 1.9, (4096, 3)).astype(np.float32))            # inside and outside the box (clamp-to-edge)
    q = {k: ops.grid_query(tabs[k], specs[k], pts, want_idx=True) for k in specs}
    assert torch.equal(q["reference"][0], q["bricks"][0]) and torch.equal(q["reference"][1], q["bricks"][1])
    o, d = syn.sphere_rays(333, seed=9)
    N = 97
    m = {k: ops.march(tabs[k], specs[k], T(o), T(d), 2.0, 6.0, N, want_ior=True, want_vox=True) for k in specs}
    for a, b in zip(m["reference"], m["bricks"]):
        assert torch.equal(a, b)
    # against the oracle as well (positions, voxel indices)
    pos, dirs, dist, n, g, vox = R.path_sampler(o, d, R.build_table(grid, list(dims), nmin, nmax), list(dims), nmin, nmax, 2.0, 6.0, N, return_idx=True)
    np.testing.assert_array_equal(m["bricks"][0].cpu().numpy()[..., :3].transpose(1, 0, 2), pos)
    np.testing.assert_array_equal(m["bricks"][3].cpu().numpy().transpose(1, 0, 2), vox)


@pytest.mark.parametrize("stage", ["radiance", "all"])
def test_model_renders_the_same_bits_from_a_bricked_table(stage):
    G, S, F, P, B = 24, 16, 24, 4, 160
    nmin, nmax = [-1.5] * 3, [1.5] * 3
    grid = R.conv3d_normal(syn.scale_ior(syn.sphere_grid(G, 1.5, 0.6), 0.5).reshape(-1, 1), [G] * 3, 3, 1.0).reshape(G, G, G).astype(np.float32)
    pf = syn.init_params_flat(2, fine=True, bias_scale=0.05)
    flat = {k: T(v) for k, v in pf.items()}
    if stage == "all":
        rng = np.random.default_rng(4)
        so3 = syn.init_mlp_flat(rng, [(60, 128), (128, 128), (128, 128), (188, 128), (128, 3)], 0.05)
        so3[-(128 * 3 + 3):-3] = (0.05 * rng.standard_normal(128 * 3)).astype(np.float32)
        flat["so3_mlp"] = T(so3)
    o, d = syn.sphere_rays(B, seed=11)
    rays = Rays(T(o), None, T(d), None)
    key = prng.PRNGKey(5)
    outs = {}
    for layout in ("reference", "bricks"):
        model = models.NerfModel(ndim=[G] * 3, nmin=nmin, nmax=nmax, grid=T(grid), near=2.0, far=6.0, num_coarse_samples=S, num_fine_samples=F,
                                 num_path_samples=P, stage=stage, table_layout=layout)
        assert torch.equal(model.table_reference(), ops.table_reference_order(model.table, model.spec))
        ret, _ = model.apply(models.make_variables(dict(flat)), key, key, rays, False)
        outs[layout] = ret
    for lvl in range(2):
        for a, b in zip(outs["reference"][lvl], outs["bricks"][lvl]):
            assert torch.equal(a, b)
