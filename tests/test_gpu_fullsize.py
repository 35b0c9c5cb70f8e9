"""BASELINE.json's full sizes (4096 rays x 128 samples, N = 1536 eikonal steps, 512^3 grid) through size-independent
properties: the oracle would take minutes here, so each test checks something the domain guarantees at any size."""
import numpy as np
import pytest

torch = pytest.importorskip("torch")

from samplenerfro_amd import synthetic as syn

pytestmark = pytest.mark.gpu
B, S, P, G = 4096, 128, 12, 512
N = S * P


@pytest.fixture(scope="module")
def world():
    from samplenerfro_amd import models
    from samplenerfro_amd.utils import Rays
    dev = torch.device("cuda:0")
    out = {}
    for name, radius in (("vacuum", 0.0), ("sphere", 0.6)):
        if radius > 0:
            a = torch.linspace(-1.5, 1.5, G, dtype=torch.float64, device=dev)
            r = torch.sqrt(a[:, None, None] ** 2 + a[None, :, None] ** 2 + a[None, None, :] ** 2)
            grid = (1.0 + 0.5 * torch.clamp((radius - r) / (3.0 / (G - 1)) + 0.5, 0.0, 1.0)).float()
            del r
        else:
            grid = torch.ones((G, G, G), device=dev)
        out[name] = models.NerfModel(ndim=[G] * 3, nmin=[-1.5] * 3, nmax=[1.5] * 3, grid=grid, num_coarse_samples=S, num_fine_samples=256,
                                     num_path_samples=P, device=dev)
        del grid
    pf = syn.init_params_flat(0, fine=True, bias_scale=0.05)
    out["variables"] = models.make_variables({k: torch.from_numpy(v).to(dev) for k, v in pf.items()})
    o, d = syn.sphere_rays(B)
    out["rays"] = Rays(torch.from_numpy(o).to(dev), None, torch.from_numpy(d).to(dev), None)
    return out


def test_vacuum_march_is_repeated_addition(world):
    """KAT 1: n = 1, grad n = 0  =>  rp_k by k repeated fp32 additions of fl(step * d), dir constant, dist by repeated
    addition of the step length — bit-exact at 4096 x 1536."""
    from samplenerfro_amd import ops
    m, rays = world["vacuum"], world["rays"]
    pd, dr, _, _ = ops.march(m.table, m.spec, rays.origins, rays.viewdirs, m.near, m.far, N)
    step = torch.tensor(np.float32((m.far - m.near) / (N - 1)), device=pd.device)
    d = rays.viewdirs
    p = rays.origins + torch.tensor(np.float32(m.near), device=pd.device) * d
    inc = (step / torch.ones_like(d[:, :1])) * d                         # (step / n) * rd with n == 1
    t = torch.full((B,), np.float32(m.near), device=pd.device)
    nrm = torch.sqrt(torch.clamp((d * d)[:, 0] + (d * d)[:, 1] + (d * d)[:, 2], min=1e-6))
    dn = d / nrm[:, None]
    for k in range(N):
        if k in (0, 1, 2, 7, 100, 777, N - 1):
            assert torch.equal(pd[k, :, :3], p) and torch.equal(pd[k, :, 3], t) and torch.equal(dr[k, :, :3], dn)
        q = p + inc
        dl = p - q
        t = t + torch.sqrt((dl * dl)[:, 0] + (dl * dl)[:, 1] + (dl * dl)[:, 2])
        p = q


def test_rays_are_independent(world):
    """Permuting the rays of a batch permutes the outputs bit-exactly (march, MLP tiles, compositing, resampling: no cross-ray term)."""
    m, v, rays = world["sphere"], world["variables"], world["rays"]
    key = np.array([0, 42], np.uint32)
    perm = torch.randperm(B, generator=torch.Generator().manual_seed(1)).to(rays.origins.device)
    ret_a, _ = m.apply(v, key, key, rays, False)
    ret_b, _ = m.apply(v, key, key, type(rays)(rays.origins[perm].contiguous(), None, rays.viewdirs[perm].contiguous(), None), False)
    for la, lb in zip(ret_a, ret_b):
        for a, b in zip(la, lb):
            assert torch.equal(a[perm], b)


def test_compositing_invariants_and_sorted_resampling(world):
    m, v, rays = world["sphere"], world["variables"], world["rays"]
    key = np.array([0, 7], np.uint32)
    taps = {}
    ret, _ = m.apply(v, key, key, rays, True, taps=taps)
    for lvl, wname in ((0, "weights_c"), (1, "weights_f")):
        rgb, dist, acc, trans, tb = ret[lvl]
        w = taps[wname]
        assert torch.isfinite(rgb).all() and torch.isfinite(dist).all()
        assert float((w.sum(0) + trans.reshape(-1) - 1).abs().max()) < 2e-5          # partition of unity: sum w + T_last = 1
        assert float((acc - w.sum(0)).abs().max()) < 1e-5
        assert float(rgb.min()) >= -0.001 - 1e-5 and float(rgb.max()) <= 1.001 + 1e-5  # convex combination of padded sigmoids / bkgd
        assert bool((w >= 0).all())
    z = taps["rows_pd"][..., 3]                                                        # merged coarse + fine depths, [S+F, B]
    assert bool((z[1:] >= z[:-1]).all())
    idx = taps["idx_f"].long()
    assert int(idx.min()) >= 0 and int(idx.max()) <= N - 1
    zn = taps["path_pd"][..., 3]                                                       # node depths [N, B]
    cols = torch.arange(B, device=z.device)[None, :]
    lo = zn[idx, cols]
    hi = zn[torch.clamp(idx + 1, max=N - 1), cols]
    assert bool((z >= lo).all())                                                       # idx = max(searchsorted_left - 1, 0)
    inner = (idx + 1 <= N - 1) & (z > zn[0][None, :])
    assert bool((z[inner] <= hi[inner]).all())
    assert bool(((z > lo) | (idx == 0))[z > zn[0][None, :]].all())                     # 'left': z equal to a node depth maps to the node before


def test_gradient_is_the_mean_of_the_half_batch_gradients():
    """mse terms are means over rays: grad(batch) = (grad(first half) + grad(second half)) / 2 — a checksum of the whole
    backward path (different tile / workgroup decompositions) at the full 4096 x 128 size."""
    from samplenerfro_amd import models, utils as U
    from samplenerfro_amd.train import TrainState, train_step
    from samplenerfro_amd.utils import Rays
    dev = torch.device("cuda:0")
    Gs = 64
    flags = U.default_flags(num_coarse_samples=S, num_fine_samples=0, num_path_samples=P, white_bkgd=False, bg_weight=0.0, bg_smooth_weight=0.0,
                            use_online_sparsity=False, randomized=False)
    model, variables = models.construct_nerf(np.array([0, 3], np.uint32), None, flags, [Gs] * 3, [-1.5] * 3, [1.5] * 3,
                                             torch.ones((Gs, Gs, Gs), device=dev))
    pf = syn.init_params_flat(3, fine=False, bias_scale=0.05)
    for k in ("coarse_mlp", "bkgd_mlp"):
        variables["flat"][k].copy_(torch.from_numpy(pf[k]).to(dev))
    o, d = syn.sphere_rays(B, seed=11)
    pix = torch.from_numpy(np.random.default_rng(4).uniform(0, 1, (B, 3)).astype(np.float32)).to(dev)
    o, d = torch.from_numpy(o).to(dev), torch.from_numpy(d).to(dev)
    jitter = np.arange(0, N, P) + 5
    grads = []
    for lo, hi in ((0, B), (0, B // 2), (B // 2, B)):
        state = TrainState.create(model, variables, flags)
        taps = {}
        train_step(model, np.array([1, 1], np.uint32), state, {"rays": Rays(o[lo:hi].contiguous(), None, d[lo:hi].contiguous(), None),
                                                               "pixels": pix[lo:hi].contiguous(), "annealed_alpha": 0.0}, jitter=jitter, taps=taps)
        grads.append(taps["grads"].double())
    full, halves = grads[0], 0.5 * (grads[1] + grads[2])
    cos = float((full * halves).sum() / (full.norm() * halves.norm()))
    err = float((full - halves).abs().max() / full.abs().max())
    print(f"full vs mean-of-halves gradient: cosine {cos:.7f}, max err / max |g| {err:.2e}")
    assert cos > 0.99999 and err < 2e-3
