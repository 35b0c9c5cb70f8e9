"""Analytic checks of the oracle's optional branches that no golden case exercises (the shipped configs switch them off)."""
import numpy as np

from oracle import ref_np as R
from samplenerfro_amd import synthetic as syn

F32 = np.float32


def test_noise_branch_closed_form_in_vacuum_with_a_silent_network():
    """add_gaussian_noise (rnerf/model_utils.py:438-453, rnerf/models.py:310-317,445-452).  With every weight zero the network's raw sigma is
    0, so sigma = softplus(noise_std * z - 1) is a function of the draw alone; in a vacuum table (straight unit-speed rays) the coarse
    opacity has the closed form acc = 1 - exp(-sum_i sigma_i * delta_i), delta = node spacing (last: 1e-3).  noise_std = 0 with draws and
    noise_std without draws both leave the noise-free result."""
    G, ext, B, Nc, P = 8, 1.5, 5, 8, 3
    ndim, nmin, nmax = [G] * 3, [-ext] * 3, [ext] * 3
    table = R.build_table(np.ones(G ** 3), ndim, nmin, nmax)
    o, d = syn.sphere_rays(B, seed=2)
    cfg = R.ModelConfig(ndim, nmin, nmax, num_coarse_samples=Nc, num_fine_samples=4, num_path_samples=P)
    flat = {k: np.zeros_like(v) for k, v in syn.init_params_flat(0).items()}
    params = syn.params_tree(flat)
    jit = np.arange(0, Nc * P, P) + 1
    rng = np.random.default_rng(0)
    z_c = rng.standard_normal((B, Nc)).astype(F32); z_f = rng.standard_normal((B, Nc + 4)).astype(F32)
    std = 1.5
    taps = {}
    ret, _ = R.nerf_forward(cfg, params, table, o, d, jit, noise_std=std, noise_c=z_c, noise_f=z_f, taps=taps)
    assert np.array_equal(taps["raw_sigma_c"][..., 0], z_c * F32(std)) and np.array_equal(taps["raw_sigma_f"][..., 0], z_f * F32(std))
    step = (cfg.far - cfg.near) / (Nc * P - 1)
    delta = np.full((B, Nc), P * step); delta[:, -1] = 1e-3
    sigma = np.log1p(np.exp(std * z_c.astype(np.float64) - 1.0))
    want = 1.0 - np.exp(-(sigma * delta).sum(-1))
    assert np.abs(ret[0][2] - want).max() < 2e-6
    plain, _ = R.nerf_forward(cfg, params, table, o, d, jit)
    zero, _ = R.nerf_forward(cfg, params, table, o, d, jit, noise_std=0.0, noise_c=z_c, noise_f=z_f)
    off, _ = R.nerf_forward(cfg, params, table, o, d, jit, noise_std=std)              # randomized = False: no draws, no noise
    for lvl in range(2):
        for a, b, c in zip(plain[lvl], zero[lvl], off[lvl]):
            assert np.array_equal(a, b) and np.array_equal(a, c)
    assert np.abs(plain[0][2] - ret[0][2]).max() > 0.05
