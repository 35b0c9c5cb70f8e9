"""The path driven by a host that is neither Python nor torch: tests/c_host/host_path.cpp is compiled with hipcc against include/rnerf.h,
linked with librnerf.so, and runs one evaluation forward (rnerf_forward) and one optimisation step (rnerf_rng_split3 ->
rnerf_train_forward_backward -> rnerf_adam_update) on inputs this test writes to a file.  Its outputs must be the bits the Python host
gets from model.apply / train_step on the same inputs: the C ABI is the product, the Python layer one of its callers."""
import os
import struct
import subprocess
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F32 = np.float32


@pytest.mark.timeout(600)
def test_cpp_host_reproduces_the_python_host(tmp_path):
    from oracle import ref_np as R
    from samplenerfro_amd import build, models, prng, synthetic as syn, utils
    from samplenerfro_amd.train import TrainState, train_step
    lib = build.build()
    exe = str(tmp_path / "host_path")
    cmd = [build._hipcc(), "--offload-arch=gfx950", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "c_host", "host_path.cpp"), "-o", exe,
           "-L" + os.path.dirname(lib), "-lrnerf", "-Wl,-rpath," + os.path.dirname(lib)]
    subprocess.check_call(cmd)
    G, Nc, Nf, P, B, ps, ext = 24, 16, 24, 4, 160, 8, 1.5
    grid = R.conv3d_normal(syn.scale_ior(syn.sphere_grid(G, ext, 0.6), 0.5).reshape(-1, 1), [G] * 3, 3, 1.0).reshape(G, G, G).astype(F32)
    pf = syn.init_params_flat(3, fine=True, bias_scale=0.1)
    o, d = syn.sphere_rays(B, seed=3)
    rng = np.random.default_rng(7)
    pix = rng.uniform(0, 1, (B, 3)).astype(F32)
    ev = rng.standard_normal((ps, ps, 3)).astype(F32); ev /= np.linalg.norm(ev, axis=-1, keepdims=True)
    k0, k1, kt = prng.PRNGKey(5), prng.PRNGKey(9), np.array([1, 2], np.uint32)
    with open(tmp_path / "in.bin", "wb") as f:
        f.write(struct.pack("6i", G, Nc, Nf, P, B, ps)); f.write(struct.pack("3d", 2.0, 6.0, ext))
        f.write(np.concatenate([k0, k1]).astype(np.uint32).tobytes()); f.write(kt.tobytes())
        for a in (grid, pf["coarse_mlp"], pf["fine_mlp"], pf["bkgd_mlp"], o, d, pix, ev):
            f.write(np.ascontiguousarray(a, F32).tobytes())
    out = subprocess.run([exe, str(tmp_path / "in.bin"), str(tmp_path / "out.bin")], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    n_theta = 2 * 595844 + 56963
    res = np.fromfile(tmp_path / "out.bin", F32)
    assert res.size == 18 * B + (n_theta + 8) + n_theta
    oc, of, g, th = res[:9 * B], res[9 * B:18 * B], res[18 * B:18 * B + n_theta + 8], res[18 * B + n_theta + 8:]

    # the Python host on the same inputs
    dev = torch.device("cuda:0")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    model = models.NerfModel(ndim=[G] * 3, nmin=[-ext] * 3, nmax=[ext] * 3, grid=T(grid), num_coarse_samples=Nc, num_fine_samples=Nf, num_path_samples=P,
                             white_bkgd=False, precision="f16x3")
    variables = models.make_variables({k: T(v) for k, v in pf.items()})
    rays = utils.Rays(T(o), None, T(d), None)
    ret, _ = model.apply(variables, k0, k1, rays, False)
    for lvl, buf in ((0, oc), (1, of)):
        rgb, dist, acc, trans, tb = [x.cpu().numpy() for x in ret[lvl]]
        assert np.array_equal(buf[:3 * B].reshape(B, 3), rgb) and np.array_equal(buf[3 * B:4 * B], dist) and np.array_equal(buf[4 * B:5 * B], acc)
        assert np.array_equal(buf[5 * B:6 * B], trans.reshape(-1)) and np.array_equal(buf[6 * B:].reshape(B, 3), tb)
    flags = utils.default_flags(num_coarse_samples=Nc, num_fine_samples=Nf, num_path_samples=P, white_bkgd=False, bg_weight=0.025, bg_smooth_weight=1.0,
                                bg_patch_size=ps, use_online_sparsity=False, randomized=True)
    state = TrainState.create(model, variables, flags)
    batch = {"rays": rays, "pixels": T(pix), "annealed_alpha": 0.5, "env_rays": utils.Rays(None, None, T(ev), None)}
    state, stats, _ = train_step(model, kt, state, batch, flags)
    g_py = state.grads.cpu().numpy()
    assert np.array_equal(g[:n_theta], g_py[:n_theta])                     # every gradient bit
    assert np.allclose(g[n_theta:], g_py[n_theta:], rtol=1e-6, atol=1e-9)  # the stats scalars (atomics in the loss sums)
    assert np.array_equal(th, state.theta.cpu().numpy())                   # and the parameters after the update
    assert abs(float(stats.loss) - g[n_theta]) < 1e-6 and "host_path ok" in out.stdout
