"""BASELINE configs[0] on the only real pixels the reference ships: example_data/imgs/r_0.png as the loader sees it
(tests/golden/example_image.npz, made by tests/golden/make_example_image.py: factor 2 -> 400 x 400), the camera of
example_data/transforms_train.json, the 128^3 grid voxelised from example_data/voxelize/mesh_4_128_1.5_1.165.obj (tests/golden/example_obj.npz)
after ri = 0.5 and the (3, 1.0) prefilter — the radiance stage of configs/example.yaml: 64 + 128 samples, P = 12 (N = 768 eikonal steps
through the refracting object), bg_weight 0.025, bg_smooth_weight 1.0 on a 128 x 128 env-map patch, randomized sampling, Adam with the
reference schedule (train.py:346-465; batching single_image, datasets.py:169-170: random pixels of the one view).

Two checks.  (1) The first 20 optimisation steps against an INDEPENDENT loop on the host: the numpy oracle marches and resamples with the
jitter / stratified draws it derives itself from the same jax.random key chain, torch float64 autograd differentiates train.py:75-162's
loss on those rows, Adam by the optax formulas — per-step losses within 1e-5.  (2) 1000 steps at the yaml's batch of 1024: the training
PSNR and the PSNR of the rendered 400 x 400 view against the photograph rise past stated marks (the "PSNR vs ref" half of BASELINE's metric
on real data; the reference's own curve cannot be produced offline).  Differences from the yaml, stated: lr_delay_steps 0 and
anneal_delay_steps 0 (the yaml holds the rate at 1 % and the background terms off for the first 2500 steps — longer than this test)."""
import math
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from oracle import ref_np as R, torch_ref as TR      # noqa: E402

pytestmark = pytest.mark.gpu
F32 = np.float32
S, F, P = 64, 128, 12
ANNEAL_MAX = 160000.0        # configs/example.yaml anneal_max_steps


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


@pytest.fixture(scope="module")
def scene():
    import cases
    from samplenerfro_amd import ops
    img = np.load(os.path.join(ROOT, "tests", "golden", "example_image.npz"))["rgba_sum4"]
    pixels = (img[..., :3].astype(F32) / F32(1020.0)).reshape(-1, 3)                 # datasets.py:340-357: / 255, INTER_AREA halving, [..., :3]
    _, _, counts = cases.load_example_obj()
    grid = cases.example_grid(counts).astype(F32)
    H = W = 400
    focal = 0.5 * W / math.tan(0.5 * cases.EXAMPLE_CAMERA_ANGLE_X)                    # datasets.py:361
    o, _, v = ops.generate_rays(cases.EXAMPLE_C2W, H, W, torch.device("cuda:0"), focal=focal)
    o_np, _, v_np = R.generate_rays(cases.EXAMPLE_C2W, H, W, focal=focal)
    assert np.array_equal(o.cpu().numpy(), o_np) and np.array_equal(v.cpu().numpy(), v_np)      # the device's rays are the oracle's bits
    gen = np.random.default_rng(cases.SEED)
    ev = gen.standard_normal((128, 128, 3)).astype(F32)
    ev = R.safe_l2_normalize(ev)
    return dict(pixels=pixels, grid=grid, o=o_np.reshape(-1, 3), v=v_np.reshape(-1, 3), H=H, W=W, ev=ev,
                table=R.build_table(grid, [128] * 3, [-1.5] * 3, [1.5] * 3))


def _flags(batch, max_steps):
    from samplenerfro_amd import utils as U
    return U.default_flags(num_coarse_samples=S, num_fine_samples=F, num_path_samples=P, white_bkgd=False, use_online_sparsity=False,
                           randomized=True, near=2.0, far=6.0, batch_size=batch, bg_weight=0.025, bg_smooth_weight=1.0, bg_patch_size=128,
                           lr_init=5e-4, lr_final=5e-6, lr_delay_steps=0, lr_delay_mult=0.01, max_steps=max_steps, config="configs/example")


def _device_setup(scene, flags, seed):
    from samplenerfro_amd import models, synthetic as syn
    from samplenerfro_amd.train import TrainState
    model, variables = models.construct_nerf(np.array([0, seed], np.uint32), None, flags, [128] * 3, [-1.5] * 3, [1.5] * 3, T(scene["grid"]))
    pf = syn.init_params_flat(seed, fine=True)
    for k in ("coarse_mlp", "fine_mlp", "bkgd_mlp"):
        variables["flat"][k].copy_(T(pf[k]))
    return model, TrainState.create(model, variables, flags), pf


@pytest.mark.timeout(1500)
def test_first_20_losses_equal_the_host_oracle_loop(scene):
    from samplenerfro_amd import prng, utils as U
    from samplenerfro_amd.train import train_step
    B, steps = 96, 20
    flags = _flags(B, 200000)
    model, state, pf = _device_setup(scene, flags, 3)
    env = U.Rays(None, None, T(scene["ev"]), None)
    pick = np.random.default_rng(11)
    idx = [pick.integers(0, scene["H"] * scene["W"], B) for _ in range(steps)]      # datasets.py:169-170
    rng = prng.PRNGKey(20200823)
    got = []
    r = rng
    for i in range(steps):
        batch = {"rays": U.Rays(T(scene["o"][idx[i]]), None, T(scene["v"][idx[i]]), None), "pixels": T(scene["pixels"][idx[i]]),
                 "annealed_alpha": max(i + 1, 0) / ANNEAL_MAX, "env_rays": env}
        state, stats, r = train_step(model, r, state, batch, flags)                      # the product step: two C calls, keys on the device
        got.append((float(stats.loss), float(stats.loss_c), float(stats.loss_bg), float(stats.loss_bg_smooth)))

    # ---- the same 20 steps on the host: numpy fp32 oracle (march, resampling) + torch float64 autograd + the optax Adam formulas -------------
    names = ["coarse_mlp", "fine_mlp", "bkgd_mlp"]
    th = {k: torch.tensor(pf[k], dtype=torch.float64, requires_grad=True) for k in names}
    mu = {k: torch.zeros_like(th[k]) for k in names}; nu = {k: torch.zeros_like(th[k]) for k in names}
    cfg = R.ModelConfig([128] * 3, [-1.5] * 3, [1.5] * 3, near=2.0, far=6.0, num_coarse_samples=S, num_fine_samples=F, num_path_samples=P)
    f64 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float64)
    eps32 = float(np.finfo(np.float32).eps)
    env_enc = f64(R.pos_enc(scene["ev"].reshape(-1, 3), 0, 4))
    r = rng
    worst = 0.0
    for i in range(steps):
        r, key_0, key_1 = prng.split(r, 3)                                              # train.py:72
        kj, _ = prng.split(key_0); ku, _ = prng.split(key_1)                             # models.py:232, 366
        jitter = (np.arange(0, S * P, P) + prng.randint(kj, (S,), 0, P)).astype(np.int32)            # models.py:240-242
        u = (np.arange(F, dtype=F32) * F32(1.0 / F))[None, :] + prng.uniform(ku, (B, F), maxval=1.0 / F - eps32)   # model_utils.py:345-354
        u = np.minimum(u, F32(1.0 - eps32)).astype(F32)
        o, d, pix = scene["o"][idx[i]], scene["v"][idx[i]], f64(scene["pixels"][idx[i]])
        params = {k: th[k].detach().numpy().astype(F32) for k in names}
        from samplenerfro_amd import synthetic as syn
        taps = {}
        R.nerf_forward(cfg, syn.params_tree(params), scene["table"], o, d, jitter, u_fine=u, taps=taps)      # the rows the reference would sample
        jit = jitter.astype(np.int64)

        def level(name, pos, dirs, t, bk):
            n = pos.shape[1]
            raw = TR.nerf_mlp(th[name], f64(R.pos_enc(pos.reshape(-1, 3), 0, 10)), f64(R.pos_enc(dirs.reshape(-1, 3), 0, 4))).reshape(B, n, 4)
            rgb, sigma = TR.activations(raw)
            comp, acc, w, trans, tb = TR.volumetric_rendering(rgb, sigma, f64(t), f64(dirs), bk)
            return comp, trans, tb

        pos_c, dir_c, t_c = taps["ray_pos"][:, jit], taps["ray_dir"][:, jit], taps["ray_dist"][:, jit]
        bk = TR.bkgd_mlp(th["bkgd_mlp"], f64(R.pos_enc(dir_c[:, -1], 0, 4)))
        levels = [level("coarse_mlp", pos_c, dir_c, t_c, bk), level("fine_mlp", taps["pos_f"], taps["dir_f"], taps["z_f"], bk)]
        alpha = (i + 1) / ANNEAL_MAX
        total, parts = TR.radiance_loss(levels, pix, flags.bg_weight, alpha)
        envc = TR.bkgd_mlp(th["bkgd_mlp"], env_enc).reshape(128, 128, 3)
        smooth = (0.5 * ((envc[1:, :] - envc[:-1, :]) ** 2).reshape(-1) + 0.5 * ((envc[:, 1:] - envc[:, :-1]) ** 2).reshape(-1)).mean()   # train.py:127-130
        (total + flags.bg_smooth_weight * float(alpha > 0) * smooth).backward()
        want = (float(parts["loss"]), float(parts["loss_c"]), flags.bg_weight * float(parts["loss_bg"]), float(smooth))
        for a, b in zip(got[i], want):
            worst = max(worst, abs(a - b))
            assert abs(a - b) < 1e-5, (i, got[i], want)
        lr = U.learning_rate_decay(i, flags.lr_init, flags.lr_final, flags.max_steps, flags.lr_delay_steps, flags.lr_delay_mult)   # optax count = updates so far
        with torch.no_grad():
            for k in names:
                g = th[k].grad
                mu[k] = 0.9 * mu[k] + 0.1 * g; nu[k] = 0.999 * nu[k] + 0.001 * g * g
                th[k] -= lr * (mu[k] / (1 - 0.9 ** (i + 1))) / (torch.sqrt(nu[k] / (1 - 0.999 ** (i + 1))) + 1e-8)
                th[k].grad = None
    print("example scene, 20 steps: largest |device - host loop| over the four loss terms =", worst)
    assert got[-1][0] < got[0][0]                                   # and it learns (the rate is 0 at the very first update, utils.py:519)


@pytest.mark.timeout(900)
def test_training_on_the_example_view_raises_the_psnr(scene):
    from samplenerfro_amd import prng, utils as U
    from samplenerfro_amd.train import train_step
    B, steps = 1024, 1000
    flags = _flags(B, 30000)                                        # (the yaml's commented short schedule: max_steps 30000)
    model, state, _ = _device_setup(scene, flags, 7)
    env = U.Rays(None, None, T(scene["ev"]), None)
    o_d, v_d, pix_d = T(scene["o"]), T(scene["v"]), T(scene["pixels"])
    pick = torch.Generator(device="cpu").manual_seed(5)
    r = prng.PRNGKey(7)
    curve = []
    for i in range(steps):
        idx = torch.randint(0, scene["H"] * scene["W"], (B,), generator=pick).to("cuda:0")
        batch = {"rays": U.Rays(o_d[idx], None, v_d[idx], None), "pixels": pix_d[idx], "annealed_alpha": (i + 1) / ANNEAL_MAX, "env_rays": env}
        state, stats, r = train_step(model, r, state, batch, flags)
        curve.append(stats.psnr.clone())            # (Stats fields are views into the step's device buffer: the next step overwrites them)
        if i == 7:
            early = state.theta.detach().clone()    # weights that have seen 8 batches of real pixels: the field is still partly opaque
    curve = torch.stack([c.reshape(()) for c in curve]).cpu().numpy()
    first, last = float(curve[:10].mean()), float(curve[-20:].mean())
    print("train PSNR, means of 100 steps:", [round(float(curve[k:k + 100].mean()), 2) for k in range(0, steps, 100)])
    # the rendered view against the photograph (render_image: eval precision, chunks of 8192 like utils.py:241-244)
    fn = lambda k0, k1, rays, path=None: model.apply(state.variables, k0, k1, rays, False, path=path)
    rays_hw = U.Rays(o_d.reshape(scene["H"], scene["W"], 3), None, v_d.reshape(scene["H"], scene["W"], 3), None)
    rgb, _, _ = U.render_image(fn, rays_hw, prng.PRNGKey(1), False, chunk=8192)
    mse = float(((rgb.reshape(-1, 3) - pix_d) ** 2).mean())
    psnr_view = U.compute_psnr(mse)
    print(f"example view: train PSNR {first:.2f} -> {last:.2f} dB over {steps} steps of {B} rays; rendered 400 x 400 view vs the photograph {psnr_view:.2f} dB")
    # the same TRAINED weights rendered in the other arithmetics: fp32-grade f16x3 (the training arithmetic and the default render pass), f16f8
    # and the single-pass f16 leg — how far apart the pictures are on weights that have seen real data, not on an initialisation
    # (on ONE view the field goes fully transparent within ~20 steps — the background MLP can explain a single photograph and the loss_bg term
    #  rewards transparency; the independent host loop of the test above takes the same road, so this is the algorithm, not the kernels —
    #  and from then on every arithmetic renders the same background: the comparison is made on the weights of step 8)
    import copy
    from samplenerfro_amd import _lib
    from samplenerfro_amd.train import _bump
    final = state.theta.detach().clone()
    state.theta.copy_(early); _bump(state.theta)
    pics = {}
    for name in ("f16x3", "f16f8", "f16", "bf16"):
        m2 = copy.copy(model)
        m2.precision = m2.eval_precision = _lib.PRECISIONS[name]
        m2._packed, m2._jit_cache, m2._ws, m2._key_cache, m2._u_lin, m2._side = {}, {}, {}, {}, None, None
        f2 = lambda k0, k1, rays, path=None, m2=m2: m2.apply(state.variables, k0, k1, rays, False, path=path)
        img = U.render_image(f2, rays_hw, prng.PRNGKey(1), False, chunk=8192)
        pics[name] = img[0]
        if name == "f16x3":
            print("opacity of the step-8 field over the view: mean %.3e, max %.3e" % (float(img[2].mean()), float(img[2].max())))
            assert float(img[2].mean()) > 0.02
    d = {k: float((v - pics["f16x3"]).abs().max()) for k, v in pics.items() if k != "f16x3"}
    print("max |dRGB| against the f16x3 render of the step-8 weights:", {k: f"{v:.2e}" for k, v in d.items()})
    state.theta.copy_(final); _bump(state.theta)
    assert model.eval_precision == model.precision == _lib.PREC_F16X3      # the default render pass IS the training arithmetic (round 6)
    assert d["f16f8"] < 1e-4 and d["f16"] < 1e-3 and d["bf16"] < 1e-2
    assert np.isfinite(curve).all() and bool(torch.isfinite(rgb).all())
    # measured (MI355X, this seed): 100-step means 15.07, 16.10, 16.19, 16.34, 16.41, 16.48, 16.55, 16.54, 16.58, 16.61 dB; first 10 steps 12.55;
    # the rendered view 16.6 dB.  (ONE view: within ~20 steps the optimiser makes the field transparent — opacity 0.64 -> exactly 0; the host
    # oracle loop of the test above walks the same road — and what keeps improving afterwards is the background MLP with its 4 view-direction
    # octaves.  The marks are about the curve rising on real pixels through the whole product path, not about image quality.)
    m = [float(curve[k:k + 100].mean()) for k in range(0, steps, 100)]
    assert m[0] < m[4] < m[9] and m[9] > m[0] + 1.2, m
    assert last > first + 3.5 and last > 16.2, (first, last)
    assert psnr_view > 16.0, psnr_view


@pytest.mark.timeout(900)
def test_single_pass_f16_arithmetic_trains_the_example_view_like_the_default(scene):
    """The single-pass leg (f16 forward + f16 backward: north_star's arithmetic, 1.58 x the default's rays/s) on the same real pixels, same
    batches and keys: its PSNR curve must follow the fp32-grade default's — the evidence that the leg is a usable training mode and not
    only a fast one.  (The volumetric field matters for the first ~20 steps only, see above; afterwards both runs train the background MLP,
    whose arithmetic is the same in both.)  (Measured: 100-step means 15.07 / 16.10 / 16.19 / 16.34 / 16.41 / 16.48 dB for the default and 15.07 / 16.20 / 16.32 / 16.50 /
    16.46 / 16.51 for the leg — two trajectories of a chaotic optimisation 0.1-0.16 dB apart, neither consistently ahead; rendered views 16.50 / 16.53.)"""
    from samplenerfro_amd import _lib, prng, utils as U
    from samplenerfro_amd.train import TrainState, train_step
    B, steps = 1024, 600
    env = U.Rays(None, None, T(scene["ev"]), None)
    o_d, v_d, pix_d = T(scene["o"]), T(scene["v"]), T(scene["pixels"])
    curves, views = {}, {}
    for name in ("f16x3", "f16"):
        flags = _flags(B, 30000)
        flags.backward_precision = name
        model, state, _ = _device_setup(scene, flags, 7)
        if name == "f16":
            model.precision = model.eval_precision = _lib.PREC_F16
            model._packed = {}
            state = TrainState.create(model, state.variables, flags)
        pick = torch.Generator(device="cpu").manual_seed(5)
        r = prng.PRNGKey(7)
        c = []
        for i in range(steps):
            idx = torch.randint(0, scene["H"] * scene["W"], (B,), generator=pick).to("cuda:0")
            batch = {"rays": U.Rays(o_d[idx], None, v_d[idx], None), "pixels": pix_d[idx], "annealed_alpha": (i + 1) / ANNEAL_MAX, "env_rays": env}
            state, stats, r = train_step(model, r, state, batch, flags)
            c.append(stats.psnr.clone())
        curves[name] = torch.stack([x.reshape(()) for x in c]).cpu().numpy()
        fn = lambda k0, k1, rays, path=None: model.apply(state.variables, k0, k1, rays, False, path=path)
        rays_hw = U.Rays(o_d.reshape(scene["H"], scene["W"], 3), None, v_d.reshape(scene["H"], scene["W"], 3), None)
        rgb, _, _ = U.render_image(fn, rays_hw, prng.PRNGKey(1), False, chunk=8192)
        views[name] = U.compute_psnr(float(((rgb.reshape(-1, 3) - pix_d) ** 2).mean()))
        assert state.nonfinite_grads() == 0
    m = {k: [float(v[j:j + 100].mean()) for j in range(0, steps, 100)] for k, v in curves.items()}
    print("train PSNR, 100-step means:", {k: [round(x, 2) for x in v] for k, v in m.items()}, "views:", {k: round(v, 2) for k, v in views.items()})
    assert np.isfinite(curves["f16"]).all()
    assert max(abs(a - b) for a, b in zip(m["f16"], m["f16x3"])) < 0.35
    assert abs(views["f16"] - views["f16x3"]) < 0.3 and m["f16"][-1] > m["f16"][0] + 1.0


def test_the_training_loop_through_the_device_batcher_equals_host_indexed_batches(scene):
    """SURVEY 8f N4 in the loop: `for batch in DeviceBatcher(...)` -> train_step on the reference's photograph and camera (views resident on
    the device, the reference's numpy draws, ONE gather launch per ray set) gives the very losses and parameters of the same steps fed with
    batches indexed on the host out of the ray arrays `_generate_rays` makes (what the reference's Dataset does) — bit for bit."""
    import cases
    from samplenerfro_amd import prng, utils as U
    from samplenerfro_amd.datasets import DeviceBatcher
    from samplenerfro_amd.train import train_step, flush_range_retry
    B, steps, ps = 96, 4, 8
    H, W = scene["H"], scene["W"]
    focal = 0.5 * W / math.tan(0.5 * cases.EXAMPLE_CAMERA_ANGLE_X)
    flags = _flags(B, 200000)
    flags.bg_patch_size = ps
    images = scene["pixels"].reshape(1, H, W, 3)
    c2w = np.asarray(cases.EXAMPLE_C2W, F32)[None, :3, :4]
    o_np, d_np, v_np = R.generate_rays(cases.EXAMPLE_C2W, H, W, focal=focal)
    o_np, d_np, v_np = (a.reshape(-1, 3) for a in (o_np, d_np, v_np))
    kw = dict(batch_size=B, batching="single_image", patch_size=ps, focal=focal, device="cuda:0", prefetch=0)
    out = {}
    for how in ("device", "host"):
        model, state, pf = _device_setup(scene, flags, 3)
        bat = DeviceBatcher(images, c2w, rng=np.random.RandomState(11), **kw)
        r = prng.PRNGKey(20200823)
        losses = []
        for i in range(steps):
            if how == "device":
                batch = next(bat)
            else:
                dr = bat.draw()
                idx, env = dr["ray_indices"], dr["env_indices"]
                batch = {"rays": U.Rays(T(o_np[idx]), T(d_np[idx]), T(v_np[idx]), None), "pixels": T(scene["pixels"][idx]),
                         "env_rays": U.Rays(T(o_np[env]), T(d_np[env]), T(v_np[env]), None)}
            batch["annealed_alpha"] = max(i + 1, 0) / ANNEAL_MAX
            state, stats, r = train_step(model, r, state, batch, flags)
            losses.append((float(stats.loss), float(stats.loss_c), float(stats.loss_bg), float(stats.loss_bg_smooth)))
        flush_range_retry(model, state)
        out[how] = (losses, state.theta.clone())
    assert out["device"][0] == out["host"][0], (out["device"][0], out["host"][0])
    assert torch.equal(out["device"][1], out["host"][1])
    assert all(np.isfinite(v) for l in out["device"][0] for v in l)
