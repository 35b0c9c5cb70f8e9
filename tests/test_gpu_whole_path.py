"""The whole-path entry points of the C ABI (include/rnerf.h: rnerf_rng_*, rnerf_forward, rnerf_train_forward_backward,
rnerf_adam_update, rnerf_graph_*): one call per ray batch must give what the stage-by-stage host sequence gives — bit for bit for the
forward and the gradient — and the launch graph must replay the eager step."""
import ctypes as C

import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu
F32 = np.float32
DEV = "cuda:0"


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


def _scene(Nc=16, Nf=24, P=4, B=200, G=24, bd_cut=False, seed=3, **kw):
    from oracle import ref_np as R
    from samplenerfro_amd import models, synthetic as syn
    from samplenerfro_amd.utils import Rays
    ndim, nmin, nmax = [G] * 3, [-1.5] * 3, [1.5] * 3
    grid = R.conv3d_normal(syn.scale_ior(syn.sphere_grid(G, 1.5, 0.6), 0.5).reshape(-1, 1), ndim, 3, 1.0).reshape(ndim)
    extra = dict(bd_cut_dist=0.1, cfg_name="glass") if bd_cut else {}
    model = models.NerfModel(ndim=ndim, nmin=nmin, nmax=nmax, grid=T(grid.astype(F32)), num_coarse_samples=Nc, num_fine_samples=Nf,
                             num_path_samples=P, precision="f16x3", white_bkgd=False, **extra, **kw)
    pf = syn.init_params_flat(seed, fine=Nf > 0, bias_scale=0.1)
    variables = models.make_variables({k: T(v) for k, v in pf.items()})
    o, d = syn.sphere_rays(B, seed=seed)
    return model, variables, Rays(T(o), None, T(d), None), pf


@pytest.mark.parametrize("Nc,P", [(64, 12), (7, 1), (128, 24), (33, 5)])
def test_device_key_chain_matches_the_host_prng(Nc, P):
    from samplenerfro_amd import _lib, prng
    lib = _lib.load()
    st = _lib.current_stream()
    for seed in (0, 20200823, 2 ** 40 + 17):
        rng = prng.PRNGKey(seed)
        state = T(rng.view(np.int32).copy())
        keys4 = torch.zeros(4, dtype=torch.int32, device=DEV)
        _lib.check(lib.rnerf_rng_split3(state.data_ptr(), keys4.data_ptr(), st))
        want = prng.split(rng, 3)
        assert np.array_equal(state.cpu().numpy().view(np.uint32), want[0])
        assert np.array_equal(keys4.cpu().numpy().view(np.uint32), np.concatenate([want[1], want[2]]))
        jit = torch.zeros(Nc, dtype=torch.int32, device=DEV); key_u = torch.zeros(2, dtype=torch.int32, device=DEV)
        _lib.check(lib.rnerf_rng_forward(keys4.data_ptr(), Nc, P, 1, jit.data_ptr(), key_u.data_ptr(), st))
        key, _ = prng.split(want[1])
        assert np.array_equal(jit.cpu().numpy(), np.arange(0, Nc * P, P, dtype=np.int32) + prng.randint(key, (Nc,), 0, P))
        ku, _ = prng.split(want[2])
        assert np.array_equal(key_u.cpu().numpy().view(np.uint32), ku)
        _lib.check(lib.rnerf_rng_forward(keys4.data_ptr(), Nc, P, 0, jit.data_ptr(), key_u.data_ptr(), st))
        assert np.array_equal(jit.cpu().numpy(), np.arange(0, Nc * P, P, dtype=np.int32))
        B, F = 37, 24
        u = torch.zeros((F, B), dtype=torch.float32, device=DEV)
        _lib.check(lib.rnerf_stratified_u_dev(key_u.data_ptr(), B, F, u.data_ptr(), st))
        from samplenerfro_amd import ops
        assert torch.equal(u, ops.stratified_u(ku, B, F, DEV))


@pytest.mark.parametrize("Nf,randomized,bd_cut,B", [(0, False, False, 200), (24, False, False, 200), (24, True, False, 333), (24, True, True, 64)])
def test_rnerf_forward_equals_the_staged_sequence(Nf, randomized, bd_cut, B):
    """model.apply through rnerf_forward (one C call) against the stage-by-stage host sequence: every output bit for bit, with the
    jitter / draws from the device key chain, with an injected jitter, and on a pre-marched path."""
    from samplenerfro_amd import prng
    model, variables, rays, _ = _scene(Nf=Nf, B=B, bd_cut=bd_cut)
    k0, k1 = prng.PRNGKey(5), prng.PRNGKey(9)
    jit = np.arange(0, 64, 4, dtype=np.int32) + np.array([1, 0, 3, 2] * 4, np.int32)
    for kw in ({}, {"jitter": jit}, {"path": True}):
        res = []
        for whole in (True, False):
            model.whole_path = whole
            k = dict(kw)
            if k.get("path"):
                k["path"] = model.prefetch_path(rays)
            ret, _ = model.apply(variables, k0, k1, rays, randomized, **k)
            res.append(ret)
        assert len(res[0]) == len(res[1]) == (2 if Nf else 1)
        for la, lb in zip(*res):
            for a, b in zip(la, lb):
                assert a.shape == b.shape and torch.equal(a, b), kw


def test_rnerf_forward_direct_ctypes_call():
    """The entry point as a non-Python host would use it: descriptor, workspace query, one call; outputs = model.apply."""
    from samplenerfro_amd import _lib, prng
    model, variables, rays, _ = _scene(Nf=24, B=150)
    lib = _lib.load()
    m = model.c_model(variables)
    B = 150
    ws = torch.empty(lib.rnerf_forward_workspace_bytes(C.byref(m), B), dtype=torch.uint8, device=DEV)
    keys = T(np.concatenate([prng.PRNGKey(5), prng.PRNGKey(9)]).view(np.int32).copy())
    jit = torch.zeros(16, dtype=torch.int32, device=DEV); key_u = torch.zeros(2, dtype=torch.int32, device=DEV)
    st = _lib.current_stream()
    _lib.check(lib.rnerf_rng_forward(keys.data_ptr(), 16, 4, 1, jit.data_ptr(), key_u.data_ptr(), st))
    u = model.make_u(None, B, False)
    oc = torch.zeros(9 * B, device=DEV); of = torch.zeros(9 * B, device=DEV)
    _lib.check(lib.rnerf_forward(C.byref(m), rays.origins.data_ptr(), rays.viewdirs.data_ptr(), B, jit.data_ptr(), u.data_ptr(), 0, None, None,
                                 oc.data_ptr(), of.data_ptr(), ws.data_ptr(), 0, st), "rnerf_forward")
    model.whole_path = False
    ret, _ = model.apply(variables, prng.PRNGKey(5), prng.PRNGKey(9), rays, False)
    assert torch.equal(of[:3 * B].view(B, 3), ret[1][0]) and torch.equal(oc[3 * B:4 * B], ret[0][1]) and torch.equal(of[6 * B:].view(B, 3), ret[1][4])
    # argument errors come back as status codes with a message, not as crashes
    assert lib.rnerf_forward(C.byref(m), None, None, B, jit.data_ptr(), u.data_ptr(), 0, None, None, oc.data_ptr(), of.data_ptr(), ws.data_ptr(), 0, st) == -1
    assert b"origins" in lib.rnerf_last_error()


def _train_setup(Nf, B, bd_cut=False, **flag_kw):
    from samplenerfro_amd import utils
    from samplenerfro_amd.train import TrainState
    model, variables, rays, pf = _scene(Nf=Nf, B=B, bd_cut=bd_cut)
    kw = dict(num_coarse_samples=16, num_fine_samples=Nf, num_path_samples=4, white_bkgd=False, bg_weight=0.025, bg_smooth_weight=1.0, bg_patch_size=8,
              use_online_sparsity=False, randomized=True, lr_delay_steps=10, max_steps=1000)
    kw.update(flag_kw)
    flags = utils.default_flags(**kw)
    rng = np.random.default_rng(7)
    ev = rng.standard_normal((8, 8, 3)).astype(F32); ev /= np.linalg.norm(ev, axis=-1, keepdims=True)
    batch = {"rays": rays, "pixels": T(rng.uniform(0, 1, (B, 3)).astype(F32)), "annealed_alpha": 0.5, "env_rays": utils.Rays(None, None, T(ev), None)}
    state = TrainState.create(model, variables, flags)
    return model, state, batch, flags


@pytest.mark.parametrize("Nf,bd_cut,bwd,tail", [(0, False, "f32", False), (24, False, "f32", False), (24, True, "tf32", False), (24, False, "f32", True)])
def test_whole_train_step_equals_the_staged_sequence(Nf, bd_cut, bwd, tail, monkeypatch):
    """rnerf_train_forward_backward + rnerf_adam_update against the stage-by-stage train_step: identical gradient bits (same kernels,
    same order, same device-drawn jitter / stratified draws), parameters after three Adam steps within float rounding."""
    from samplenerfro_amd.train import train_step
    if tail:                      # the opt-in second stream: background-MLP weight gradient by the co-resident kernel beside the NerfMLP wgrad
        from samplenerfro_amd import train as _train
        monkeypatch.setattr(_train, "_CORESIDENT_BKGD_WGRAD", True)
    out = {}
    for whole in (True, False):
        model, state, batch, flags = _train_setup(Nf, 160, bd_cut, backward_precision=bwd)
        model.whole_path = whole
        rng = np.array([1, 2], np.uint32)
        gs, losses = [], []
        for i in range(3):
            taps = None if whole else {}
            state, stats, rng = train_step(model, rng, state, batch, flags, taps=taps)
            gs.append(state.grads[:state.theta.numel()].clone() if whole else taps["grads"])
            if i == 0:
                theta1 = state.theta.clone()
            losses.append([float(stats.loss), float(stats.loss_c), float(stats.loss_bg), float(stats.loss_bg_smooth), float(stats.weight_l2), float(stats.psnr)])
        out[whole] = (gs, losses, state.theta.clone(), rng, state.step, theta1)
    # first step: same parameters in -> the NerfMLP gradient bits are identical (same kernels, same order).  The background MLP's weight
    # gradient comes from the co-resident wgrad kernel on the tail stream in the whole path (a different summation order over the rows):
    # compared to float rounding
    n_bk = 56963
    ga, gb = out[True][0][0], out[False][0][0]
    assert torch.equal(ga[:-n_bk], gb[:-n_bk])
    assert (ga[-n_bk:] - gb[-n_bk:]).abs().max().item() <= 2e-6 * gb[-n_bk:].abs().max().item()
    assert np.allclose(out[True][1], out[False][1], rtol=2e-6, atol=1e-7)
    assert np.array_equal(out[True][3], out[False][3]) and out[True][4] == out[False][4] == 3
    # same gradient bits into the first update: the two Adam implementations (rnerf_adam_update / torch elementwise) differ by rounding only
    lr0 = flags.lr_init * flags.lr_delay_mult
    assert (out[True][5] - out[False][5]).abs().max().item() < 1e-5 * lr0
    # later steps start from parameters that differ in the last bit; Adam turns noise-level gradient entries into lr-sized updates, so
    # the parameters are only compared against the sum of the learning rates (tests/test_gpu_train_2rank.py makes the same point)
    d = (out[True][2] - out[False][2]).abs().max().item()
    assert d < 3 * flags.lr_init, d
    for a, b in zip(out[True][0][1:], out[False][0][1:]):
        assert (a - b).abs().max().item() <= 2e-4 * b.abs().max().item()


def test_adam_update_matches_the_optax_formulas():
    """rnerf_adam_update against float64 numpy: weight-decay gradient, value clip, global-norm clip over theta AND the frozen variables,
    Adam with bias correction, the reference's learning-rate schedule evaluated on the device, the step counter."""
    from samplenerfro_amd import _lib, utils
    lib = _lib.load()
    n, nf = 100003, 6541
    g = np.random.default_rng(0)
    theta = g.standard_normal(n).astype(F32); mu = (0.01 * g.standard_normal(n)).astype(F32); nu = (1e-4 * g.random(n)).astype(F32)
    grads = (0.05 * g.standard_normal(n)).astype(F32); frozen = g.standard_normal(nf).astype(F32)
    for (wd, gv, gn, count) in ((0.0, 0.0, 0.0, 0), (3.0, 0.02, 0.0, 7), (3.0, 0.02, 0.5, 2500), (0.5, 0.0, 1.5, 123456)):
        a = _lib.AdamCfg()
        a.lr_init, a.lr_final, a.lr_delay_mult, a.max_steps, a.lr_delay_steps = 5e-4, 5e-6, 0.01, 200000, 2500
        a.b1, a.b2, a.eps, a.weight_decay_mult, a.grad_max_val, a.grad_max_norm, a.n_all, a.lr_override = 0.9, 0.999, 1e-8, wd, gv, gn, n + nf, 0.0
        t_th, t_mu, t_nu, t_g, t_fr = T(theta), T(mu), T(nu), T(grads), T(frozen)
        step = torch.tensor([count], dtype=torch.int32, device=DEV)
        scratch = torch.zeros(_lib.ADAM_SCRATCH_FLOATS, device=DEV)
        _lib.check(lib.rnerf_adam_update(C.byref(a), t_th.data_ptr(), t_mu.data_ptr(), t_nu.data_ptr(), t_g.data_ptr(), n, t_fr.data_ptr(), nf, step.data_ptr(),
                                         scratch.data_ptr(), _lib.current_stream()), "rnerf_adam_update")
        wd2 = 2 * wd / (n + nf)
        G = grads.astype(np.float64) + wd2 * theta.astype(np.float64)
        Fg = wd2 * frozen.astype(np.float64)
        if gv > 0:
            G = np.clip(G, -gv, gv); Fg = np.clip(Fg, -gv, gv)
        if gn > 0:
            norm = np.sqrt((G ** 2).sum() + (Fg ** 2).sum())
            G = G * min(gn / (1e-7 + norm), 1.0)
        lr = utils.learning_rate_decay(count, 5e-4, 5e-6, 200000, 2500, 0.01)
        t = count + 1
        m2 = 0.9 * mu + 0.1 * G; v2 = 0.999 * nu + 0.001 * G * G
        th2 = theta - lr * (m2 / (1 - 0.9 ** t)) / (np.sqrt(v2 / (1 - 0.999 ** t)) + 1e-8)
        assert int(step.item()) == count + 1 and float(scratch[3]) == 0.0
        assert np.abs(t_mu.cpu().numpy() - m2).max() < 1e-8 and np.abs(t_nu.cpu().numpy() - v2).max() < 1e-9
        upd, want = t_th.cpu().numpy().astype(np.float64) - theta, th2 - theta
        assert np.abs(upd - want).max() < 2e-7 + 1e-4 * np.abs(want).max(), (wd, gv, gn, count)


def test_graph_step_skips_an_update_with_nonfinite_gradients():
    """The skip of rnerf_adam_update (rnerf_adam_cfg.skip_nonfinite) is decided on the device, so it holds inside a captured graph too: a batch
    whose rows leave f16's range leaves theta / mu / nu untouched replay after replay, the step counter advances, and the count is there to read."""
    from samplenerfro_amd.graph import GraphTrainStep
    from samplenerfro_amd import synthetic as syn, utils
    B = 96
    model, state, batch, flags = _train_setup(0, B)
    lo, _ = state.segments["coarse_mlp"]
    state.theta[lo + 63 * 256: lo + 63 * 256 + 256] = 3.0e5            # Dense_0 biases: every first-layer activation beyond 65504
    theta0 = state.theta.clone()
    g = GraphTrainStep(model, state, flags, B, np.array([4, 2], np.uint32), env_rays=batch["env_rays"], prefetch=False)
    g.load(dict(rays=batch["rays"], pixels=batch["pixels"]))
    for k in range(4):                                                  # eager warm-up, capture, replays
        g.step()
    g.synchronize()
    assert state.nonfinite_grads() > 0 and torch.equal(state.theta, theta0) and float(state.mu.abs().max()) == 0.0 and float(state.nu.abs().max()) == 0.0
    assert int(state.step_dev.item()) == 4 and state.step == 4
    g.close()


@pytest.mark.parametrize("Nf,prefetch", [(24, True), (0, False)])
def test_graph_replay_equals_the_eager_steps(Nf, prefetch):
    """GraphTrainStep (one hipGraph launch per step, device-resident keys / step counter / schedule) against eager train_step calls fed
    the same batches: same losses and parameters after 6 steps (the first is the eager warm-up, then captured graphs alternate slots)."""
    from samplenerfro_amd.graph import GraphTrainStep
    from samplenerfro_amd.train import train_step
    from samplenerfro_amd import synthetic as syn, utils
    B, steps = 96, 6
    rays_k, pix_k = [], []
    for k in range(steps + 1):
        o, d = syn.sphere_rays(B, seed=100 + k)
        rays_k.append(utils.Rays(T(o), None, T(d), None)); pix_k.append(T(np.random.default_rng(k).uniform(0, 1, (B, 3)).astype(F32)))
    model, state, batch, flags = _train_setup(Nf, B)
    rng = np.array([4, 2], np.uint32)
    eager = []
    for k in range(steps):
        b = dict(batch, rays=rays_k[k], pixels=pix_k[k])
        state, stats, rng = train_step(model, rng, state, b, flags)
        eager.append(float(stats.loss))
    theta_e = state.theta.clone()
    model2, state2, batch2, flags2 = _train_setup(Nf, B)
    g = GraphTrainStep(model2, state2, flags2, B, np.array([4, 2], np.uint32), env_rays=batch2["env_rays"], prefetch=prefetch)
    g.load(dict(rays=rays_k[0], pixels=pix_k[0]))
    replay = []
    for k in range(steps):
        if prefetch:
            g.load_next(dict(rays=rays_k[k + 1], pixels=pix_k[k + 1]))
        elif k > 0:
            g.load(dict(rays=rays_k[k], pixels=pix_k[k]))
        replay.append(float(g.step().loss))
    g.synchronize()
    assert sum(1 for v in g.graphs.values() if isinstance(v, tuple)) == (2 if prefetch else 1)      # graphs were captured and replayed
    assert np.allclose(replay, eager, rtol=1e-6, atol=0), (replay, eager)
    assert np.array_equal(g.rng(), rng) and state2.step == steps and int(state2.step_dev.item()) == steps
    assert (state2.theta - theta_e).abs().max().item() < 1e-7
    g.close()


def test_graph_step_follows_the_annealed_alpha_ramp():
    """train.py:350-351 ramps annealed_alpha from 0 every step, and loss_bg / loss_bg_smooth are gated on annealed_alpha > 0 (train.py:92,130).
    The gate is a launch argument frozen into a captured graph: GraphTrainStep must read the batch's value every step and keep one graph per
    gate (it used to replay the constructor's value for ever).  Steps with alpha = 0, 0, 0.3, 0.6, 0, 0.9 against eager steps."""
    from samplenerfro_amd.graph import GraphTrainStep
    from samplenerfro_amd.train import train_step
    from samplenerfro_amd import synthetic as syn, utils
    B, alphas = 96, [0.0, 0.0, 0.3, 0.6, 0.0, 0.9]
    rays_k, pix_k = [], []
    for k in range(len(alphas) + 1):
        o, d = syn.sphere_rays(B, seed=300 + k)
        rays_k.append(utils.Rays(T(o), None, T(d), None)); pix_k.append(T(np.random.default_rng(50 + k).uniform(0, 1, (B, 3)).astype(F32)))
    model, state, batch, flags = _train_setup(12, B)
    rng = np.array([8, 1], np.uint32)
    eager, eager_bg = [], []
    for k, a in enumerate(alphas):
        state, stats, rng = train_step(model, rng, state, dict(batch, rays=rays_k[k], pixels=pix_k[k], annealed_alpha=a), flags)
        eager.append(float(stats.loss)); eager_bg.append((float(stats.loss_bg), float(stats.loss_bg_smooth)))
    theta_e = state.theta.clone()
    assert eager_bg[0] == (0.0, 0.0) and eager_bg[4] == (0.0, 0.0) and eager_bg[2][1] > 0 and eager_bg[5][1] > 0      # the gate really switches the smoothness term on and off (loss_bg may be 0: no ray of this scene keeps trans > 0.5)
    model2, state2, batch2, flags2 = _train_setup(12, B)
    g = GraphTrainStep(model2, state2, flags2, B, np.array([8, 1], np.uint32), env_rays=batch2["env_rays"], annealed_alpha=0.5, prefetch=True)
    g.load(dict(rays=rays_k[0], pixels=pix_k[0], annealed_alpha=alphas[0]))
    replay, replay_bg = [], []
    for k in range(len(alphas)):
        nxt = dict(rays=rays_k[k + 1], pixels=pix_k[k + 1])
        if k + 1 < len(alphas):
            nxt["annealed_alpha"] = alphas[k + 1]
        g.load_next(nxt)
        st = g.step()
        replay.append(float(st.loss)); replay_bg.append((float(st.loss_bg), float(st.loss_bg_smooth)))
    g.synchronize()
    assert np.allclose(replay, eager, rtol=1e-6, atol=0), (replay, eager)
    assert np.allclose(np.array(replay_bg), np.array(eager_bg), rtol=1e-5, atol=1e-9), (replay_bg, eager_bg)
    assert (state2.theta - theta_e).abs().max().item() < 1e-7
    assert sum(1 for v in g.graphs.values() if isinstance(v, tuple)) >= 3                   # both gates were captured (per slot as needed)
    g.close()


def test_a_replaced_schedule_that_returns_zero_is_a_zero_step():
    """a replaced lr_fn reaches rnerf_adam_update as (use_lr_override, lr_override); 0.0 — a warm-up, or the reference's own
    start_rate at count 0 (rnerf/utils.py:521) — must be a ZERO learning rate, not "no override" (the device used to test lr_override > 0)."""
    from samplenerfro_amd.train import train_step
    model, state, batch, flags = _train_setup(0, 96)
    assert getattr(model, "whole_path", False)
    th0 = state.theta.clone()
    state.lr_fn = lambda count: 0.0
    state, stats, _ = train_step(model, np.array([1, 5], np.uint32), state, batch, flags)
    torch.cuda.synchronize()
    assert torch.equal(state.theta, th0)                              # the parameters did not move ...
    assert float(state.mu.abs().max()) > 0 and state.step == 1        # ... although the step ran (moments updated, counter advanced)
    state.lr_fn = lambda count: 1e-3
    state, stats, _ = train_step(model, np.array([1, 6], np.uint32), state, batch, flags)
    assert not torch.equal(state.theta, th0)


def test_adam_update_counts_nonfinite_gradients():
    from samplenerfro_amd import _lib
    lib = _lib.load()
    n = 5000
    a = _lib.AdamCfg()
    a.lr_init, a.lr_final, a.lr_delay_mult, a.max_steps, a.lr_delay_steps = 5e-4, 5e-6, 0.01, 1000, 0
    a.b1, a.b2, a.eps, a.n_all = 0.9, 0.999, 1e-8, n
    g = torch.zeros(n, device=DEV); g[7] = float("inf"); g[4000] = float("nan"); g[4001] = -float("inf")
    th, mu, nu = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    step = torch.zeros(1, dtype=torch.int32, device=DEV); scratch = torch.zeros(_lib.ADAM_SCRATCH_FLOATS, device=DEV)
    _lib.check(lib.rnerf_adam_update(C.byref(a), th.data_ptr(), mu.data_ptr(), nu.data_ptr(), g.data_ptr(), n, None, 0, step.data_ptr(), scratch.data_ptr(),
                                     _lib.current_stream()), "rnerf_adam_update")
    assert float(scratch[3]) == 3.0


@pytest.mark.parametrize("Nf,B", [(0, 160), (24, 160), (24, 2100)])
def test_train_step_does_not_depend_on_what_its_workspace_held(Nf, B):
    """Every accumulator word of a step (range flags of the operand streams, the dgrads' row-scale references, the env-map sum) is cleared by
    ONE launch at the head of the step (csrc/mlp.hip: nerfmlp_step_zero) instead of a memset inside each producer: a workspace full of
    0xFF bytes (NaN patterns), of 0x3F bytes (finite floats) or of the previous step's values must give the same gradient bits and the same
    statistics as a zeroed one.  B = 2100 with 16 + 24 samples: 132 + 329 row tiles, the two levels NOT side by side (the coarse dgrad is
    then the second writer of the shared dY buffer and clears its own reference)."""
    from samplenerfro_amd.train import train_step
    res = []
    for fill in (0x00, 0xFF, 0x3F, None):
        model, state, batch, flags = _train_setup(Nf, B)
        rng = np.array([5, 6], np.uint32)
        if fill is None:                                     # the previous step's values: a first step on other parameters' twin
            train_step(model, np.array([9, 9], np.uint32), _train_setup(Nf, B)[1], batch, flags)
        else:
            train_step(model, rng, _train_setup(Nf, B)[1], batch, flags)          # sizes the model's cached workspace
            model._ws["train"].fill_(fill)
        state, stats, _ = train_step(model, rng, state, batch, flags)
        torch.cuda.synchronize()
        res.append((state.grads.clone(), [float(stats.loss), float(stats.loss_bg_smooth), float(stats.weight_l2)]))
    for g, st in res[1:]:
        assert torch.equal(g, res[0][0])
        assert st == res[0][1]


def test_side_streams_are_shared_by_every_model_of_the_process():
    """HIP multiplexes streams onto GPU_MAX_HW_QUEUES (4) hardware queues: one stream per role and process, not per model (models.shared_stream)."""
    from samplenerfro_amd.models import shared_stream
    a, _, _, _ = _scene(Nf=0, B=32)
    b, _, _, _ = _scene(Nf=24, B=32)
    assert a.tail_stream() is b.tail_stream() and a.comm_stream() is b.tail_stream()      # the collective is issued from the (then idle) tail stream
    assert a.tail_stream().cuda_stream != shared_stream(a.device, "march").cuda_stream
    assert shared_stream(a.device, "march") is shared_stream(a.device, "march")
