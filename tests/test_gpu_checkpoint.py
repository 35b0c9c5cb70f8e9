"""SURVEY §8f N1 on the device: a reference-format (flax msgpack) checkpoint restored, rendered through the product path (rnerf_forward) and
compared with the oracle evaluated on the very same parameter tree; export -> import -> bit-identical render; resume of a TrainState.
Reference: eval.py:124-152 (restore + graft by weight name), train.py:424-427 (save)."""
import gzip
import os

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

from oracle import ref_np as R
from samplenerfro_amd import checkpoint, models, prng, synthetic as syn, utils as U
from samplenerfro_amd.utils import Rays

HERE = os.path.dirname(os.path.abspath(__file__))
G, EXT, B, S, F, P = 24, 1.5, 192, 16, 24, 4


def _fixture_dir(tmp_path):
    d = tmp_path / "radiance"
    d.mkdir(exist_ok=True)
    (d / "checkpoint_7").write_bytes(gzip.open(os.path.join(HERE, "golden", "flax_checkpoint_7.msgpack.gz"), "rb").read())
    return str(d)


def _scene(dev, precision="f16x3"):
    ndim, nmin, nmax = [G] * 3, [-EXT] * 3, [EXT] * 3
    grid = R.conv3d_normal(syn.scale_ior(syn.sphere_grid(G, EXT, 0.6), 0.5).reshape(-1, 1), ndim, 3, 1.0).reshape(ndim)
    table = R.build_table(grid, ndim, nmin, nmax)
    model = models.NerfModel(ndim=ndim, nmin=nmin, nmax=nmax, grid=torch.from_numpy(grid).to(dev), num_coarse_samples=S, num_fine_samples=F,
                             num_path_samples=P, precision=precision)
    o, d = syn.sphere_rays(B, seed=11)
    rays = Rays(torch.from_numpy(o).to(dev), None, torch.from_numpy(d).to(dev), None)
    cfg = R.ModelConfig(ndim, nmin, nmax, num_coarse_samples=S, num_fine_samples=F, num_path_samples=P)
    return model, table, cfg, rays, o, d


def _render(model, variables, rays, jitter):
    key = prng.PRNGKey(3)
    assert getattr(model, "whole_path", False)                      # the product path: ONE rnerf_forward call
    ret, _ = model.apply(variables, key, key, rays, False, jitter=jitter)
    torch.cuda.synchronize()
    return ret


def _glorot_checkpoint_dir(tmp_path, seed=21, step=1234):
    """A weights-only checkpoint_<step> in the reference's byte format with trained-looking values (glorot kernels, N(0, 0.1) biases),
    assembled with the encoders of tests/golden/make_flax_ckpt.py — i.e. WITHOUT samplenerfro_amd.checkpoint, the reader under test."""
    import sys
    import msgpack
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_flax_ckpt as M
    rng = np.random.default_rng(seed)

    def arr(shape, salt):
        if len(shape) == 2:
            lim = np.sqrt(6.0 / (shape[0] + shape[1]))
            return M.ext_array(rng.uniform(-lim, lim, shape).astype(np.float32))
        return M.ext_array((0.1 * rng.standard_normal(shape)).astype(np.float32))

    state = {"step": M.ext_array(np.asarray(step, np.int32)), "params": {"params": M.tree(arr)}}
    d = tmp_path / "glorot"
    d.mkdir(exist_ok=True)
    (d / ("checkpoint_%d" % step)).write_bytes(msgpack.packb(state, use_bin_type=True))
    return str(d)


def test_restored_checkpoint_renders_like_the_oracle_on_the_same_tree(tmp_path):
    """eval.py:124-131 on the device: restore -> variables -> ONE rnerf_forward call -> the oracle evaluated on the very same tree."""
    dev = torch.device("cuda:0")
    d = _glorot_checkpoint_dir(tmp_path)
    pretrain = checkpoint.restore_checkpoint(d)                     # eval.py:125
    assert int(pretrain["step"]) == 1234
    tree = checkpoint.find_params(pretrain)                         # pretrain["params"]["params"] (eval.py:128-131)
    variables = checkpoint.variables_from_checkpoint(d, dev)
    model, table, cfg, rays, o, dd = _scene(dev)
    jitter = np.arange(0, S * P, P) + P // 2
    ret = _render(model, variables, rays, jitter)
    otree = {k: tree[k] for k in ("coarse_mlp", "fine_mlp", "bkgd_mlp")}
    oret, _ = R.nerf_forward(cfg, otree, table, o, dd, jitter)
    for lvl in range(2):
        rgb = ret[lvl][0].cpu().numpy()
        assert np.isfinite(rgb).all()
        assert float(np.abs(rgb - oret[lvl][0]).max()) < 1e-5      # (north_star's contract is 1e-4)
        assert float(np.abs(ret[lvl][2].cpu().numpy() - oret[lvl][2]).max()) < 1e-5
    assert float(ret[1][0].std()) > 1e-3                            # a non-trivial image


def test_committed_byte_fixture_renders_within_fp32_conditioning(tmp_path):
    """tests/golden/flax_checkpoint_7.msgpack.gz through the same path.  Its arithmetic-pattern weights make an ILL-CONDITIONED network (raw
    outputs of magnitude ~2000: the fp32 oracle itself is 2e-2 away from the fp64 oracle in colour), so the device is held to the
    conditioning of fp32 arithmetic on this tree — no further from the fp64 oracle than twice the fp32 oracle is — and to the exact-fp32
    on-device arbiter (precision "f32", csrc/mlp_f32.hip)."""
    dev = torch.device("cuda:0")
    d = _fixture_dir(tmp_path)
    tree = checkpoint.find_params(checkpoint.restore_checkpoint(d))
    variables = checkpoint.variables_from_checkpoint(d, dev)
    model, table, cfg, rays, o, dd = _scene(dev)
    jitter = np.arange(0, S * P, P) + P // 2
    ret = _render(model, variables, rays, jitter)
    otree = {k: tree[k] for k in ("coarse_mlp", "fine_mlp", "bkgd_mlp")}
    o32, _ = R.nerf_forward(cfg, otree, table, o, dd, jitter)
    o64, _ = R.nerf_forward(cfg, otree, table.astype(np.float64), o, dd, jitter, dtype=np.float64)
    ret32 = _render(_scene(dev, "f32")[0], variables, rays, jitter)   # RNERF_PREC_F32: the same scene through the arbiter kernel
    for lvl in range(2):
        rgb = ret[lvl][0].cpu().numpy().astype(np.float64)
        cond = float(np.abs(o32[lvl][0] - o64[lvl][0]).max())
        err = float(np.abs(rgb - o64[lvl][0]).max())
        arb = float(np.abs(ret32[lvl][0].cpu().numpy() - o64[lvl][0]).max())
        print(f"level {lvl}: |fp32 oracle - fp64 oracle| = {cond:.2e}, |f16x3 - fp64 oracle| = {err:.2e}, |f32 arbiter - fp64 oracle| = {arb:.2e}")
        assert np.isfinite(rgb).all() and cond > 1e-4               # (the fixture really is ill-conditioned)
        assert err <= 2 * cond and arb <= 2 * cond


def test_export_import_round_trip_renders_the_same_bits(tmp_path):
    dev = torch.device("cuda:0")
    pf = syn.init_params_flat(5, fine=True, bias_scale=0.1)
    variables = models.make_variables({**{k: torch.from_numpy(v).to(dev) for k, v in pf.items()}, "so3_mlp": torch.zeros(65411, device=dev)})
    model, _table, _cfg, rays, _o, _d = _scene(dev)
    jitter = np.arange(0, S * P, P) + P // 2
    a = _render(model, variables, rays, jitter)
    p = checkpoint.save_checkpoint(str(tmp_path / "radiance"), variables, 1000)            # train.py:424-427
    assert p.endswith("checkpoint_1000")
    back = checkpoint.variables_from_checkpoint(str(tmp_path / "radiance"), dev)
    b = _render(model, back, rays, jitter)
    for lvl in range(2):
        for x, y in zip(a[lvl], b[lvl]):
            assert torch.equal(x, y)
    # eval.py:124-152: graft by weight name into fresh variables, then render
    fresh = models.make_variables({**{k: torch.from_numpy(v).to(dev) for k, v in syn.init_params_flat(9, fine=True).items()},
                                   "so3_mlp": torch.zeros(65411, device=dev)})
    grafted, step = checkpoint.graft_pretrained(fresh, str(tmp_path), "radiance", F)
    assert step == 1000
    c = _render(model, grafted, rays, jitter)
    assert torch.equal(c[1][0], a[1][0])


def test_train_state_resumes_from_a_reference_checkpoint(tmp_path):
    """train.py:322: restore -> the step counter, the parameters and one optimisation step on the restored state."""
    from samplenerfro_amd.train import TrainState, train_step
    dev = torch.device("cuda:0")
    d = _fixture_dir(tmp_path)
    model, _table, _cfg, rays, _o, _d = _scene(dev)
    pf = syn.init_params_flat(5, fine=True)
    variables = models.make_variables({**{k: torch.from_numpy(v).to(dev) for k, v in pf.items()}, "so3_mlp": torch.zeros(65411, device=dev)})
    flags = U.default_flags(num_coarse_samples=S, num_fine_samples=F, num_path_samples=P, white_bkgd=False, bg_weight=0.025, bg_smooth_weight=0.0,
                            use_online_sparsity=False, randomized=False)
    state = TrainState.create(model, variables, flags).restore_flax(checkpoint.restore_checkpoint(d))
    assert state.step == 7
    want = checkpoint.variables_from_checkpoint(d, dev)
    for name, (lo, hi) in state.segments.items():
        assert torch.equal(state.theta[lo:hi], want["flat"][name])
    pix = torch.from_numpy(np.random.default_rng(2).uniform(0, 1, (B, 3)).astype(np.float32)).to(dev)
    state, stats, _ = train_step(model, prng.PRNGKey(0), state, {"rays": rays, "pixels": pix, "annealed_alpha": 0.5})
    torch.cuda.synchronize()
    assert state.step == 8 and int(state.step_dev.item()) == 8 and np.isfinite(float(stats.loss))
