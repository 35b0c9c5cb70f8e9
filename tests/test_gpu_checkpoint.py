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


def _scene(dev):
    ndim, nmin, nmax = [G] * 3, [-EXT] * 3, [EXT] * 3
    grid = R.conv3d_normal(syn.scale_ior(syn.sphere_grid(G, EXT, 0.6), 0.5).reshape(-1, 1), ndim, 3, 1.0).reshape(ndim)
    table = R.build_table(grid, ndim, nmin, nmax)
    model = models.NerfModel(ndim=ndim, nmin=nmin, nmax=nmax, grid=torch.from_numpy(grid).to(dev), num_coarse_samples=S, num_fine_samples=F,
                             num_path_samples=P, precision="f16x3")
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


def test_restored_checkpoint_renders_like_the_oracle_on_the_same_tree(tmp_path):
    dev = torch.device("cuda:0")
    d = _fixture_dir(tmp_path)
    pretrain = checkpoint.restore_checkpoint(d)                     # eval.py:125
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
    # the fixture's weights are a non-trivial network: the colours are not a constant
    assert float(ret[1][0].std()) > 1e-4


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
