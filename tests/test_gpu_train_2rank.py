"""Data-parallel train_step end to end: two processes (gloo over 127.0.0.1, both on cuda:0) each own half of the rays; after the
in-step all-reduce both replicas must hold bit-identical parameters, equal (up to summation order) to one process on the whole batch."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
B, STEPS = 128, 3


def _setup(lo, hi):
    from samplenerfro_amd import models, synthetic as syn, utils
    from samplenerfro_amd.train import TrainState
    dev = torch.device("cuda:0")
    G = 24
    grid = syn.scale_ior(syn.sphere_grid(G, 1.5, 0.6), 0.5).astype(np.float32)
    flags = utils.default_flags(num_coarse_samples=8, num_fine_samples=12, num_path_samples=4, white_bkgd=False, bg_weight=0.0,
                                bg_smooth_weight=0.0, use_online_sparsity=False, lr_delay_steps=0, max_steps=1000, randomized=False)
    model, variables = models.construct_nerf(np.array([0, 7], np.uint32), None, flags, [G] * 3, [-1.5] * 3, [1.5] * 3,
                                             torch.from_numpy(grid).to(dev))
    pf = syn.init_params_flat(5, fine=True, bias_scale=0.1)
    for k in ("coarse_mlp", "fine_mlp", "bkgd_mlp"):
        variables["flat"][k].copy_(torch.from_numpy(pf[k]).to(dev))
    o, d = syn.sphere_rays(B, seed=5)
    pix = np.random.default_rng(5).uniform(0, 1, (B, 3)).astype(np.float32)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a[lo:hi])).to(dev)
    batch = {"rays": utils.Rays(T(o), None, T(d), None), "pixels": T(pix), "annealed_alpha": 0.5}
    state = TrainState.create(model, variables, flags)
    state.lr_fn = lambda c: 1e-3                      # a visible update from the first step on
    return model, state, batch


def _run(model, state, batch):
    from samplenerfro_amd.train import train_step
    rng = np.array([1, 2], np.uint32)
    jitter = np.arange(0, 32, 4) + 1                  # the same coarse jitter on every rank and step
    g0 = None
    for i in range(STEPS):
        taps = {}
        state, stats, rng = train_step(model, rng, state, batch, jitter=jitter, taps=taps)
        if i == 0:
            g0 = taps["grads"].detach().cpu()         # the (all-reduced) gradient of the first step
    return state.theta.detach().cpu(), float(stats.loss), g0


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from samplenerfro_amd import distributed as D
    D.init("gloo")
    per = B // world
    theta, loss, g0 = _run(*_setup(rank * per, (rank + 1) * per))
    torch.save({"theta": theta, "loss": loss, "g0": g0}, f"{out}.{rank}")
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_train_step_matches_single_process(tmp_path):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "rank")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert torch.equal(r0["theta"], r1["theta"])                      # replicas bit-identical after 3 steps
    assert torch.equal(r0["g0"], r1["g0"])
    theta, loss, g0 = _run(*_setup(0, B))                               # one process, whole batch
    # stats are pmean'ed: the reported loss is the mean of the two shard losses = the whole-batch mse
    assert abs(r0["loss"] - r1["loss"]) < 1e-12 and abs(r0["loss"] - loss) < 1e-5
    # mean of the two shard gradients = gradient of the whole-batch mean loss (only the summation order differs); the parameters
    # themselves are not compared: Adam normalises noise-level gradient entries, which turns 1e-9 differences into lr-sized ones
    err = (g0 - r0["g0"]).abs().max().item() / g0.abs().max().item()
    assert err < 1e-5, err
    assert (theta - r0["theta"]).abs().max().item() < 3.1e-3            # bounded by STEPS * lr


# ---- range_retry with two ranks: only rank 0's shard holds rows outside f16's range --------------------------------------------------
def _range_setup(rank, world):
    """Rays sorted by the largest x their (straight) path reaches: rank 0 gets the half that goes beyond x = 1.64 — where the doctored coarse
    network (unit 7 = relu(200 x + b), Dense_1[7 -> 3] = 200) leaves f16's range —, rank 1 the half that stays below ~1.05."""
    from samplenerfro_amd import synthetic as syn, utils
    o, d = syn.sphere_rays(B, seed=5)
    order = np.argsort(-np.maximum(o[:, 0] + 2 * d[:, 0], o[:, 0] + 6 * d[:, 0]), kind="stable")
    per = B // world
    idx = order[rank * per:(rank + 1) * per]
    model, state, _ = _setup(0, B)
    lo, _ = state.segments["coarse_mlp"]
    state.theta[lo:lo + 256] = 0.0
    state.theta[lo + 7] = 200.0
    state.theta[lo + 63 * 256 + 256 + 7 * 256 + 3] = 200.0
    pix = np.random.default_rng(5).uniform(0, 1, (B, 3)).astype(np.float32)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a[idx])).to("cuda:0")
    return model, state, {"rays": utils.Rays(T(o), None, T(d), None), "pixels": T(pix), "annealed_alpha": 0.5}


def _range_run(model, state, batch):
    from samplenerfro_amd.train import train_step
    theta0 = state.theta.clone()
    state, stats, _ = train_step(model, np.array([1, 2], np.uint32), state, batch, jitter=np.arange(0, 32, 4) + 1, range_retry=True)
    return {"theta": state.theta.detach().cpu(), "retries": state.range_retries, "loss": float(stats.loss), "loss_c": float(stats.loss_c),
            "moved": not torch.equal(state.theta, theta0)}


def _range_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    from samplenerfro_amd import distributed as D
    D.init("gloo")
    torch.save(_range_run(*_range_setup(rank, world)), f"{out}.{rank}")
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_range_retry_is_decided_by_the_reduced_gradient(tmp_path):
    """The re-run decision of train_step(range_retry=True) is read from the ALL-REDUCED gradient (rnerf_adam_update's count), so every rank takes
    it together: rank 1's own rows are inside f16's range (alone it never re-runs), rank 0's are not; both re-run, both apply the same
    update, and the replicas stay bit-identical and finite."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "range")
    mp.spawn(_range_worker, args=(2, port, out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert r0["retries"] == 1 and r1["retries"] == 1
    assert torch.equal(r0["theta"], r1["theta"]) and torch.isfinite(r0["theta"]).all() and r0["moved"] and r1["moved"]
    assert np.isfinite(r0["loss"] + r0["loss_c"]) and r0["loss"] == r1["loss"]
    alone = _range_run(*_range_setup(1, 2))                  # rank 1's shard in one process: inside the range, no re-run
    assert alone["retries"] == 0 and np.isfinite(alone["loss"] + alone["loss_c"])
    hot = _range_run(*_range_setup(0, 2))                    # rank 0's shard alone: re-run
    assert hot["retries"] == 1 and np.isfinite(hot["loss"] + hot["loss_c"])
