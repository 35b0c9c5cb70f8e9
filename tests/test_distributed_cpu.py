"""N > 1 host logic on CPU: two gloo processes on 127.0.0.1 (no GPU, no HIP calls)."""
import os
import socket

import numpy as np
import pytest

torch = pytest.importorskip("torch")
import torch.distributed as dist
import torch.multiprocessing as mp

from samplenerfro_amd import distributed as D, prng
from samplenerfro_amd.utils import Rays


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def fake_render_fn(key_0, key_1, rays):
    """Deterministic stand-in with the structure of model.apply: ret = [coarse, fine], 5-tuples."""
    o, d = rays.origins, rays.viewdirs
    rgb = torch.stack([o[:, 0] + d[:, 0], o[:, 1] * d[:, 1], o[:, 2] - d[:, 2]], -1) + float(key_0[1] % 7)
    dist_ = (o * d).sum(-1)
    acc = torch.sigmoid(o.sum(-1))
    lvl = (rgb, dist_, acc, acc[:, None], rgb)
    return [lvl, lvl], 0.0


def _worker(rank, world, port, H, W, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    r, w = D.init("gloo")
    assert (r, w) == (rank, world)
    g = torch.Generator().manual_seed(0)
    rays = Rays(torch.randn(H, W, 3, generator=g), None, torch.randn(H, W, 3, generator=g), None)
    rng = prng.PRNGKey(3)
    full = D.render_image_sharded(fake_render_fn, rays, rng, False, chunk=5, gather=True)
    part = D.render_image_sharded(fake_render_fn, rays, rng, False, chunk=64, gather=False)
    lo, hi = D.shard_bounds(H, world, rank)
    for f, p in zip(full, part):
        assert torch.equal(f[lo:hi], p)
    # gradient all-reduce (mean) of two flat buffers + a stats tail, one collective
    a = torch.full((1000,), float(rank + 1)); b = torch.arange(7, dtype=torch.float32) * (rank + 1); st = torch.tensor([float(rank)] * 13)
    D.allreduce_mean_([a, b], st)
    m = (1 + world) / 2.0
    assert torch.allclose(a, torch.full((1000,), m)) and torch.allclose(b, torch.arange(7, dtype=torch.float32) * m)
    assert torch.allclose(st, torch.full((13,), (world - 1) / 2.0))
    assert D.max_over_ranks(1.0 + rank) == float(world)
    # the split form the train step uses: a big part started asynchronously, a small part in between, then the wait
    big = torch.full((4096,), float(rank + 1)); small = torch.full((9,), float(10 * (rank + 1)))
    h = D.allreduce_begin(big)
    D.allreduce_mean_([small])
    D.allreduce_end_mean_(h, big)
    assert torch.allclose(big, torch.full((4096,), m)) and torch.allclose(small, torch.full((9,), 10 * m))
    if rank == 0:
        torch.save([t.clone() for t in full], tmp)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_cover_exactly():
    for n in (0, 1, 7, 800, 4096):
        for w in (1, 2, 3, 8):
            b = [D.shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(120)
def test_two_rank_render_and_allreduce(tmp_path):
    H, W, world = 9, 6, 2
    port = _free_port()
    out = str(tmp_path / "full.pt")
    mp.spawn(_worker, args=(world, port, H, W, out), nprocs=world, join=True)
    full = torch.load(out)
    # single-process result must equal the 2-rank assembled image
    from samplenerfro_amd import utils
    g = torch.Generator().manual_seed(0)
    rays = Rays(torch.randn(H, W, 3, generator=g), None, torch.randn(H, W, 3, generator=g), None)
    ref = utils.render_image(fake_render_fn, rays, prng.PRNGKey(3), False, chunk=7)
    for f, r in zip(full, ref):
        assert torch.equal(f, r.reshape(f.shape))


def _adam_worker(rank, world, port, tmp):
    """Data-parallel optimiser step: per-rank gradients -> one all-reduce (mean) -> identical Adam updates on every rank."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    D.init("gloo")
    from samplenerfro_amd.train import TrainState, _N_STATS
    n = 1001
    theta = torch.linspace(-1, 1, n)
    state = TrainState(0, theta.clone(), torch.zeros(n), torch.zeros(n), {}, {"all": (0, n)}, lambda c: 1e-2)
    for step in range(3):
        g = torch.Generator().manual_seed(100 * step + rank)
        state.grads[:n] = torch.randn(n, generator=g)
        state.grads[n:] = float(rank)                                # the stats tail
        D.allreduce_mean_([state.grads])
        assert torch.allclose(state.grads[n:], torch.full((_N_STATS,), (world - 1) / 2.0))
        state.apply_gradients(state.grads[:n])
    torch.save(state.theta.clone(), f"{tmp}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_data_parallel_adam(tmp_path):
    world, n = 2, 1001
    port = _free_port()
    base = str(tmp_path / "theta")
    mp.spawn(_adam_worker, args=(world, port, base), nprocs=world, join=True)
    t0, t1 = torch.load(base + ".0"), torch.load(base + ".1")
    assert torch.equal(t0, t1)                                       # replicas stay bit-identical
    # float64 optax.adam on the mean gradient
    th = np.linspace(-1, 1, n); mu = np.zeros(n); nu = np.zeros(n)
    for step in range(3):
        gs = [torch.randn(n, generator=torch.Generator().manual_seed(100 * step + r)).numpy().astype(np.float64) for r in range(world)]
        g = sum(gs) / world
        mu = 0.9 * mu + 0.1 * g; nu = 0.999 * nu + 0.001 * g * g
        t = step + 1
        th = th - 1e-2 * (mu / (1 - 0.9 ** t)) / (np.sqrt(nu / (1 - 0.999 ** t)) + 1e-8)
    assert np.abs(t0.numpy() - th).max() < 1e-5


def _world8_worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    assert D.init("gloo", timeout_s=60) == (rank, world)
    H, W = 800, 3                                                    # BASELINE configs[4]'s row count: 100 image rows per rank on 8
    lo, hi = D.shard_bounds(H, world, rank)
    assert hi - lo == 100 and lo == 100 * rank
    g = torch.Generator().manual_seed(1)
    rays = Rays(torch.randn(H, W, 3, generator=g), None, torch.randn(H, W, 3, generator=g), None)
    part = D.render_image_sharded(fake_render_fn, rays, prng.PRNGKey(3), False, chunk=128, gather=False)      # no collective at all
    assert part[0].shape == (100, W, 3)
    full = D.render_image_sharded(fake_render_fn, rays, prng.PRNGKey(3), True, chunk=128, gather=True)        # what pmap + all_gather returns
    # a 4096-ray training batch = 512 rays per rank, one jax.random key per rank (train.py:338-339)
    o = torch.arange(4096 * 3, dtype=torch.float32).reshape(4096, 3)
    mine = D.shard_rays(Rays(o, None, o, None), world, rank)
    assert mine.origins.shape == (512, 3) and float(mine.origins[0, 0]) == 512.0 * 3 * rank
    keys = prng.split(prng.PRNGKey(20200823), world)
    assert len({tuple(k) for k in keys.tolist()}) == world
    # the step's exchange over all eight ranks, and over the first four only (bench.py's in-run scaling curve)
    buf = torch.full((1 << 14,), float(rank))
    D.allreduce_mean_([buf])
    assert torch.allclose(buf, torch.full_like(buf, 3.5))
    sub = dist.new_group(ranks=[0, 1, 2, 3])
    if rank < 4:
        with D.use_group(sub):
            assert D.world() == (rank, 4) and D.active()
            b2 = torch.full((100,), float(rank))
            h = D.allreduce_begin(b2); D.allreduce_end_mean_(h, b2); D.barrier()
            assert torch.allclose(b2, torch.full_like(b2, 1.5)) and D.max_over_ranks(float(rank)) == 3.0
    assert D.world() == (rank, world)
    D.barrier()
    if rank == 0:
        torch.save([t.clone() for t in full], tmp)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_eight_ranks_shards_keys_and_subgroups(tmp_path):
    """World size 8 on CPU (gloo): the shard arithmetic of BASELINE configs[3] / [4] (512 rays, 100 image rows per rank), eight distinct
    per-rank keys, the gradient all-reduce over all ranks and over a sub-group, and the assembled image equal to the single-process one."""
    world = 8
    port = _free_port()
    out = str(tmp_path / "full8.pt")
    mp.spawn(_world8_worker, args=(world, port, out), nprocs=world, join=True)
    full = torch.load(out)
    from samplenerfro_amd import utils
    g = torch.Generator().manual_seed(1)
    rays = Rays(torch.randn(800, 3, 3, generator=g), None, torch.randn(800, 3, 3, generator=g), None)
    ref = utils.render_image(fake_render_fn, rays, prng.PRNGKey(3), True, chunk=97)
    for k, (f, r) in enumerate(zip(full, ref)):
        # (the stand-in's sigmoid runs through torch's CPU vector / remainder loops, which round differently for different chunk lengths: 1 ulp)
        assert torch.allclose(f, r.reshape(f.shape), rtol=0, atol=1e-6 if k == 2 else 0.0), (k, (f - r.reshape(f.shape)).abs().max())
