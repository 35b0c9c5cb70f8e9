"""One process per GPU: ray sharding and the (only) collectives of the path.

* Forward / eval: rays are independent; each rank renders a contiguous block of image rows against its own replica of
  the table and the weights.  The reference's `all_gather` (eval.py:96) exists only because pmap must return identical
  values; here results are gathered to every rank only if the caller asks (`gather=True`).
* Training: one all-reduce(mean) of the flat gradient buffers (+ the 13 Stats scalars in the tail) per step replaces
  `jax.lax.pmean(grads)` / `pmean(stats)` (train.py:166-167).
Backend: "nccl" (= RCCL over xGMI) on GPUs, "gloo" in the CPU tests.
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist

from . import prng
from .utils import Rays, namedtuple_map


def init(backend: Optional[str] = None) -> Tuple[int, int]:
    """Initialise torch.distributed from the torchrun environment; returns (rank, world). No-op for world == 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if torch.cuda.is_available():
        # one process per GPU: bind before anything allocates (NerfModel defaults to the current device; RCCL needs distinct devices)
        torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1))
    if (world > 1 or _force()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world


def _force() -> bool:
    """RNERF_FORCE_DIST=1: bring the process group up and issue every collective of the path even with ONE rank — how the RCCL
    code path (group init on the device, the asynchronous gradient all-reduce and its stream hand-over, barrier, max) is executed on a
    one-GPU box (tests/test_gpu_rccl.py).  Arithmetically a no-op: the sum over one rank divided by one."""
    return os.environ.get("RNERF_FORCE_DIST") == "1"


def _skip() -> bool:
    """RNERF_SKIP_ALLREDUCE=1 (bench.py's `collectives.exposed_us` only): the gradient exchange is left out, so that the step can be timed
    with and without it.  The replicas' parameters diverge: never set outside a timing loop."""
    return os.environ.get("RNERF_SKIP_ALLREDUCE") == "1"


def active() -> bool:
    """True when the collectives of the path are to be issued: more than one rank, or a forced one-rank group."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _force())


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_bounds(n: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced [lo, hi) slice of n items for `rank` (first n % world ranks get one more)."""
    q, r = divmod(n, world_size)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_rays(rays: Rays, world_size: int, rank: int) -> Rays:
    """This rank's contiguous slice of a [B, ...] ray batch (utils.shard, rnerf/utils.py:531-534, without the reshape)."""
    lo, hi = shard_bounds(rays.origins.shape[0], world_size, rank)
    return namedtuple_map(lambda r: None if r is None else r[lo:hi], rays)


def allreduce_mean_(buffers: Sequence[torch.Tensor], extra: Optional[torch.Tensor] = None) -> None:
    """In-place mean over ranks of the flat gradient buffers (train.py:166) and, if given, the stats vector (:167).

    All buffers are flattened into ONE contiguous tensor so a step costs a single all-reduce (5.26 MB for the reference
    network: latency-bound over xGMI, so fewer, larger messages win)."""
    rank, w = world()
    if not active() or _skip():
        return
    parts = [b.reshape(-1) for b in buffers] + ([extra.reshape(-1)] if extra is not None else [])
    if len(parts) == 1 and parts[0].is_contiguous():             # the train step keeps gradients + stats in one buffer already
        dist.all_reduce(parts[0], op=dist.ReduceOp.SUM)
        parts[0] /= w
        return
    flat = torch.cat(parts)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat /= w
    off = 0
    for p in parts:
        p.copy_(flat[off:off + p.numel()])
        off += p.numel()


def allreduce_begin(buf: torch.Tensor, force: bool = False):
    """Start the SUM all-reduce of a contiguous flat buffer without blocking the issuing stream (returns None on one rank).
    The train step starts the NerfMLP gradients (95 % of the bytes) right behind the last wgrad, so the collective runs beside
    the background-MLP backward and the loss tail instead of after them.  force: issue the collective even in a one-rank group
    (tests/test_gpu_rccl.py drives RCCL itself that way on a one-GPU box)."""
    if not (active() or (force and dist.is_available() and dist.is_initialized())) or _skip():
        return None
    return dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True)


def allreduce_end_mean_(handle, buf: torch.Tensor) -> None:
    """Wait for allreduce_begin(buf) on the current stream and turn the sum into the mean."""
    if handle is None:
        return
    handle.wait()
    buf /= world()[1]


def max_over_ranks(x: float, device=None) -> float:
    if not active():
        return x
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def render_image_sharded(render_fn: Callable, rays: Rays, rng, normalize_disp: bool, chunk: int = 8192, gather: bool = True):
    """utils.render_image (rnerf/utils.py:331-389) with the image ROWS sharded over ranks.

    Each rank renders rows [lo, hi) with the same key pair (the reference passes one key to all devices, eval.py:102).
    gather=False returns this rank's block (no collective at all); gather=True all-gathers the blocks so that every
    rank holds the full image (what the reference's pmap + all_gather returns).
    """
    rank, w = world()
    H, W = rays.origins.shape[:2]
    lo, hi = shard_bounds(H, w, rank)
    _unused, key_0, key_1 = prng.split(rng, 3)
    local = namedtuple_map(lambda r: None if r is None else r[lo:hi].reshape(((hi - lo) * W, -1)), rays)
    outs: List[List[torch.Tensor]] = []
    n = (hi - lo) * W
    for i in range(0, n, chunk):
        chunk_rays = namedtuple_map(lambda r: None if r is None else r[i:i + chunk], local)
        outs.append(list(render_fn(key_0, key_1, chunk_rays)[0][-1][:3]))
    if n > 0:
        rgb, distance, acc = [torch.cat(r, dim=0) for r in zip(*outs)]
    else:
        dev = rays.origins.device
        rgb, distance, acc = torch.empty((0, 3), device=dev), torch.empty((0,), device=dev), torch.empty((0,), device=dev)
    rgb = rgb.reshape(hi - lo, W, 3); distance = distance.reshape(hi - lo, W, 1); acc = acc.reshape(hi - lo, W, 1)
    if gather and w > 1:
        res = []
        rows_max = shard_bounds(H, w, 0)[1]
        for t in (rgb, distance, acc):
            pad = torch.zeros((rows_max,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
            pad[:hi - lo] = t
            bufs = [torch.empty_like(pad) for _ in range(w)]
            dist.all_gather(bufs, pad)
            res.append(torch.cat([bufs[k][:shard_bounds(H, w, k)[1] - shard_bounds(H, w, k)[0]] for k in range(w)], dim=0))
        rgb, distance, acc = res
    if normalize_disp:
        if distance.numel() > 0:
            mn, mx = distance.min(), distance.max()
        else:                               # a rank without rows (H < world) still takes part in the MIN / MAX reduction below
            mn = torch.tensor(float("inf"), device=distance.device); mx = torch.tensor(float("-inf"), device=distance.device)
        if gather is False and w > 1:
            dist.all_reduce(mn, op=dist.ReduceOp.MIN); dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        distance = (distance - mn) / (mx - mn)
    return rgb, distance, acc
