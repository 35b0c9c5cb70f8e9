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


DEFAULT_TIMEOUT_S = 300.0      # a rank that never shows up / never joins a collective fails the others after this long, it never hangs them


def init(backend: Optional[str] = None, timeout_s: Optional[float] = None) -> Tuple[int, int]:
    """Initialise torch.distributed from the torchrun environment; returns (rank, world). No-op for world == 1.

    timeout_s bounds the rendezvous and every collective (gloo: the operation raises; RCCL: the watchdog aborts the
    communicator and the process exits non-zero), so a rank that died takes the job down instead of hanging it.  On RCCL the
    group is bound to this rank's device (`device_id`): the communicator is created eagerly — a mis-set topology fails here,
    at start-up, not inside the first step's all-reduce — and barriers know their device."""
    import datetime
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = 0
    if torch.cuda.is_available():
        # one process per GPU: bind before anything allocates (NerfModel defaults to the current device; RCCL needs distinct devices)
        local = int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
        torch.cuda.set_device(local)
    if (world > 1 or _force()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kw = {"timeout": datetime.timedelta(seconds=DEFAULT_TIMEOUT_S if timeout_s is None else float(timeout_s))}
        if backend == "nccl":
            lws = int(os.environ.get("LOCAL_WORLD_SIZE", world))
            if lws > torch.cuda.device_count():
                raise RuntimeError(f"RCCL needs one device per rank: {lws} ranks on this node, {torch.cuda.device_count()} visible device(s)")
            kw["device_id"] = torch.device("cuda", local)
        dist.init_process_group(backend, rank=rank, world_size=world, **kw)
    return rank, world


_FORCE = False      # process-wide switches set by the two functions below (tests and bench.py only); the product reads no environment for them
_SKIP = False


def force_single_rank_group(on: bool = True) -> None:
    """Bring the process group up and issue every collective of the path even with ONE rank — how the RCCL code path (group init on
    the device, the asynchronous gradient all-reduce and its stream hand-over, barrier, max) is executed on a one-GPU box
    (tests/test_gpu_rccl.py, bench.py --force-dist).  Arithmetically a no-op: the sum over one rank divided by one.  Call before init()."""
    global _FORCE
    _FORCE = bool(on)


class skip_allreduce:
    """Context manager for bench.py's `collectives.exposed_us` only: inside it the gradient exchange is left out, so that the step can be
    timed with and without it.  The replicas' parameters diverge: never use outside a timing loop."""

    def __enter__(self):
        global _SKIP
        _SKIP = True

    def __exit__(self, *exc):
        global _SKIP
        _SKIP = False


def _force() -> bool:
    return _FORCE


def _skip() -> bool:
    return _SKIP


_GROUP = None       # the ranks the path's collectives run over: None = every rank (the default group); set by use_group()


class use_group:
    """Run the path's collectives (gradient all-reduce, barrier, max) over a sub-group of the ranks: bench.py's in-run scaling curve
    times the same step on the first 1, 2, 4, ... ranks of ONE launch while the others wait.  `group` comes from
    torch.distributed.new_group (every rank of the job must take part in creating it); a rank outside the group must not issue collectives."""

    def __init__(self, group):
        self.group = group

    def __enter__(self):
        global _GROUP
        self.prev, _GROUP = _GROUP, self.group
        return self

    def __exit__(self, *exc):
        global _GROUP
        _GROUP = self.prev


def active() -> bool:
    """True when the collectives of the path are to be issued: more than one rank, or a forced one-rank group."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size(_GROUP) > 1 or _force())


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(_GROUP), dist.get_world_size(_GROUP)
    return 0, 1


def barrier() -> None:
    if active():
        dist.barrier(group=_GROUP)


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
        dist.all_reduce(parts[0], op=dist.ReduceOp.SUM, group=_GROUP)
        parts[0] /= w
        return
    flat = torch.cat(parts)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=_GROUP)
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
    return dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True, group=_GROUP)


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
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=_GROUP)
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
            dist.all_gather(bufs, pad, group=_GROUP)
            res.append(torch.cat([bufs[k][:shard_bounds(H, w, k)[1] - shard_bounds(H, w, k)[0]] for k in range(w)], dim=0))
        rgb, distance, acc = res
    if normalize_disp:
        if distance.numel() > 0:
            mn, mx = distance.min(), distance.max()
        else:                               # a rank without rows (H < world) still takes part in the MIN / MAX reduction below
            mn = torch.tensor(float("inf"), device=distance.device); mx = torch.tensor(float("-inf"), device=distance.device)
        if gather is False and w > 1:
            dist.all_reduce(mn, op=dist.ReduceOp.MIN, group=_GROUP); dist.all_reduce(mx, op=dist.ReduceOp.MAX, group=_GROUP)
        distance = (distance - mn) / (mx - mn)
    return rgb, distance, acc
