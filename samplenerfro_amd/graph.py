"""One launch graph per training step (hipGraph through the C ABI: rnerf_graph_begin / _end / _launch).

The reference's train step is ONE XLA executable per step (`jax.pmap(train_step)`, train.py:239-243).  The counterpart
here: the whole step — key split, jitter and stratified draws (device-resident jax.random keys), march, both MLP levels, compositing,
loss, every backward kernel, gradient clipping and Adam with the learning-rate schedule evaluated on the device — is captured once and
replayed with one `hipGraphLaunch` per step.  At the reference's own operating point (1024 rays, 64 + 128 samples:
configs/example.yaml:8-9,20) a step is launch-bound when the host issues its ~45 kernels one by one; replayed it is bound by the kernels.

    g = GraphTrainStep(model, state, flags, batch_size, rng, env_rays=...)
    g.load(batch0)                      # rays / pixels of the first step -> static slot
    for batch in loader:
        g.load_next(batch)              # rays of the step after: their march runs on a side branch of the CURRENT step's graph
        stats = g.step()

Two graphs alternate (A reads slot 0 and marches slot 1, B the reverse), so the march of step k+1 — which reads neither the parameters
nor anything of step k — overlaps the tail of step k inside the graph, exactly like the side-stream prefetch of the eager path.
With more than one rank the step is two graphs around the gradient all-reduce (RCCL): [forward + backward] and [clip + Adam].
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Any, Dict, Optional

import numpy as np
import torch

from . import _lib, distributed, ops
from .train import TrainState, _bump, adam_cfg, frozen_sq_of, train_cfg
from .utils import Rays, Stats


class GraphTrainStep:
    def __init__(self, model, state: TrainState, flags, batch_size: int, rng, env_rays: Optional[Rays] = None, annealed_alpha: float = 0.5,
                 prefetch: bool = True):
        if not flags.stage.startswith("radiance"):
            raise NotImplementedError("GraphTrainStep: the radiance stages (stage all* reads a device counter back every step)")
        if state._lr_fn_default is not state.lr_fn:
            raise ValueError("GraphTrainStep evaluates the reference's learning-rate schedule on the device; state.lr_fn was replaced")
        self.lib = _lib.load()
        self.model, self.state, self.flags = model, state, flags
        self.B = int(batch_size)
        dev = state.theta.device
        self.dev = dev
        self.annealed = float(annealed_alpha)
        self.prefetch = bool(prefetch)
        B, N = self.B, model.num_samples
        z3 = lambda: torch.zeros((B, 3), dtype=torch.float32, device=dev)
        self.origins, self.viewdirs, self.pixels = [z3(), z3()], [z3(), z3()], [z3(), z3()]
        self.path_pd = [torch.empty((N, B, 4), dtype=torch.float32, device=dev) for _ in range(2 if prefetch else 1)]
        self.path_dr = [torch.empty((N, B, 4), dtype=torch.float32, device=dev) for _ in range(2 if prefetch else 1)]
        self.env = None
        if flags.bg_smooth_weight > 0:
            if env_rays is None:
                raise ValueError("bg_smooth_weight > 0 needs env_rays")
            self.env = ops._chk(env_rays.viewdirs.reshape(-1, 3).clone(), "env_rays.viewdirs")
        self.rng_state = torch.from_numpy(np.asarray(rng, np.uint32).reshape(2).view(np.int32).copy()).to(dev)
        self.keys4 = torch.zeros(4, dtype=torch.int32, device=dev)
        self.m = model.c_model()
        # annealed_alpha of the batch in each static slot (train.py:350-351 ramps it every step; loss_bg / loss_bg_smooth are gated on
        # annealed_alpha > 0, train.py:92,130).  In the radiance stages only that gate reaches the kernels, as launch arguments frozen into a
        # captured graph: one cfg + one graph per (slot, gate, frozen-variables key), chosen per step from the batch's value.
        self.alpha = [self.annealed, self.annealed]
        self.cfgs: Dict[Any, Any] = {}
        self.c = self._cfg(self.annealed)
        self.a = adam_cfg(state, flags, None)
        self.ws = torch.empty(self.lib.rnerf_train_workspace_bytes(C.byref(self.m), C.byref(self.c), B), dtype=torch.uint8, device=dev)
        from .models import shared_stream      # (one stream per role and process: hardware queues are few)
        self.main = shared_stream(dev, "graph")
        self.side = shared_stream(dev, "march")
        self.slot = 0                    # the slot the NEXT step() trains on
        self.graphs: Dict[Any, Any] = {}
        self._marched = False
        self.split = distributed.active()       # two graphs around the all-reduce when the collective is live

    def _cfg(self, annealed: float):
        """rnerf_train_cfg for this gate and the current frozen variables (kept alive: captured graphs were issued from it)."""
        fkey = frozen_sq_of(self.state, self.state.variables)[2]
        key = (annealed > 0, fkey)
        if key not in self.cfgs:
            self.cfgs[key] = train_cfg(self.model, self.state, self.flags, annealed)
        return self.cfgs[key]

    def set_annealed(self, annealed_alpha: float) -> None:
        """annealed_alpha of the batch the next step() trains on (batches given to load / load_next carry their own "annealed_alpha")."""
        self.alpha[self.slot] = float(annealed_alpha)

    # ---- data ----------------------------------------------------------------------------------------------------------------------
    def _put(self, slot: int, batch) -> None:
        rays: Rays = batch["rays"]
        if batch.get("annealed_alpha") is not None:
            self.alpha[slot] = float(np.asarray(batch["annealed_alpha"]).reshape(-1)[0])
        with torch.cuda.stream(self.main):
            self.origins[slot].copy_(rays.origins, non_blocking=True)
            self.viewdirs[slot].copy_(rays.viewdirs, non_blocking=True)
            if batch.get("pixels") is not None:
                self.pixels[slot].copy_(batch["pixels"][..., :3], non_blocking=True)

    def load(self, batch) -> None:
        """The batch the next step() trains on (its march has not been prefetched: it runs eagerly before the first replay)."""
        self.main.wait_stream(torch.cuda.current_stream())
        self._put(self.slot, batch)
        self._marched = False

    def load_next(self, batch) -> None:
        """The batch of the step AFTER the next step(): its rays are marched on the side branch of the next step's graph."""
        self.main.wait_stream(torch.cuda.current_stream())
        self._put(1 - self.slot, batch)

    def next_buffers(self) -> Dict[str, torch.Tensor]:
        """The static buffers of the step AFTER the next step() — {"origins", "viewdirs", "pixels"}, [B, 3] each — for a loader that
        writes its batch in place (on a stream ordered before the next step(), e.g. the current one followed by order_after_loader())
        instead of going through load_next()'s three copies."""
        k = 1 - self.slot if self.prefetch else self.slot
        return {"origins": self.origins[k], "viewdirs": self.viewdirs[k], "pixels": self.pixels[k]}

    def order_after_loader(self) -> None:
        """Order the next step() behind whatever the current stream has written into next_buffers()."""
        self.main.wait_stream(torch.cuda.current_stream())

    # ---- the step ------------------------------------------------------------------------------------------------------------------
    def _march(self, slot: int, stream) -> None:
        k = slot if self.prefetch else 0
        m = self.model
        _lib.check(self.lib.rnerf_march(m.table.data_ptr(), C.byref(m.spec), self.origins[slot].data_ptr(), self.viewdirs[slot].data_ptr(), self.B,
                                        m.near, m.far, m.num_samples, self.path_pd[k].data_ptr(), self.path_dr[k].data_ptr(), None, None, stream),
                   "rnerf_march")

    def _issue_front(self, slot: int) -> None:
        """key split + forward + backward of `slot` on main; the march of the other slot on the side branch."""
        lib, st, sd = self.lib, self.main.cuda_stream, self.side.cuda_stream
        k = slot if self.prefetch else 0
        _lib.check(lib.rnerf_rng_split3(self.rng_state.data_ptr(), self.keys4.data_ptr(), st), "rnerf_rng_split3")
        if not self.prefetch:
            self._march(slot, st)
        nxt = None
        if self.prefetch:          # the march of the other slot: forked behind the last wgrad inside the call, joined by the caller
            nxt = _lib.Prefetch(self.origins[1 - slot].data_ptr(), self.viewdirs[1 - slot].data_ptr(), self.path_pd[1 - slot].data_ptr(),
                                self.path_dr[1 - slot].data_ptr(), sd, int(getattr(self.model, "march_beside_wgrad", True)))
        _lib.check(lib.rnerf_train_forward_backward(C.byref(self.m), C.byref(self.c), self.state.theta.data_ptr(), self.origins[slot].data_ptr(),
                                                    self.viewdirs[slot].data_ptr(), self.pixels[slot].data_ptr(), _lib.ptr(self.env), self.B,
                                                    self.keys4.data_ptr(), None, None, 0, self.path_pd[k].data_ptr(), self.path_dr[k].data_ptr(),
                                                    self.state.grads.data_ptr(), self.ws.data_ptr(), 0, C.byref(nxt) if nxt is not None else None, st),
                   "rnerf_train_forward_backward")

    def _issue_back(self) -> None:
        s = self.state
        fs = frozen_sq_of(s, s.variables)
        frozen = s.variables["flat"].get("so3_mlp") if fs[1] > 0 else None
        _lib.check(self.lib.rnerf_adam_update(C.byref(self.a), s.theta.data_ptr(), s.mu.data_ptr(), s.nu.data_ptr(), s.grads.data_ptr(), s.theta.numel(),
                                              _lib.ptr(frozen), fs[1], s.step_dev.data_ptr(), s.adam_scratch.data_ptr(), self.main.cuda_stream),
                   "rnerf_adam_update")

    def _capture(self, slot: int):
        lib, st, sd = self.lib, self.main.cuda_stream, self.side.cuda_stream
        out = C.c_void_p()
        if not self.split:
            _lib.check(lib.rnerf_graph_begin(st), "rnerf_graph_begin")
            self._issue_front(slot)
            self._issue_back()
            if self.prefetch:
                _lib.check(lib.rnerf_join(st, sd), "rnerf_join")
            _lib.check(lib.rnerf_graph_end(st, C.byref(out)), "rnerf_graph_end")
            return (out.value, None)
        _lib.check(lib.rnerf_graph_begin(st), "rnerf_graph_begin")
        self._issue_front(slot)
        if self.prefetch:
            _lib.check(lib.rnerf_join(st, sd), "rnerf_join")
        _lib.check(lib.rnerf_graph_end(st, C.byref(out)), "rnerf_graph_end")
        back = C.c_void_p()
        _lib.check(lib.rnerf_graph_begin(st), "rnerf_graph_begin")
        self._issue_back()
        _lib.check(lib.rnerf_graph_end(st, C.byref(back)), "rnerf_graph_end")
        return (out.value, back.value)

    def _warm(self, slot: int) -> None:
        """One eager pass before the first capture: lazy one-time initialisation inside the launchers (kernel attributes) must not
        happen under capture.  It is a real step on real data — state advances exactly like a replay."""
        self._issue_front(slot)
        if self.prefetch:
            _lib.check(self.lib.rnerf_join(self.main.cuda_stream, self.side.cuda_stream), "rnerf_join")
        if self.split:
            with torch.cuda.stream(self.main):
                distributed.allreduce_mean_([self.state.grads])
        self._issue_back()

    def step(self, annealed_alpha: Optional[float] = None) -> Stats:
        s, slot = self.state, self.slot
        if annealed_alpha is not None:
            self.alpha[slot] = float(annealed_alpha)
        self.annealed = self.alpha[slot]
        # everything the caller's stream did to the parameters, the step counter or the static buffers since the last step (restore, an
        # eval render, logging copies) is ordered before this step's in-place update
        self.main.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(self.main):
            s.sync_step_counter()
        self.c = self._cfg(self.annealed)
        if not self._marched and self.prefetch:
            self._march(slot, self.main.cuda_stream)            # first step (or after load()): nothing prefetched this slot's path
            self._marched = True
        key = (slot, self.annealed > 0, frozen_sq_of(s, s.variables)[2])
        if not self.graphs:
            self._warm(slot)                                    # the very first step runs eagerly (and warms every launcher)
            self.graphs["warm"] = True
        else:
            if key not in self.graphs:
                self.graphs[key] = self._capture(slot)
            front, back = self.graphs[key]
            _lib.check(self.lib.rnerf_graph_launch(front, self.main.cuda_stream), "rnerf_graph_launch")
            if back is not None:
                with torch.cuda.stream(self.main):
                    distributed.allreduce_mean_([s.grads])
                _lib.check(self.lib.rnerf_graph_launch(back, self.main.cuda_stream), "rnerf_graph_launch")
        s.step += 1
        s._step_dev_value = s.step
        _bump(s.theta)
        # whoever reads the parameters / the stats next does so on the caller's stream: order it behind this step (no host wait)
        torch.cuda.current_stream().wait_stream(self.main)
        if self.prefetch:
            self.slot = 1 - slot
        n = s.theta.numel()
        s8 = s.grads[n:]
        two = self.model.num_fine_samples > 0
        return Stats(loss=s8[0], psnr=s8[6], loss_c=s8[1], psnr_c=(s8[7] if two else 0.0), weight_l2=s8[4], loss_sp=0.0, loss_nrm=0.0,
                     annealing_rate=self.annealed, coarse_alpha_target=0.0, fine_alpha_target=0.0, loss_bg=s8[2],
                     loss_bg_c=0.0, loss_bg_smooth=s8[3])

    def synchronize(self) -> None:
        self.main.synchronize()

    def rng(self) -> np.ndarray:
        """The current jax.random key of the training loop (device -> host; synchronises)."""
        return self.rng_state.cpu().numpy().view(np.uint32).copy()

    def close(self) -> None:
        self.main.synchronize()
        for k, v in list(self.graphs.items()):
            if isinstance(v, tuple):
                for g in v:
                    if g:
                        self.lib.rnerf_graph_destroy(g)
        self.graphs.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
