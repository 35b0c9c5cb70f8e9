"""the batch sampler of the training loop with the views resident on the device.

Reference: `rnerf/datasets.py` — `Dataset.__init__ / run / __next__` (:59-119: a daemon thread that keeps a queue of 3 batches),
`_train_init` (:123-145) and `_next_train` (:151-205).  The reference keeps the rays of every view as host arrays, draws ray indices with
numpy's global generator and indexes images and rays on the host; every batch is then shipped to the devices (`utils.shard`).

Here only the DRAW stays on the host — the same numpy calls in the same order, so a seeded run draws the reference's indices — in the same
kind of prefetch thread (a draw without replacement over 640 000 pixels is a 5 ms permutation, as in the reference).  The GATHER is one device
op (`rnerf_sample_batch`, csrc/render.hip): pixels from the resident image tensor, rays generated on the fly for exactly the drawn pixels
with the arithmetic of `rnerf_generate_rays` (bit-identical to indexing the reference's ray arrays, tests/test_gpu_batcher.py).  A training
loop then has no per-step host tensor work: 8 B per ray of indices go up, nothing comes down.
"""
from __future__ import annotations

import queue
import threading
from typing import Any, Dict, Optional

import numpy as np
import torch

from . import ops
from .utils import Rays


class DeviceBatcher(threading.Thread):
    """`Dataset(split="train")` of the reference for views that are already decoded: next(batcher) -> {"pixels", "rays", "env_rays"}.

    images [n, H, W, C] float32 in [0, 1] (C = 3; the reference composites RGBA over white / black before, datasets.py:352-361),
    camtoworlds [n, 3, 4] (or [n, 4, 4]); focal= (Blender model, :216-242) or cam_mat= (OpenCV, :486-518).  rng: the numpy generator the
    draws come from — `np.random` (the module: the reference's global generator, seed it like train.py:188 does) or a RandomState.
    prefetch=0 draws in the calling thread (deterministic interleaving with other users of the generator: the tests)."""

    def __init__(self, images, camtoworlds, *, batch_size: int, device, focal: Optional[float] = None, cam_mat=None, pixel_center: bool = True,
                 batching: str = "single_image", patch_size: int = 0, precrop_iters: int = 0, precrop_frac: float = 0.5, rng=np.random,
                 prefetch: int = 3):
        super().__init__(daemon=True)
        if batching not in ("single_image", "all_images"):
            raise NotImplementedError(f"{batching} batching strategy is not implemented.")        # datasets.py:144-145
        if (focal is None) == (cam_mat is None):
            raise ValueError("give focal= (Blender camera) or cam_mat= (OpenCV camera)")
        img = torch.as_tensor(np.asarray(images, np.float32) if not isinstance(images, torch.Tensor) else images, dtype=torch.float32)
        if img.dim() != 4:
            raise ValueError("images must be [n, H, W, C]")
        self.n_examples, self.h, self.w, self.channels = (int(v) for v in img.shape)
        self.images = img.to(device).contiguous()
        c2w = np.asarray(camtoworlds, np.float32)[:, :3, :4]
        if c2w.shape != (self.n_examples, 3, 4):
            raise ValueError("camtoworlds must be [n, 3, 4] or [n, 4, 4]")
        self.camtoworlds = torch.from_numpy(np.ascontiguousarray(c2w)).to(device)
        self.camera = dict(focal=focal, cam_mat=cam_mat, pixel_center=pixel_center)
        self.batch_size, self.batching, self.patch_size = int(batch_size), batching, int(patch_size)
        self.precrop_iters, self.precrop_frac = int(precrop_iters), float(precrop_frac)
        self.train_it = 0
        self.rng = rng
        self.device = torch.device(device)
        self._bad = torch.zeros(1, dtype=torch.int32, device=device)
        self.queue: "queue.Queue" = queue.Queue(max(int(prefetch), 1))
        self._threaded = prefetch > 0
        if self._threaded:
            self.start()

    # ---- the draw: numpy calls of _next_train, in its order --------------------------------------------------------------------------
    def _crop_coords(self):
        dH = int(self.h // 2 * self.precrop_frac)
        dW = int(self.w // 2 * self.precrop_frac)
        return np.arange(self.h * self.w).reshape(self.h, self.w)[(self.h // 2 - dH):(self.h // 2 + dH), (self.w // 2 - dW):(self.w // 2 + dW)]

    def draw(self) -> Dict[str, Any]:
        """One batch as INDICES: {"ray_indices": int64 [B] flat over (image, row, column), "env_indices": int64 [ps, ps] or None}."""
        hw = self.h * self.w
        if self.batching == "all_images":
            ray_indices = self.rng.choice(self.n_examples * hw, (self.batch_size,), replace=False)
        else:
            image_index = self.rng.randint(0, self.n_examples, ())
            if self.train_it < self.precrop_iters:
                ray_indices = self.rng.choice(self._crop_coords().reshape(-1), (self.batch_size,), replace=False)
            else:
                ray_indices = self.rng.choice(hw, (self.batch_size,), replace=False)
            ray_indices = ray_indices.astype(np.int64) + int(image_index) * hw
        env = None
        if self.patch_size > 0:
            image_index = self.rng.randint(0, self.n_examples, ())
            ps = self.patch_size
            if self.train_it < self.precrop_iters:
                coords = self._crop_coords()
                pH, pW = coords.shape
                x = self.rng.randint(low=0, high=pW - ps)
                y = self.rng.randint(low=0, high=pH - ps)
            else:
                coords = np.arange(hw).reshape(self.h, self.w)
                x = self.rng.randint(low=0, high=self.w - ps)
                y = self.rng.randint(low=0, high=self.h - ps)
            env = coords[y:(y + ps), x:(x + ps)].astype(np.int64) + int(image_index) * hw
        self.train_it += 1
        return {"ray_indices": np.ascontiguousarray(ray_indices, np.int64), "env_indices": env}

    # ---- the gather: one device op per ray set ------------------------------------------------------------------------------------------
    def gather(self, drawn: Dict[str, Any]) -> Dict[str, Any]:
        idx = torch.from_numpy(drawn["ray_indices"]).to(self.device, non_blocking=True)
        o, d, v, pix = ops.sample_batch(self.camtoworlds, self.images, idx, self.h, self.w, bad_count=self._bad, want_directions=True, **self.camera)
        env_rays = None
        if drawn["env_indices"] is not None:
            e = torch.from_numpy(np.ascontiguousarray(drawn["env_indices"].reshape(-1))).to(self.device, non_blocking=True)
            eo, ed, ev, _ = ops.sample_batch(self.camtoworlds, None, e, self.h, self.w, bad_count=self._bad, want_directions=True, **self.camera)
            shp = tuple(drawn["env_indices"].shape) + (3,)
            env_rays = Rays(eo.reshape(shp), ed.reshape(shp), ev.reshape(shp), None)
        return {"pixels": pix, "rays": Rays(o, d, v, None), "env_rays": env_rays}

    def run(self):
        while True:
            self.queue.put(self.draw())

    def __iter__(self):
        return self

    def __next__(self) -> Dict[str, Any]:
        return self.gather(self.queue.get() if self._threaded else self.draw())

    def out_of_range_indices(self) -> int:
        """Indices outside [0, n * H * W) met so far (reads a device counter: synchronises).  Always 0 for draws made here."""
        return int(self._bad.item())

    @property
    def size(self):
        return self.n_examples
