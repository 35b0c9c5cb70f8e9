"""SURVEY 8f N4, second half: the training batch sampler (rnerf/datasets.py:151-205 `_next_train`) with the views resident on the device.

The reference indexes HOST arrays — `self.images[image_index][ray_indices]`, `r[image_index][ray_indices]` of the rays `_generate_rays`
made for every pixel — with indices drawn from numpy's global generator.  samplenerfro_amd.datasets.DeviceBatcher draws the same indices
(same numpy calls, same order) and gathers on the device (rnerf_sample_batch: pixels from the resident image, rays generated for the drawn
pixels only).  Checked here against exactly that numpy indexing, on the reference's photograph and camera (tests/golden/example_image.npz,
cases.EXAMPLE_C2W) plus a second synthetic view: pixels and rays bit for bit, both batching modes, the pre-crop phase, the env-map patch,
both camera models.

And PINNED BY THE REFERENCE ITSELF: tests/golden/reference_numpy.npz holds batches computed by the reference's own `Dataset._next_train`
(executed from its source, with its own `utils.namedtuple_map`, on the rays of its own `_generate_rays`:
tests/golden/make_from_reference_numpy.py) — the draws of DeviceBatcher indexed into the oracle's arrays (CPU) and the device gather (GPU)
must reproduce them bit for bit."""
import math
import os
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from oracle import ref_np as R

F32 = np.float32


def _views():
    import cases
    img = np.load(os.path.join(HERE, "golden", "example_image.npz"))["rgba_sum4"]
    photo = img[..., :3].astype(F32) / F32(1020.0)                               # [400, 400, 3]
    rng = np.random.default_rng(3)
    second = rng.uniform(0, 1, photo.shape).astype(F32)
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    c2w2 = np.concatenate([q, rng.uniform(-3, 3, (3, 1))], -1).astype(F32)
    images = np.stack([photo, second])
    c2w = np.stack([np.asarray(cases.EXAMPLE_C2W, F32)[:3, :4], c2w2])
    H = W = 400
    focal = 0.5 * W / math.tan(0.5 * cases.EXAMPLE_CAMERA_ANGLE_X)
    return images, c2w, H, W, focal


def _reference_next_train(state, images, rays, *, h, w, batch_size, batching, patch_size, precrop_iters, precrop_frac, train_it):
    """rnerf/datasets.py:151-205 as it stands, on host arrays (images [n, H*W, 3] / rays of [n, H*W, 3]; all_images: flattened), drawing
    from the RandomState `state` where the reference names np.random."""
    n = images.shape[0] if batching == "single_image" else None
    if batching == "all_images":
        ray_indices = state.choice(rays[0].shape[0], (batch_size,), replace=False)
        batch_pixels = images[ray_indices]
        batch_rays = [r[ray_indices] for r in rays]
        n_examples = rays[0].shape[0] // (h * w)
        rays_img = [r.reshape(n_examples, h * w, 3) for r in rays]
    else:
        n_examples = n
        rays_img = rays
        image_index = state.randint(0, n_examples, ())
        if train_it < precrop_iters:
            dH = int(h // 2 * precrop_frac); dW = int(w // 2 * precrop_frac)
            coords = np.arange(rays[0][0].shape[0]).reshape(h, w)[(h // 2 - dH):(h // 2 + dH), (w // 2 - dW):(w // 2 + dW)]
            ray_indices = state.choice(coords.reshape(-1), (batch_size,), replace=False)
        else:
            ray_indices = state.choice(rays[0][0].shape[0], (batch_size,), replace=False)
        batch_pixels = images[image_index][ray_indices]
        batch_rays = [r[image_index][ray_indices] for r in rays]
    env = None
    if patch_size > 0:
        image_index = state.randint(0, n_examples, ())
        if train_it < precrop_iters:
            dH = int(h // 2 * precrop_frac); dW = int(w // 2 * precrop_frac)
            coords = np.arange(h * w).reshape(h, w)[(h // 2 - dH):(h // 2 + dH), (w // 2 - dW):(w // 2 + dW)]
            pH, pW = coords.shape
            x = state.randint(low=0, high=pW - patch_size); y = state.randint(low=0, high=pH - patch_size)
        else:
            coords = np.arange(h * w).reshape(h, w)
            x = state.randint(low=0, high=w - patch_size); y = state.randint(low=0, high=h - patch_size)
        ri = coords[y:(y + patch_size), x:(x + patch_size)]
        env = [r[image_index][ri] for r in rays_img]
    return batch_pixels, batch_rays, env


@pytest.mark.gpu
@pytest.mark.parametrize("batching,opencv", [("single_image", False), ("all_images", False), ("single_image", True)])
def test_device_batches_equal_the_references_host_indexing(batching, opencv):
    from samplenerfro_amd.datasets import DeviceBatcher
    images, c2w, H, W, focal = _views()
    K = [[612.3, 0, 201.7], [0, 609.8, 197.2], [0, 0, 1]]
    cam = dict(cam_mat=K) if opencv else dict(focal=focal)
    # the reference's host arrays: every ray of every view (datasets.py:216-242 / :486-518 — the oracle's generate_rays is held to them bit
    # for bit by tests/test_reference_numpy_pin.py)
    per_view = [R.generate_rays(c2w[i], H, W, pixel_center=True, **cam) for i in range(len(c2w))]
    rays = [np.stack([pv[k].reshape(-1, 3) for pv in per_view]) for k in range(3)]           # origins, directions, viewdirs: [n, H*W, 3]
    imgs = images.reshape(len(c2w), -1, 3)
    if batching == "all_images":
        rays = [r.reshape(-1, 3) for r in rays]; imgs = imgs.reshape(-1, 3)
    kw = dict(batch_size=1024, batching=batching, patch_size=16, precrop_iters=2, precrop_frac=0.5)
    bat = DeviceBatcher(images, c2w, device="cuda:0", rng=np.random.RandomState(7), prefetch=0, **kw, **cam)
    ref_state = np.random.RandomState(7)
    for it in range(4):                                                                     # two pre-crop batches, two full-image ones
        b = next(bat)
        px, rr, env = _reference_next_train(ref_state, imgs, rays, h=H, w=W, train_it=it, **kw)
        assert np.array_equal(b["pixels"].cpu().numpy(), px), it
        for got, want in zip((b["rays"].origins, b["rays"].directions, b["rays"].viewdirs), rr):
            assert np.array_equal(got.cpu().numpy(), want), it
        for got, want in zip((b["env_rays"].origins, b["env_rays"].directions, b["env_rays"].viewdirs), env):
            assert got.shape == (16, 16, 3) and np.array_equal(got.cpu().numpy(), want), it
    assert bat.out_of_range_indices() == 0


@pytest.mark.gpu
def test_the_prefetch_thread_feeds_a_training_loop_and_bad_indices_are_counted():
    from samplenerfro_amd import ops
    from samplenerfro_amd.datasets import DeviceBatcher
    images, c2w, H, W, focal = _views()
    bat = DeviceBatcher(images, c2w, device="cuda:0", focal=focal, batch_size=4096, rng=np.random.RandomState(1), prefetch=3)
    seen = [next(bat) for _ in range(5)]
    assert all(b["pixels"].shape == (4096, 3) and b["rays"].viewdirs.shape == (4096, 3) and b["env_rays"] is None for b in seen)
    assert all(bool(torch.isfinite(b["rays"].viewdirs).all()) for b in seen) and bat.out_of_range_indices() == 0
    bad = torch.zeros(1, dtype=torch.int32, device="cuda:0")
    idx = torch.tensor([0, 5, 2 * H * W, -1], dtype=torch.int64, device="cuda:0")
    ops.sample_batch(bat.camtoworlds, bat.images, idx, H, W, focal=focal, bad_count=bad)
    assert int(bad.item()) == 2


def test_draws_follow_the_references_call_order_without_a_device():
    """The index half runs anywhere: same numpy calls in the same order as _next_train -> the same generator state afterwards."""
    from samplenerfro_amd import datasets

    class NoDevice(datasets.DeviceBatcher):
        def __init__(self, **kw):                      # the draw needs only the geometry
            self.n_examples, self.h, self.w = 3, 20, 30
            self.batch_size, self.batching, self.patch_size = kw["batch_size"], kw["batching"], kw["patch_size"]
            self.precrop_iters, self.precrop_frac, self.train_it, self.rng = kw["precrop_iters"], 0.5, 0, kw["rng"]

    for batching in ("single_image", "all_images"):
        kw = dict(batch_size=64, batching=batching, patch_size=4, precrop_iters=1)
        b = NoDevice(rng=np.random.RandomState(11), **kw)
        ref = np.random.RandomState(11)
        hw = 20 * 30
        dummy_rays = [np.zeros((3, hw, 3), F32)] * 3 if batching == "single_image" else [np.zeros((3 * hw, 3), F32)] * 3
        dummy_img = np.zeros((3, hw, 3), F32) if batching == "single_image" else np.zeros((3 * hw, 3), F32)
        for it in range(3):
            d = b.draw()
            _reference_next_train(ref, dummy_img, dummy_rays, h=20, w=30, train_it=it, precrop_frac=0.5, **kw)
            assert d["ray_indices"].dtype == np.int64 and d["ray_indices"].shape == (64,) and d["env_indices"].shape == (4, 4)
            assert 0 <= d["ray_indices"].min() and d["ray_indices"].max() < 3 * hw
            assert ref.randint(1 << 30) == b.rng.randint(1 << 30), (batching, it)           # both generators are in the same state


def _golden():
    d = np.load(os.path.join(HERE, "golden", "reference_numpy.npz"))
    x = {k[3:]: d[k] for k in d.files if k.startswith("in_")}
    y = {k[4:]: d[k] for k in d.files if k.startswith("out_")}
    return x, y


def _golden_batcher(x, batching, device, cls=None):
    from samplenerfro_amd import datasets
    kw = dict(batch_size=int(x["bat_batch_size"]), batching=batching, patch_size=int(x["bat_patch_size"]) if batching == "single_image" else 0,
              precrop_iters=int(x["bat_precrop_iters"]), precrop_frac=float(x["bat_precrop_frac"]), rng=np.random.RandomState(int(x["bat_seed"])), prefetch=0,
              focal=float(x["focal"]), pixel_center=True)
    return (cls or datasets.DeviceBatcher)(x["bat_images"], x["c2w"], device=device, **kw)


def test_draws_reproduce_the_references_own_next_train_without_a_device():
    """The index half against batches the REFERENCE computed (its _next_train, from its source): DeviceBatcher's draws, used to index the
    oracle's ray arrays (bit-equal to the reference's _generate_rays, tests/test_reference_numpy_pin.py) and the images, give its pixels and
    rays bit for bit — both batching modes, two pre-crop batches, two full-image ones, the env-map patch."""
    from samplenerfro_amd import datasets
    x, y = _golden()
    H, W, n = int(x["H"]), int(x["W"]), x["c2w"].shape[0]
    per_view = [R.generate_rays(x["c2w"][i], H, W, pixel_center=True, focal=float(x["focal"])) for i in range(n)]
    flat = [np.concatenate([pv[k].reshape(-1, 3) for pv in per_view]) for k in range(3)]      # origins, directions, viewdirs over (view, row, column)
    img = x["bat_images"].reshape(-1, 3)

    class NoDevice(datasets.DeviceBatcher):
        def __init__(self, images, c2w, device=None, **kw):
            self.n_examples, self.h, self.w = n, H, W
            self.batch_size, self.batching, self.patch_size = kw["batch_size"], kw["batching"], kw["patch_size"]
            self.precrop_iters, self.precrop_frac, self.train_it, self.rng = kw["precrop_iters"], kw["precrop_frac"], 0, kw["rng"]

    for batching in ("single_image", "all_images"):
        bat = _golden_batcher(x, batching, None, NoDevice)
        for it in range(int(x["bat_steps"])):
            d = bat.draw()
            pre = f"bat_{batching}_{it}_"
            assert np.array_equal(img[d["ray_indices"]], y[pre + "pixels"]), (batching, it)
            for k, f in enumerate(("origins", "directions", "viewdirs")):
                assert np.array_equal(flat[k][d["ray_indices"]], y[pre + f]), (batching, it, f)
                if batching == "single_image":
                    assert np.array_equal(flat[k][d["env_indices"]], y[pre + "env_" + f]), (batching, it, f)


@pytest.mark.gpu
def test_device_batches_equal_the_references_own_next_train():
    """The whole batcher on the device against the batches the reference's own _next_train computed (tests/golden/reference_numpy.npz)."""
    x, y = _golden()
    for batching in ("single_image", "all_images"):
        bat = _golden_batcher(x, batching, "cuda:0")
        for it in range(int(x["bat_steps"])):
            b = next(bat)
            pre = f"bat_{batching}_{it}_"
            assert np.array_equal(b["pixels"].cpu().numpy(), y[pre + "pixels"]), (batching, it)
            for f in ("origins", "directions", "viewdirs"):
                assert np.array_equal(getattr(b["rays"], f).cpu().numpy(), y[pre + f]), (batching, it, f)
                if batching == "single_image":
                    assert np.array_equal(getattr(b["env_rays"], f).cpu().numpy(), y[pre + "env_" + f]), (batching, it, f)
        assert bat.out_of_range_indices() == 0
