"""include/rnerf.h: "thread-compatible (one model per GPU per thread), no global state".  Two host threads, each with its own model,
workspace and HIP stream on the SAME device, drive the whole-path entry points concurrently (render passes and optimisation steps); every
result must be the bits of the same work done by one thread alone.  What could break it: lazily initialised tables / kernel attributes
inside the library, the pooled ordering events of csrc/pipeline.hip, a shared scratch buffer."""
import threading

import numpy as np
import pytest

torch = pytest.importorskip("torch")

pytestmark = pytest.mark.gpu
F32 = np.float32


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to("cuda:0")


def _work(seed, rounds, out, barrier=None, use_stream=True):
    from samplenerfro_amd import models, synthetic as syn, utils
    from samplenerfro_amd.train import TrainState, train_step
    torch.cuda.set_device(0)
    G, B = 24, 160 + 32 * seed
    grid = syn.scale_ior(syn.sphere_grid(G, 1.5, 0.5 + 0.05 * seed), 0.5).astype(F32)
    flags = utils.default_flags(num_coarse_samples=8, num_fine_samples=12, num_path_samples=4, white_bkgd=False, bg_weight=0.025,
                                bg_smooth_weight=0.0, use_online_sparsity=False, lr_delay_steps=0, max_steps=1000, randomized=True)
    stream = torch.cuda.Stream() if use_stream else torch.cuda.current_stream()
    with torch.cuda.stream(stream):
        model, variables = models.construct_nerf(np.array([0, 7 + seed], np.uint32), None, flags, [G] * 3, [-1.5] * 3, [1.5] * 3, T(grid))
        pf = syn.init_params_flat(seed, fine=True, bias_scale=0.1)
        for k in ("coarse_mlp", "fine_mlp", "bkgd_mlp"):
            variables["flat"][k].copy_(T(pf[k]))
        o, d = syn.sphere_rays(B, seed=seed)
        rays = utils.Rays(T(o), None, T(d), None)
        pix = T(np.random.default_rng(seed).uniform(0, 1, (B, 3)).astype(F32))
        state = TrainState.create(model, variables, flags)
        state.lr_fn = lambda c: 1e-3
        batch = {"rays": rays, "pixels": pix, "annealed_alpha": 0.5}
        if barrier is not None:
            barrier.wait()                          # both threads enter the library's first-call paths together
        rng = np.array([seed, 2], np.uint32)
        res = []
        for i in range(rounds):
            ret, _ = model.apply(state.variables, rng, rng, rays, True)          # one rnerf_forward
            res.append(ret[-1][0].clone())
            state, stats, rng = train_step(model, rng, state, batch, flags)        # rnerf_train_forward_backward + rnerf_adam_update
            res.append(stats.loss.clone())
        res.append(state.theta.detach().clone())
        stream.synchronize()
    out[seed] = [r.cpu() for r in res]


@pytest.mark.timeout(600)
def test_two_host_threads_two_streams_one_device():
    rounds = 6
    together, alone = {}, {}
    bar = threading.Barrier(2)
    errs = []

    def guarded(seed):
        try:
            _work(seed, rounds, together, bar)
        except Exception as e:      # noqa: BLE001 — surfaced below
            errs.append(e)
            bar.abort()

    th = [threading.Thread(target=guarded, args=(s,)) for s in (1, 2)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=500)
    assert not errs, errs
    assert all(not t.is_alive() for t in th)
    for s in (1, 2):
        _work(s, rounds, alone, None)
    for s in (1, 2):
        assert len(together[s]) == len(alone[s]) == 2 * rounds + 1
        for a, b in zip(together[s], alone[s]):
            assert torch.equal(a, b), f"thread {s}: concurrent result differs from the single-threaded one"
