#!/usr/bin/env python3
"""bench.py — throughput of the SampleNeRFRO hot path on MI355X (contract: see the task brief / DESIGN.md §Measurement).

One "step" = one pass of the hot path (march -> bkgd MLP -> PE+NerfMLP -> composite [-> resample -> PE+NerfMLP ->
composite]) over one synthetic batch of rays that is already resident in HBM.  Default workload = BASELINE.json
configs[1] ("ship_straight": 4096 rays x 128 samples, G=512 grid == 1, flat N_f=0).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--fine F] [--precision P] [--rays B]

N > 1 is launched by the driver through torch.distributed.run (one process per GPU, RCCL).  Rays shard
embarrassingly (weak scaling, no data-path collective); the only collectives are the timing barrier / max.
Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MLP_FLOP_PER_ROW = 2 * 593408          # NerfMLP MACs*2 per sample row
BKGD_FLOP_PER_RAY = 2 * 56448
PEAK_MFMA_16BIT = 2.5e15               # dense bf16/f16 MFMA peak (MI355X_MICROARCH.md)
PEAK_HBM = 8.0e12
TABLE_LAYOUT = "reference"             # --table-layout bricks: the 2x2x2-brick order of the IoR table (same values and indices)
PRIME_STEPS = 6                         # untimed steps every Stepper runs at construction, before the contract's W warm-up steps
CPU_WARMUP, CPU_TIMED = 3, 5           # cpu_baseline: 3 warm-up + 5 timed passes, median
PRECISION_NOTES = {
    "f16x3": "fp32 operands split into hi + lo f16 parts, 3 MFMAs per product, fp32 accumulate (fp32-grade: |dRGB| ~1e-6 vs the oracle)",
    "bf16x3": "fp32 operands split into hi + lo bf16 parts, 3 MFMAs per product, fp32 accumulate",
    "f16x2": "exact hi + lo f16 weights x activations rounded to f16, 2 MFMAs per product (opt-in inference mode: |dRGB| 3e-5..7e-5, "
             "inside the 1e-4 contract without the margin f16x3 keeps)",
    "f16f8": "f16 main term + the two cross terms of the hi/lo split on the fp8 (e4m3) MFMA: 3 MFMAs per product like f16x3, less power "
             "(the default arithmetic of the render pass: |dRGB| ~2e-6 vs the oracle; a weight of magnitude >= 3.99 makes the launch fall back to "
             "f16x3 on the device)",
}


def build_scene(cfg, device, precision, fine, stage="radiance", eval_precision=None):
    import torch
    from samplenerfro_amd import models, ops, synthetic as syn
    G, ext = cfg["G"], cfg["extent"]
    ndim, nmin, nmax = [G] * 3, [-ext] * 3, [ext] * 3
    if cfg["radius"] > 0:
        a = torch.linspace(-ext, ext, G, dtype=torch.float64, device=device)
        r = torch.sqrt(a[:, None, None] ** 2 + a[None, :, None] ** 2 + a[None, None, :] ** 2)
        h = 2.0 * ext / (G - 1)
        raw = 1.0 + 0.33 * torch.clamp((cfg["radius"] - r) / h + 0.5, 0.0, 1.0)
        grid = ((raw - 1.0) * cfg["ri"] / 0.33 + 1.0).float()
        del raw, r
        if cfg["ksize"] > 0:
            grid = ops.grid_prefilter(grid, cfg["ksize"], cfg["ksigma"])
    else:
        grid = torch.ones((G, G, G), dtype=torch.float32, device=device)
    model = models.NerfModel(ndim=ndim, nmin=nmin, nmax=nmax, grid=grid, near=cfg["near"], far=cfg["far"],
                             num_coarse_samples=cfg["S"], num_fine_samples=fine, num_path_samples=cfg["P"],
                             precision=precision, eval_precision=eval_precision, device=device, stage=stage, table_layout=TABLE_LAYOUT)
    del grid
    pf = syn.init_params_flat(0, fine=fine > 0)
    flat = {k: torch.from_numpy(v).to(device) for k, v in pf.items()}
    if stage.startswith("all"):        # so3_mlp: glorot + N(0, 1e-5) output layer (rnerf/ior_utils.py:148-152)
        flat["so3_mlp"] = models.init_mlp_flat(torch.Generator().manual_seed(1), models.SO3_MLP_SHAPES, out_std=1e-5).to(device)
    variables = models.make_variables(flat)
    return model, variables, pf


def cpu_baseline(cfg, pf, fine, sample_rays, seed, train=False):
    """The oracle (numpy fp32 restatement, multi-threaded BLAS for the matmuls) on a bounded sample of the workload."""
    from oracle import ref_np as R
    from samplenerfro_amd import synthetic as syn
    G, ext = cfg["G"], cfg["extent"]
    ndim, nmin, nmax = [G] * 3, [-ext] * 3, [ext] * 3
    if cfg["radius"] > 0:
        grid = syn.scale_ior(syn.sphere_grid(G, ext, cfg["radius"]), cfg["ri"]).astype(np.float32)
        # the prefilter is a one-off, not part of a step: the separable fp32 filter is enough for a timing input
        from scipy.ndimage import gaussian_filter1d  # noqa: F401  (only to smooth the timing input)
        for ax in range(3):
            grid = gaussian_filter1d(grid, cfg["ksigma"], axis=ax, mode="nearest", truncate=(cfg["ksize"] // 2) / cfg["ksigma"])
    else:
        grid = np.ones((G, G, G), np.float32)
    table = R.build_table(grid, ndim, nmin, nmax)
    del grid
    o, d = syn.sphere_rays(sample_rays, seed=seed)
    mc = R.ModelConfig(ndim, nmin, nmax, near=cfg["near"], far=cfg["far"], num_coarse_samples=cfg["S"],
                       num_fine_samples=fine, num_path_samples=cfg["P"])
    jitter = np.arange(0, mc.num_samples, cfg["P"]) + (cfg["P"] // 2)
    params = syn.params_tree(pf)
    times = []
    for it in range(CPU_WARMUP + CPU_TIMED):   # warm-up passes (thread pools, page faults, BLAS autotuning), then the median of the timed ones
        if train:
            _, dt = cpu_train_step(R, mc, pf, params, table, o, d, jitter, cfg, fine, seed)
        else:
            t0 = time.perf_counter()
            R.nerf_forward(mc, params, table, o, d, jitter)
            dt = time.perf_counter() - t0
        if it >= CPU_WARMUP:
            times.append(dt)
    dt = float(np.median(times))
    return sample_rays / dt, dt


def parity_vs_oracle(model, pf, cfg, fine, device, n_rays=4096, others=None):
    """BASELINE metric 'PSNR vs ref' / max-abs error: the GPU path against the fp32 oracle on a small sample of the SAME workload
    (same table, initial weights, rays, jitter).  The oracle is the checker here, never the thing measured."""
    import torch
    from oracle import ref_np as R
    from samplenerfro_amd import models, synthetic as syn
    from samplenerfro_amd.utils import Rays
    G, ext = cfg["G"], cfg["extent"]
    ndim, nmin, nmax = [G] * 3, [-ext] * 3, [ext] * 3
    table = model.table_reference().cpu().numpy().reshape(-1, 4)
    o, d = syn.sphere_rays(n_rays, seed=syn.SEED + 7)
    mc = R.ModelConfig(ndim, nmin, nmax, near=cfg["near"], far=cfg["far"], num_coarse_samples=cfg["S"], num_fine_samples=fine,
                       num_path_samples=cfg["P"])
    jitter = np.arange(0, mc.num_samples, cfg["P"]) + (cfg["P"] // 2)
    oret, _ = R.nerf_forward(mc, syn.params_tree(pf), table, o, d, jitter)
    del table
    fresh = models.make_variables({k: torch.from_numpy(v).to(device) for k, v in pf.items()})
    rays = Rays(torch.from_numpy(o).to(device), None, torch.from_numpy(d).to(device), None)
    key = np.array([0, 1], np.uint32)
    def against(mdl):
        ret, _ = mdl.apply(fresh, key, key, rays, False, jitter=jitter)
        rgb = ret[-1][0].cpu().numpy().astype(np.float64); dist = ret[-1][1].cpu().numpy().astype(np.float64)
        mse = float(((rgb - oret[-1][0]) ** 2).mean())
        return {"rays": n_rays, "max_abs_rgb": float(np.abs(rgb - oret[-1][0]).max()), "max_abs_dist": float(np.abs(dist - oret[-1][1]).max()),
                "psnr_db_vs_oracle": (-10.0 * np.log10(mse)) if mse > 0 else float("inf")}

    out = against(model)
    for name, mdl in (others or {}).items():          # the labelled single-pass legs: the same oracle pass, the same rays
        out[name] = against(mdl)
    return out


def cpu_train_step(R, mc, pf, params, table, o, d, jitter, cfg, fine, seed):
    """One optimisation step on the host cores: the oracle marches / resamples (no gradient there), torch CPU fp32 autograd does
    the differentiable part of train.py's loss_fn (oracle/torch_ref.py) and Adam."""
    import torch
    from oracle import torch_ref as TR
    B = o.shape[0]
    rng = np.random.default_rng(seed)
    pix = torch.from_numpy(rng.uniform(0, 1, (B, 3)).astype(np.float32))
    ev = rng.standard_normal((128 * 128, 3)).astype(np.float32)
    ev /= np.linalg.norm(ev, axis=-1, keepdims=True)
    names = ["coarse_mlp", "bkgd_mlp"] + (["fine_mlp"] if fine > 0 else [])
    th = {k: torch.tensor(pf[k], dtype=torch.float32, requires_grad=True) for k in names}
    opt = torch.optim.Adam(list(th.values()), lr=5e-4)
    f32 = lambda a: torch.from_numpy(np.ascontiguousarray(a, np.float32))
    t0 = time.perf_counter()
    rp, rd, rdist, _, idx_grad = R.path_sampler(o, d, table, mc.ndim, mc.nmin, mc.nmax, mc.near, mc.far, mc.num_samples, np.float32)   # [B,N,*]
    jitter = np.asarray(jitter, np.int64)

    def level(name, pos, dirs, t, bk):
        S = pos.shape[1]
        raw = TR.nerf_mlp(th[name], f32(R.pos_enc(pos.reshape(-1, 3), 0, 10)), f32(R.pos_enc(dirs.reshape(-1, 3), 0, 4))).reshape(B, S, 4)
        rgb, sigma = TR.activations(raw)
        comp, acc, w, trans, tb = TR.volumetric_rendering(rgb, sigma, f32(t), f32(dirs), bk)
        return comp, trans, tb, w

    pos, dirs, t = rp[:, jitter], rd[:, jitter], rdist[:, jitter]
    bk = TR.bkgd_mlp(th["bkgd_mlp"], f32(R.pos_enc(dirs[:, -1], 0, 4)))
    comp_c, trans_c, tb_c, w_c = level("coarse_mlp", pos, dirs, t, bk)
    levels = [(comp_c, trans_c, tb_c)]
    if fine > 0:
        mid = np.float32(0.5) * (t[..., 1:] + t[..., :-1])
        u = R.linspace_u(fine, B, np.float32)
        z_f, pos_f, dir_f, _, _ = R.sample_pdf(u, mid, w_c.detach().numpy()[..., 1:-1], rp, rd, rdist, idx_grad, jitter)
        comp_f, trans_f, tb_f, _ = level("fine_mlp", pos_f, dir_f, z_f, bk)
        levels.append((comp_f, trans_f, tb_f))
    loss, _ = TR.radiance_loss(levels, pix, 0.025, 0.5)
    env = TR.bkgd_mlp(th["bkgd_mlp"], f32(R.pos_enc(ev, 0, 4))).reshape(128, 128, 3)
    loss = loss + (0.5 * ((env[1:, :] - env[:-1, :]) ** 2).reshape(-1) + 0.5 * ((env[:, 1:] - env[:, :-1]) ** 2).reshape(-1)).mean()
    opt.zero_grad()
    loss.backward()
    opt.step()
    dt = time.perf_counter() - t0
    return B / dt, dt


MFMA_PASSES = {"f16x3": 3, "bf16x3": 3, "f16f8": 3, "f16x2": 2, "f16": 1, "bf16": 1}     # MFMAs issued per algorithmic product


def sustained_mfma_tflops(device, cus):
    """The matrix pipe's ceiling on THIS chip in THIS run: back-to-back v_mfma_f32_32x32x16_f16 on every CU, no memory traffic
    (csrc/ubench/mfma_rate.hip -> librnerf_ubench.so, measurement infrastructure, not the product library), timed with events."""
    import ctypes as C
    import torch
    from samplenerfro_amd import build as B_
    if not os.path.exists(B_.LIB_UBENCH):
        return None
    ub = C.CDLL(B_.LIB_UBENCH)
    ub.rnerf_ubench_mfma.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_void_p]
    out = torch.empty(cus * 256, dtype=torch.float32, device=device)
    flop = C.c_double(0.0)
    st = torch.cuda.current_stream().cuda_stream
    res = {}
    for kind, name in ((0, "f16"), (1, "bf16")):
        assert ub.rnerf_ubench_mfma(kind, cus, 200, out.data_ptr(), C.byref(flop), st) == 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):                                   # ~3 x 12 ms: long enough for the clocks to settle under the power limit
            ub.rnerf_ubench_mfma(kind, cus, 6000, out.data_ptr(), C.byref(flop), st)
        e1.record()
        torch.cuda.synchronize()
        res[name] = 3.0 * flop.value / (e0.elapsed_time(e1) * 1e-3) / 1e12
    return res


def with_pass_ceiling(roof, passes, sustained):
    """roofline object + what separates 'cost of the fp32-grade arithmetic' from 'engine efficiency': MFMAs issued per algorithmic
    product, the measured all-CU MFMA rate of this run, and the algorithmic rate as a fraction of (that rate / passes)."""
    if roof is None or roof.get("bound") != "mfma":
        return roof
    roof["passes"] = passes
    roof["sustained_mfma_tflops"] = sustained
    roof["frac_of_pass_ceiling"] = (roof["achieved"] * passes / sustained) if sustained else None
    return roof


def dtype_label(precision, backward, train):
    """The arithmetic the path computes in, not a precision claim: operand format of the MFMAs / accumulator."""
    fwd = {"f16x3": "f16x3", "bf16x3": "bf16x3", "f16x2": "f16x2", "f16f8": "f16+fp8x2", "f16": "f16", "bf16": "bf16", "f32": "f32"}.get(precision, precision)
    if not train:
        return fwd + "/fp32-acc"
    bwd = backward
    return (fwd if bwd == fwd else fwd + " fwd, " + bwd + " bwd") + "/fp32-acc"


def relaunch_for_gpus(args):
    """`--gpus N` is the contract, not a label: one process per GPU.  Started bare (`python bench.py --gpus N`, no launcher in the
    environment) with N > 1, this process becomes the launcher: it starts `python -m torch.distributed.run --nproc-per-node N bench.py
    <same arguments>` as a CHILD (never an exec, and before anything here touches the GPU), relays the child's output — rank 0's one
    JSON line — and exits with its return code.  Started under a launcher (WORLD_SIZE set), the world size must equal --gpus."""
    env_world = os.environ.get("WORLD_SIZE")
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if env_world is None:
        if args.gpus == 1:
            return
        import socket
        import subprocess
        s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        raise SystemExit(subprocess.run(cmd, env=env).returncode)
    if int(env_world) != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={env_world} ranks; "
                         "they must agree (the line's n_gpus is the number of ranks that ran)")


class Stepper:
    """One training (or forward) step of a workload, repeated.  Training goes through train_step (two whole-path C calls per step, the
    next step's march on a side stream beside the wgrad); --graph replays the step's launch graph instead
    (samplenerfro_amd.graph.GraphTrainStep: one hipGraphLaunch per step, the march on a side branch)."""

    def __init__(self, args, cfg, model, variables, rays, key, B, world, rank, fine, device, backward, mode, stage, pipeline, graph):
        import torch
        from samplenerfro_amd import utils as U
        self.args, self.model, self.variables, self.rays, self.key, self.mode = args, model, variables, rays, key, mode
        self.train = mode == "train"
        self.stage = stage
        self.pipeline, self.graph, self.h, self.rng, self.g = pipeline, graph and self.train and stage == "radiance", None, key, None
        if not self.train:
            return
        # the shipped configs' loss terms (configs/*.yaml): bg_weight 0.025, bg_smooth_weight 1.0 on a 128x128 env-map patch,
        # randomized stratified resampling, Adam with the reference schedule; pixels are synthetic
        from samplenerfro_amd import synthetic as syn
        from samplenerfro_amd.train import TrainState
        self.flags = U.default_flags(num_coarse_samples=cfg["S"], num_fine_samples=fine, num_path_samples=cfg["P"], white_bkgd=False,
                                     bg_weight=0.025, bg_smooth_weight=1.0, bg_patch_size=128, use_online_sparsity=False, randomized=True,
                                     near=cfg["near"], far=cfg["far"], batch_size=B * world, backward_precision=backward, stage=stage)
        self.tstate = TrainState.create(model, variables, self.flags)
        gen = np.random.default_rng(syn.SEED + 1000 + rank)
        ev_d = gen.standard_normal((self.flags.bg_patch_size, self.flags.bg_patch_size, 3)).astype(np.float32)
        ev_d /= np.linalg.norm(ev_d, axis=-1, keepdims=True)
        self.batch = {"rays": rays, "pixels": torch.from_numpy(gen.uniform(0, 1, (B, 3)).astype(np.float32)).to(device), "annealed_alpha": 0.5,
                      "env_rays": U.Rays(None, None, torch.from_numpy(ev_d).to(device), None)}
        if self.graph:
            from samplenerfro_amd.graph import GraphTrainStep
            self.g = GraphTrainStep(model, self.tstate, self.flags, B, key, env_rays=self.batch["env_rays"], annealed_alpha=0.5, prefetch=pipeline)
            self.g.load(self.batch)
            self.g.load_next(self.batch)              # the synthetic batch is resident in both static slots: steps copy nothing
            for _ in range(3):                       # eager warm-up step + the capture of both slots' graphs, outside every timed region
                self.step()
        for _ in range(PRIME_STEPS):                 # settle the allocator, lazy kernel attributes and the clocks before any timing
            self.step()

    def step(self, last=False):
        if self.g is not None:
            # the rays of the step after are already in the other static slot (a loader would write them there: GraphTrainStep.next_buffers);
            # their march is a side branch of this step's graph
            return self.g.step().loss
        if self.train:
            from samplenerfro_amd.train import train_step
            nxt = self.rays if (self.pipeline and not last) else None
            if self.stage == "all":
                # a training batch is new every step, this bench's rays are not: every step gets its rays as NEW tensor objects, so the per-batch
                # shell order (ops._shell_order: a coarse pre-march + a sort, cached per tensor) is computed once per step like in a real run —
                # by the step BEFORE, on the side stream (train_step's next_rays), which is what a loader that knows the next batch allows
                from samplenerfro_amd.utils import Rays as _Rays
                if getattr(self, "_next_batch", None) is None:
                    self._next_batch = dict(self.batch, rays=_Rays(self.rays.origins.clone(), None, self.rays.viewdirs.clone(), None))
                self.batch = self._next_batch
                self._next_batch = dict(self.batch, rays=_Rays(self.rays.origins.clone(), None, self.rays.viewdirs.clone(), None))
                nxt = self._next_batch["rays"]
            # the march of step k+1 is issued on the side stream behind the backward of step k
            _, stats, self.rng = train_step(self.model, self.rng, self.tstate, self.batch, self.flags, path=self.h, next_rays=nxt)
            self.h = self.tstate.next_path
            return stats.loss
        m, a = self.model, self.args
        h = self.h if self.pipeline else None
        if self.pipeline and h is None:
            h = m.prefetch_path(self.rays, sync_inputs=False, reserve_cus=a.reserve_cus)
        self.h = m.prefetch_path(self.rays, sync_inputs=False, reserve_cus=a.reserve_cus) if (self.pipeline and not last) else None
        ret, _ = m.apply(self.variables, self.key, self.key, self.rays, False, path=h)
        return ret[-1][0]

    def close(self):
        if self.g is not None:
            self.g.close()
        self.g = self.tstate = None
        self.model._ws.clear()


def timed_steps(stepper, warmup, steps, barrier, D, device):
    import torch
    # every step — warm-up and timed — also issues the march of the step after (the pipeline's steady state): the timed region contains
    # exactly K marches and consumes K paths; the march the last timed step issues is waited for by the closing barrier
    for i in range(warmup):
        stepper.step()
    barrier()
    t0 = time.perf_counter()
    out = None
    trace = [] if getattr(stepper.args, "trace_steps", False) else None
    if trace is not None:
        import gc
        def _gc_cb(phase, info, _st={}):
            if phase == "start":
                _st["t"] = time.perf_counter()
            else:
                print("[trace-steps] gc generation %d: %.2f ms, collected %s (step index %d)" % (info["generation"], 1e3 * (time.perf_counter() - _st["t"]), info.get("collected"), len(trace)), file=sys.stderr)
        gc.callbacks.append(_gc_cb)
    for i in range(steps):
        if trace is not None:
            t1 = time.perf_counter()
        out = stepper.step()
        if trace is not None:
            trace.append(1e3 * (time.perf_counter() - t1))
    barrier()
    dt = D.max_over_ranks(time.perf_counter() - t0, device)
    if trace is not None:
        gc.callbacks.remove(_gc_cb)
        print("[trace-steps] host ms per step() call: " + " ".join("%.2f" % x for x in trace) + "  | window %.3f ms per step" % (1e3 * dt / steps), file=sys.stderr)
    assert bool(torch.isfinite(out).all())
    return dt


def pmc_lookup(workload, fine, B, mode, backward, rnerf_cus, precision="f16x3"):
    """HBM bytes and SQ / GRBM counters per launch from the committed rocprofv3 --pmc passes of THIS command (tools/r04/pmc_all.sh; PMC
    counters cannot be read from inside the process).  The JSON is stamped with the sha of the kernel sources and of bench.py it was taken
    with: a stale stamp (or no profile of this workload) leaves every counter-derived field null."""
    import hashlib
    old = {"f16x3": "f32", "f16": "tf32"}.get(backward, backward)       # (the mode names of rounds 2-4, in the file names of profiles/r04)
    rel = None
    for rnd, bw in (("r06", backward), ("r05", backward), ("r04", old)):                   # the newest committed pass whose stamp still matches wins
        psuf = "" if precision in ("f16x3", "f16f8") else "_p" + precision        # (the default arithmetics carry no suffix: forward f16f8, train f16x3)
        cand = os.path.join("profiles", rnd, "pmc_%s_f%d_%s%s.json" % (workload, fine, mode if mode == "forward" else "train_" + bw, psuf))
        if os.path.exists(os.path.join(ROOT, cand)):
            rel = cand
            break
    if B != 4096 or rel is None:
        return {}, {}, None
    path = os.path.join(ROOT, rel)
    try:
        pj = json.load(open(path))
        sha = lambda q: hashlib.sha256(open(q, "rb").read()).hexdigest()[:16]
        now = {f: sha(os.path.join(ROOT, "samplenerfro_amd", "csrc", f)) for f in pj.get("csrc_sha16", {})}
        fresh = bool(now) and now == pj.get("csrc_sha16")
        meta = {"file": rel, "head": pj.get("head"), "bench_py_sha16": pj.get("bench_py_sha16"), "kernels_unchanged_since": fresh,
                "source": "every `traffic` / `counters` field of this line is read from this COMMITTED rocprofv3 --pmc pass of the same command (counters cannot "
                          "be read from inside the process); nothing counter-derived is measured in this run, and a stale source stamp nulls them"}
        traffic, sq = {}, {}
        if fresh:
            for k, v in pj["counters"].items():
                name = k.split("::")[-1]
                if "FETCH_SIZE" in v and "WRITE_SIZE" in v:
                    # KiB -> bytes; wide (16 B/lane) streaming reads are reported at half their size on gfx950 (MI355X_MICROARCH.md, HBM):
                    # doubled for the MLP kernels (weight DMA / saved-operand streams); the march's gathers stay as reported
                    ff = 2.0 if "nerfmlp" in k else 1.0
                    traffic[name] = 1024.0 * (ff * v["FETCH_SIZE"]["mean"] + v["WRITE_SIZE"]["mean"])
                if "SQ_VALU_MFMA_BUSY_CYCLES" in v and "GRBM_GUI_ACTIVE" in v:
                    gui = v["GRBM_GUI_ACTIVE"]["mean"] / float(pj.get("xcd_instances", 8))     # reported summed over the XCDs' GRBM instances
                    sq[name] = {"mfma_busy_frac": v["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / (4.0 * rnerf_cus * gui),
                                "effective_clock_ghz": gui / v["avg_ns"]["mean"] if "avg_ns" in v else None,
                                "wave_wait_frac": (v["SQ_WAIT_ANY"]["mean"] / v["SQ_WAVE_CYCLES"]["mean"]) if "SQ_WAIT_ANY" in v and "SQ_WAVE_CYCLES" in v else None}
        return traffic, sq, meta
    except Exception:
        return {}, {}, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--trace-steps", action="store_true", help="print the host time of every step() call of every timed window to stderr (diagnostics)")
    ap.add_argument("--workload", default="ship_straight")
    ap.add_argument("--fine", type=int, default=None, help="num_fine_samples (default: the workload's flat variant, 0)")
    ap.add_argument("--precision", default="f16x3", help="arithmetic of training and of every tapped path")
    ap.add_argument("--eval-precision", default=None,
                    help="arithmetic of the pure render pass (forward mode, the frame): default = construct_nerf's default = --precision (round 6; f16f8 "
                         "— f16 main term + fp8 cross terms — is opt-in: --eval-precision f16f8)")
    ap.add_argument("--rays", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pipeline", dest="pipeline", action="store_true", default=None,
                    help="march of step k+1 beside step k (train: a side branch of the step's graph / the side stream, default ON; "
                         "forward: beside the MLP on reserved CUs, default OFF)")
    ap.add_argument("--no-pipeline", dest="pipeline", action="store_false", help="every step runs its stages strictly in sequence")
    ap.add_argument("--graph", dest="graph", action="store_true", default=False,
                    help="train: replay the step's launch graph (one hipGraphLaunch per step: samplenerfro_amd.graph.GraphTrainStep) instead of "
                         "issuing its ~30 kernels through the two whole-path C calls.  Measured 1-2 %% SLOWER on this ROCm (a graph node costs more "
                         "than a stream launch and the step is not launch-bound), so it is off by default; the line reports it under `graph_replay`")
    ap.add_argument("--no-graph", dest="graph", action="store_false")
    ap.add_argument("--no-extra", dest="extra", action="store_false",
                    help="skip the short timings of the other backward modes and of the other workload variants (train mode)")
    ap.add_argument("--no-frame", dest="frame", action="store_false",
                    help="skip the 800x800 full-frame render (ms/frame, the second part of BASELINE's metric; ~1 s)")
    ap.add_argument("--reserve-cus", type=int, default=32, help="CUs kept free of MLP workgroups for the overlapped march (forward mode)")
    ap.add_argument("--cpu-rays", type=int, default=None, help="rays in the CPU baseline sample (default 4096 forward / 512 train)")
    ap.add_argument("--backward", choices=["f16x3", "f16x3lo8", "f16", "bf16", "f32", "tf32"], default="f16x3",
                    help="arithmetic of the NerfMLP backward: f16x3 = hi + lo f16 parts, 3 MFMAs per product (fp32-grade, the reference differentiates "
                         "in fp32; default), f16 = f16 parts (11-bit significand), bf16 = 8-bit significand (round 1's arithmetic); f32 / tf32 = the "
                         "older names of the first two")
    ap.add_argument("--stage", choices=["radiance", "all"], default="radiance",
                    help="all: so3_mlp bends the gradient inside the march and is trained through its adjoint (train.py:302-310); needs a "
                         "refractive workload (e.g. ship_refractive)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak: every rank owns --rays rays (default); strong: --rays is the GLOBAL batch, split over the ranks "
                         "(BASELINE config 4 as written: 4096 rays = 512 per GPU on 8 GPUs)")
    ap.add_argument("--mode", choices=["train", "forward"], default="train",
                    help="train: the whole optimisation step (BASELINE metric 'rays/sec (train step)'); forward: the render pass only")
    ap.add_argument("--table-layout", choices=["reference", "bricks"], default="reference",
                    help="memory order of the IoR table: the reference's flat index (default) or 2x2x2 bricks of one cache line (same bits; A/B switch)")
    ap.add_argument("--dist-backend", choices=["nccl", "gloo"], default="nccl",
                    help="nccl = RCCL over xGMI, one rank per GPU (default); gloo lets the tests drive the multi-rank branch with several ranks on one device")
    ap.add_argument("--dist-timeout", type=float, default=300.0,
                    help="seconds a rendezvous / collective may take before the rank fails (a dead rank takes the job down, it never hangs it)")
    ap.add_argument("--force-dist", action="store_true",
                    help="bring the process group up and issue every collective with ONE rank too (how RCCL itself is exercised on a one-GPU box)")
    ap.add_argument("--no-curve", dest="curve", action="store_false",
                    help="N > 1: skip the in-run scaling curve (the same step on the first 1, 2, 4, ... ranks of this launch)")
    ap.add_argument("--variant-rays", type=int, default=4096, help=argparse.SUPPRESS)   # rays per rank of the ship_* variants: several ranks that SHARE one
    ap.add_argument("--fail-rank", type=int, default=-1, help=argparse.SUPPRESS)      # fault injection for tests/test_rank_failure.py: this rank
    ap.add_argument("--fail-mode", choices=["exit", "hang"], default="exit", help=argparse.SUPPRESS)   # dies / stops responding after the warm-up
    args = ap.parse_args()
    args.backward = {"f32": "f16x3", "tf32": "f16"}.get(args.backward, args.backward)
    global TABLE_LAYOUT
    TABLE_LAYOUT = args.table_layout
    relaunch_for_gpus(args)
    if args.pipeline is None:
        args.pipeline = args.mode == "train"
    if args.cpu_rays is None:
        args.cpu_rays = 4096 if args.mode == "forward" else 512
    if args.eval_precision is None:
        # the render pass runs the training arithmetic (fp32-grade f16x3) since round 6, like construct_nerf's default: f16f8 — the default of
        # rounds 4-5 — measures 2e-4 RGB on trained-like weights (tests/test_gpu_parity.py), outside north_star's 1e-4; it stays a labelled leg
        args.eval_precision = args.precision

    import torch
    import torch.distributed as dist
    from samplenerfro_amd import synthetic as syn
    from samplenerfro_amd.utils import Rays
    from samplenerfro_amd import prng

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit(f"bench.py needs a ROCm GPU (the hot path has no CPU implementation) [rank {rank} of {world}]")
    # one process per GPU (RCCL).  --dist-backend gloo lets the tests drive this very branch with several ranks on one device
    backend = args.dist_backend
    if backend == "nccl" and local_rank >= torch.cuda.device_count():
        raise SystemExit(f"bench.py: rank {rank} has LOCAL_RANK {local_rank} but only {torch.cuda.device_count()} device(s) are visible "
                         "(RCCL runs one rank per GPU)")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    from samplenerfro_amd import distributed as D
    os.environ["LOCAL_RANK"] = str(local_rank)
    if args.force_dist:
        D.force_single_rank_group(True)
    D.init(backend, timeout_s=args.dist_timeout)   # the train step's gradient all-reduce, the timing barrier and the max over ranks
    # sub-groups of the in-run scaling curve, created up front (every rank takes part in creating each): a backend that cannot split its
    # communicator says so here, before any work — the curve is then left out (all ranks agree through one all-reduce), the line is not
    curve_groups = None
    if world > 1 and args.curve and args.mode == "train":
        import datetime
        ok, curve_groups = 1, {}
        try:
            for n in [m for m in (1, 2, 4, 8, 16, 32) if m < world]:
                curve_groups[n] = dist.new_group(ranks=list(range(n)), timeout=datetime.timedelta(seconds=args.dist_timeout))
        except Exception as e:      # noqa: BLE001
            ok = 0
            print(f"bench.py [rank {rank}]: sub-group creation failed ({type(e).__name__}: {e}); scaling_curve is left out", file=sys.stderr)
        flag = torch.tensor([ok], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            curve_groups = None

    cfg = dict(syn.CONFIGS[args.workload])
    fine = cfg["F"] if args.fine is None else args.fine
    B = args.rays or min(cfg["B"], 4096 if args.workload != "glass_frame" else 8192)
    if args.scaling == "strong":
        if B % world:
            raise SystemExit("--scaling strong: the global batch must be divisible by the number of ranks (train.py:196)")
        B //= world
    t_scene = time.perf_counter()
    model, variables, pf = build_scene(cfg, device, args.precision, fine, args.stage, args.eval_precision)
    torch.cuda.synchronize()
    t_scene = time.perf_counter() - t_scene
    if args.stage == "all":
        args.pipeline = False           # the all* march reads the so3 parameters of the current step: no cross-step prefetch
    # weak scaling: every rank marches its own B rays (different seed per rank), grid + weights replicated
    o, d = syn.sphere_rays(B, seed=syn.SEED + rank)
    rays = Rays(torch.from_numpy(o).to(device), None, torch.from_numpy(d).to(device), None)
    key = prng.split(prng.PRNGKey(syn.SEED), world)[rank]      # one key per device (train.py:338-339)
    train = args.mode == "train"

    def barrier():
        D.barrier()             # (over the ranks the collectives currently run on: all of them, or the scaling curve's sub-group)
        torch.cuda.synchronize()

    # ---- the headline: W untimed warm-up steps, then exactly K timed steps between barriers, max over ranks --------------------------
    # Every step does its whole work inside the timed region; with the pipeline on, step k also marches the rays of step k+1 (one march
    # per step either way: the first timed step consumes the march the last warm-up step issued, the last one issues one nobody reads).
    stepper = Stepper(args, cfg, model, variables, rays, key, B, world, rank, fine, device, args.backward, args.mode, args.stage, args.pipeline, args.graph)
    if args.fail_rank == rank:          # fault injection (tests/test_rank_failure.py): the others must exit non-zero within --dist-timeout
        if args.fail_mode == "exit":
            os._exit(17)
        time.sleep(20.0 * args.dist_timeout)
    # The host's garbage collector: a full (generation-2) collection walks every container object of the process — torch's and numpy's import
    # graphs included — and takes 45 ms here (measured: --trace-steps; round 6).  It comes once per ~70 000 container allocations, i.e. every
    # few hundred steps of a training loop (~1 % of its time), but when it lands in a 20-step window it adds 2 ms to every step of it: the
    # slow window every committed `stability` record of rounds 3-5 shows (7.5 among 6.2 ms) was that, and in round 6 it moved into the
    # headline window.  Set-up is over here: what exists now is moved to the permanent generation, later collections walk only what the
    # steps allocate (nothing of the timed work is skipped).  The reference's own training loop runs with the collector switched OFF
    # altogether (/root/reference/train.py:340-341: `gc.disable()  # Disable automatic garbage collection for efficiency.`, `gc.collect()`).
    import gc
    gc.collect()
    gc.freeze()
    dt = timed_steps(stepper, args.warmup, args.steps, barrier, D, device)
    graph_used = stepper.g is not None
    replicas = None
    if train and D.active():
        # data-parallel invariants after every step so far (prime + warm-up + timed): the replicas' parameters are the same BITS (every rank
        # applied the same Adam arithmetic to the same reduced gradient) although every rank drew its own rays and keys (train.py:338-339)
        iv = stepper.tstate.theta.detach().view(torch.int32).to(torch.int64)
        sig = torch.stack([iv.sum(), (iv * (torch.arange(iv.numel(), device=device) % 65521 + 1)).sum()])
        lo_s, hi_s = sig.clone(), sig.clone()
        dist.all_reduce(lo_s, op=dist.ReduceOp.MIN); dist.all_reduce(hi_s, op=dist.ReduceOp.MAX)
        keys = torch.zeros((world, 2), dtype=torch.int64, device=device)
        keys[rank, 0], keys[rank, 1] = int(key[0]), int(key[1])
        dist.all_reduce(keys, op=dist.ReduceOp.SUM)
        o_sig = torch.zeros((world,), dtype=torch.float64, device=device)
        o_sig[rank] = rays.origins.double().sum()
        dist.all_reduce(o_sig, op=dist.ReduceOp.SUM)
        replicas = {"ranks": world, "parameters_bit_identical": bool(torch.equal(lo_s, hi_s)), "after_steps": PRIME_STEPS + args.warmup + args.steps + (3 if graph_used else 0),
                    "distinct_rank_keys": len({(int(a), int(b)) for a, b in keys.tolist()}), "distinct_rank_batches": len({float(v) for v in o_sig.tolist()}),
                    "what": "signature of the flat parameter buffer's bits, MIN == MAX over the ranks; one jax.random key and one ray batch per rank"}
    # dispersion of the same measurement: five more windows of K steps each on the same stepper, outside the headline
    stability = None
    if args.extra:
        wins = [1e3 * timed_steps(stepper, 0, args.steps, barrier, D, device) / args.steps for _ in range(5)]
        stability = {"windows": 5, "steps_per_window": args.steps, "ms_per_step": [round(w, 4) for w in wins], "median_ms": float(np.median(wins)),
                     "min_ms": float(min(wins)), "max_ms": float(max(wins)), "spread_frac": float((max(wins) - min(wins)) / np.median(wins)),
                     "note": "boxes of the pool differ by +-3 %; A/B deltas are only meaningful as same-box pairs (DESIGN.md §4)"}

    curve = None
    if train and world > 1 and args.curve and curve_groups is not None:
        # the same step on the first n ranks of THIS launch (the others wait at the barrier): absolute rays/s at n = 1, 2, 4, ... world from
        # one run.  Per-rank work is fixed (weak); the gradient all-reduce runs over the n ranks only.  (Afterwards the replicas have taken
        # different numbers of steps: everything below builds fresh steppers from the seeded initial weights.)
        curve = {"n": [], "rays_per_s": [], "ms_per_step": [], "steps": 10, "rays_per_gpu": B,
                 "what": "weak: the step of this line on the first n ranks of the same launch, gradient all-reduce over those n ranks, max over them"}
        for n in [m for m in (1, 2, 4, 8, 16, 32) if m < world] + [world]:
            grp = curve_groups[n] if n < world else None
            dt_n = None
            if rank < n:
                with D.use_group(grp):
                    dt_n = timed_steps(stepper, 2, curve["steps"], barrier, D, device)
            barrier()
            if rank == 0:
                curve["n"].append(n); curve["ms_per_step"].append(1e3 * dt_n / curve["steps"]); curve["rays_per_s"].append(B * n * curve["steps"] / dt_n)
    coll = None
    if train and D.active():
        # the step's one exchange, timed with events on the launch stream: the all-reduce alone (back to back, nothing else queued), and what
        # of it a step cannot hide = (step with the collective) - (step with it skipped; the replicas' parameters diverge there, timing only)
        G = stepper.tstate.grads
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(11)]
        D.allreduce_mean_([G]); torch.cuda.synchronize()
        ev[0].record()
        for i in range(10):
            D.allreduce_mean_([G]); ev[i + 1].record()
        torch.cuda.synchronize()
        ar_us = float(np.median([1e3 * ev[i].elapsed_time(ev[i + 1]) for i in range(10)]))
        with D.skip_allreduce():
            dt_skip = timed_steps(stepper, 1, 10, barrier, D, device) / 10
        dt_with = timed_steps(stepper, 1, 10, barrier, D, device) / 10
        coll = {"allreduce_us": ar_us, "allreduce_bytes": int(G.numel() * 4), "exposed_us": 1e6 * (dt_with - dt_skip),
                "step_ms_with": 1e3 * dt_with, "step_ms_without": 1e3 * dt_skip,
                "note": "allreduce_us: the flat gradient + stats buffer, median of 10 back-to-back all-reduces (events on the launch stream); exposed_us: "
                        "step time with minus without the collective (10 steps each, same stepper)"}
    other_modes = None
    if train and args.extra:
        # the same step with the other backward arithmetics (5 steps each, after 2 warm-up steps), for the record in the same line
        other_modes = {}
        stepper.close()
        for name in ("f16x3", "f16x3lo8", "f16", "bf16"):
            if name == args.backward or (args.stage == "all" and name == "bf16") or (name in ("f16x3", "f16x3lo8") and args.precision != "f16x3"):
                continue
            s2 = Stepper(args, cfg, model, variables, rays, key, B, world, rank, fine, device, name, args.mode, args.stage, args.pipeline, args.graph)
            dt_m = timed_steps(s2, 2, 5, barrier, D, device)
            s2.close()
            other_modes[name] = {"ms_per_step": 1e3 * dt_m / 5, "rays_per_s": B * world * 5 / dt_m}
    stepper.close()
    del stepper
    torch.cuda.empty_cache()
    graph_replay = None
    if train and args.extra and args.stage == "radiance":
        # the same step the other way round (launch graph <-> stream launches), 10 steps: both forms of the north star's "one executable per step"
        s3 = Stepper(args, cfg, model, variables, rays, key, B, world, rank, fine, device, args.backward, args.mode, args.stage, args.pipeline, not args.graph)
        dt_g = timed_steps(s3, 2, 10, barrier, D, device)
        s3.close()
        del s3
        torch.cuda.empty_cache()
        graph_replay = {"launch": "one hipGraphLaunch per step" if not args.graph else "stream launches (two whole-path C calls per step)",
                        "ms_per_step": 1e3 * dt_g / 10, "rays_per_s": B * world * 10 / dt_g}

    # ---- per-kernel roofline of the dominant kernels, HIP events on the launch stream ----------------------------------------------
    from samplenerfro_amd import ops, _lib
    rnerf_cus = _lib.load().rnerf_device_cus()
    S = cfg["S"]
    N = S * cfg["P"]
    path_pd, path_dr, _, _ = ops.march(model.table, model.spec, rays.origins, rays.viewdirs, cfg["near"], cfg["far"], N)
    jit = model._jitter_dev(model.make_jitter(key))
    prec_fwd = model.precision if train else model.eval_precision       # the arithmetic the measured pass runs its NerfMLP in
    prec_fwd_name = args.precision if train else args.eval_precision
    packed = model._packed_weights(variables, "coarse_mlp", prec_fwd)
    reps = max(5, min(args.steps, 20))

    def timed(fn):
        fn(); torch.cuda.synchronize()
        e = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
        e[0].record()
        for i in range(reps):
            fn(); e[i + 1].record()
        torch.cuda.synchronize()
        return float(np.mean([e[i].elapsed_time(e[i + 1]) for i in range(reps)]))

    sustained = sustained_mfma_tflops(device, rnerf_cus)       # this chip, this run: the all-CU MFMA ceiling (f16 / bf16 32x32x16)
    sus16 = sustained["f16"] if sustained else None
    out_raw = torch.empty((S, B, 4), dtype=torch.float32, device=device)
    mlp_ms = timed(lambda: ops.nerfmlp_forward(packed, prec_fwd, path_pd, path_dr, jit, S, B, out=out_raw))
    mlp_flops = MLP_FLOP_PER_ROW * S * B
    mlp_achieved = mlp_flops / (mlp_ms * 1e-3)
    # march kernel (HBM-bound by its algorithmic gather bytes: 8 corners x 16 B per step)
    march_ms = timed(lambda: ops.march(model.table, model.spec, rays.origins, rays.viewdirs, cfg["near"], cfg["far"], N, out=(path_pd, path_dr)))
    march_bytes = B * (N * 128 + 24)
    march_achieved = march_bytes / (march_ms * 1e-3)

    # ---- training kernels, each alone with HIP events on the launch stream
    train_kernels = []
    if train:
        lib = _lib.load()
        rows = S * B
        BW = _lib.BACKWARDS[args.backward]
        pbwd = ops.nerfmlp_pack_bwd(variables["flat"]["coarse_mlp"], None, BW)
        packed = model._packed_weights(variables, "coarse_mlp")
        raw_t, save_t = ops.nerfmlp_forward_train(packed, model.precision, path_pd, path_dr, jit, S, B, BW)
        d_raw = torch.randn((S, B, 4), device=device) * 1e-3
        dy_t = torch.empty(lib.rnerf_nerfmlp_dy_bytes(rows, BW), dtype=torch.uint8, device=device)
        ws_t = torch.empty(lib.rnerf_nerfmlp_wgrad_workspace_bytes(), dtype=torch.uint8, device=device)
        g_t = torch.empty(_lib.NERFMLP_PARAMS, device=device)
        jp = jit.data_ptr()
        t_f = timed(lambda: lib.rnerf_nerfmlp_forward_train(packed.data_ptr(), model.precision, path_pd.data_ptr(), path_dr.data_ptr(), jp, S, B,
                                                            raw_t.data_ptr(), save_t.data_ptr(), BW, 0, _lib.current_stream()))
        t_d = timed(lambda: ops.nerfmlp_backward(pbwd, packed, model.precision, save_t, d_raw, rows, dy=dy_t, stages="d", backward=BW))
        t_w_alone = timed(lambda: ops.nerfmlp_backward(pbwd, packed, model.precision, save_t, d_raw, rows, grads=g_t, workspace=ws_t, dy=dy_t, stages="w", backward=BW))
        t_w = t_w_alone
        if args.pipeline and args.stage == "radiance":
            # IN the step the wgrad hosts the next batch's march as co-resident waves (rnerf_prefetch.beside_wgrad): timed here the same way —
            # the march forked onto the side stream right before the wgrad, the events around the wgrad on the launch stream — so that the
            # figure is the kernel's duration as the step (and a rocprofv3 --stats of the step) sees it, not its stand-alone best case
            side = torch.cuda.Stream()
            cur = torch.cuda.current_stream()

            def wgrad_with_march():
                side.wait_stream(cur)
                with torch.cuda.stream(side):
                    ops.march(model.table, model.spec, rays.origins, rays.viewdirs, cfg["near"], cfg["far"], N, out=(path_pd2, path_dr2))
                ops.nerfmlp_backward(pbwd, packed, model.precision, save_t, d_raw, rows, grads=g_t, workspace=ws_t, dy=dy_t, stages="w", backward=BW)
                # (no join here: the events bracket the wgrad alone; the next repetition's march waits for this wgrad, as the next step's does)

            path_pd2, path_dr2 = torch.empty_like(path_pd), torch.empty_like(path_dr)
            t_w = timed(wgrad_with_march)
            del path_pd2, path_dr2
        R_pad = (rows + 255) // 256 * 256
        # sum over the wgrad jobs of (X slots + dY slots) x R x 32 B (x 2: hi + lo).  f16 modes: 11 jobs (the Dense_5 / Dense_10 concat rows and
        # the sigma head share their operand streams with the main block: 316 slot planes); bf16 body: 14 single-segment jobs (356)
        wgrad_slots = (8 * 32 + 2 * 20 + 17 + 24 + 10 + 9) if args.backward == "bf16" else (20 + 4 * 32 + 36 + 2 * 32 + 33 + 26 + 9)
        wgrad_bytes = wgrad_slots * R_pad * 32 * {"f16x3": 2.0, "f16x3lo8": 1.5}.get(args.backward, 1.0)
        # the MLP phase is MFMA-bound and its algorithmic work is 2 x 593 408 FLOP per row (forward and wgrad) / 2 x 557 696
        # (dgrad: the encodings take no gradient).  The saved-operand / dY planes the kernels stream through HBM are an implementation
        # choice (like the path record): reported next to it as `operand_stream`, never as the algorithmic fraction.
        for name, ms, flop, byt in (("nerfmlp_fwd_kernel<train>", t_f, MLP_FLOP_PER_ROW * rows, None),
                                    ("nerfmlp_dgrad_kernel", t_d, 2 * 557696 * rows, None),
                                    ("nerfmlp_wgrad_kernel" if args.backward == "bf16" else "nerfmlp_wgrad_tr_kernel", t_w, MLP_FLOP_PER_ROW * rows, wgrad_bytes)):
            tk = {"kernel": name, "bound": "mfma", "achieved": flop / (ms * 1e-3) / 1e12, "peak": PEAK_MFMA_16BIT / 1e12,
                  "unit": "TFLOP/s", "frac": flop / (ms * 1e-3) / PEAK_MFMA_16BIT, "avg_launch_ms": ms, "algorithmic_flop_per_launch": flop}
            # MFMAs per algorithmic product: the forward in its precision; dgrad / wgrad: 3 with hi + lo planes, (W hi + W lo) x dY = 2 / 1 in f16, 2 / 1 in bf16
            npass = MFMA_PASSES[args.precision] if "fwd" in name else ({"f16x3": 3, "f16x3lo8": 3, "f16": (1 if args.precision == "f16" else 2), "bf16": 2}[args.backward] if "dgrad" in name else {"f16x3": 3, "f16x3lo8": 3, "f16": 1, "bf16": 1}[args.backward])
            with_pass_ceiling(tk, npass, sustained["bf16"] if (sustained and args.backward == "bf16" and "fwd" not in name) else sus16)
            if byt is not None:
                tk["avg_launch_ms_alone"] = t_w_alone
                tk["launched"] = ("with the next batch's march co-resident, as in the step" if t_w is not t_w_alone else "alone")
                # What bounds the wgrad is the HBM stream of its operands — X and dY of every layer, read exactly once, 4 B per value in the
                # fp32-grade mode (f16 hi + lo: the size of the fp32 tensors the reference's backward reads) — not the matrix pipe (0.50 busy):
                # reported as an HBM roofline, with the MFMA view of the same launch beside it
                mf = {k: tk[k] for k in ("achieved", "peak", "unit", "frac", "algorithmic_flop_per_launch", "passes", "sustained_mfma_tflops", "frac_of_pass_ceiling") if k in tk}
                for k in ("passes", "sustained_mfma_tflops", "frac_of_pass_ceiling"):
                    tk.pop(k, None)
                tk.update({"bound": "hbm", "achieved": byt / (ms * 1e-3) / 1e9, "peak": PEAK_HBM / 1e9, "unit": "GB/s", "frac": byt / (ms * 1e-3) / PEAK_HBM,
                           "algorithmic_bytes_per_launch": byt, "mfma": mf,
                           "bytes_note": "%d slot planes x 32 B x %d plane(s) per row: the saved activations and the gradients of every layer, read "
                                         "once (DESIGN.md §3.3); `traffic` = the PMC counters' bytes of the same launch" % (wgrad_slots, 2 if args.backward in ("f16x3", "f16x3lo8") else 1)})
            train_kernels.append(tk)
        del raw_t, save_t, dy_t, ws_t
        torch.cuda.empty_cache()

    # ---- the arithmetic `north_star` names — ONE 16-bit MFMA per product — as labelled legs of the same workload (never the headline: 11 / 8
    #      significand bits against the reference's fp32).  They separate "cost of fp32 grade" from "engine efficiency": same engines, same
    #      tiling, a third of the matrix work.  Counter fields / rocprof stats of the legs: profiles/r05 (each leg is also a bench.py command
    #      of its own: --precision f16 --backward f16, --mode forward --precision f16 | bf16).
    legs = None
    if train and args.extra and args.stage == "radiance" and args.precision == "f16x3" and args.backward == "f16x3":
        legs = {}
        from samplenerfro_amd.train import TrainState, train_step
        from samplenerfro_amd import utils as U

        def grads_of(mdl, bw):            # the gradient of ONE step on the bench batch (staged path with taps; same keys -> same jitter and draws)
            vv = models_fresh_variables(pf, device)
            fl = U.default_flags(num_coarse_samples=cfg["S"], num_fine_samples=fine, num_path_samples=cfg["P"], white_bkgd=False, bg_weight=0.025,
                                 bg_smooth_weight=0.0, use_online_sparsity=False, randomized=True, near=cfg["near"], far=cfg["far"],
                                 batch_size=B * world, backward_precision=bw, stage="radiance")
            ts = TrainState.create(mdl, vv, fl)
            gen = np.random.default_rng(syn.SEED + 1000 + rank)
            bt = {"rays": rays, "pixels": torch.from_numpy(gen.uniform(0, 1, (B, 3)).astype(np.float32)).to(device), "annealed_alpha": 0.5}
            tp = {}
            with D.skip_allreduce():      # this rank's own gradient: the comparison is per rank
                train_step(mdl, key, ts, bt, fl, taps=tp)
            g = tp["grads"][:_lib.NERFMLP_PARAMS].clone()
            del ts, tp
            return g

        g_ref = grads_of(model, "f16x3")
        # the cheaper BACKWARD arithmetics behind the default forward (their step times: `other_backward_modes`): how far their gradient of
        # this very batch is from the default's — the error of 11 / 8-bit products averages down with the number of rows (524 288 here;
        # on a few hundred rows it is ~1e-3 / ~1e-2, tests/test_gpu_backward.py), which is why the default does not rely on it
        bw_err = {}
        for bw in ("f16x3lo8", "f16", "bf16"):       # (f16x3lo8: the hi + lo mode with its lo planes stored as e4m3 bytes — 3/4 of the operand bytes)
            gb = grads_of(model, bw)
            bw_err[bw] = float((gb - g_ref).abs().max() / g_ref.abs().max())
            del gb
        legs["backward_modes_behind_the_f16x3_forward"] = {"grad_err_rel_max_vs_f16x3": bw_err, "rows": S * B}
        for pname in ("f16", "bf16"):
            mp = model_with_precision(model, pname)
            vp = models_fresh_variables(pf, device)
            P = _lib.PRECISIONS[pname]
            pk = mp._packed_weights(vp, "coarse_mlp", P)
            ms_k = timed(lambda: ops.nerfmlp_forward(pk, P, path_pd, path_dr, jit, S, B, out=out_raw))
            ach = mlp_flops / (ms_k * 1e-3) / 1e12
            sp = Stepper(args, cfg, mp, vp, rays, key, B, world, rank, fine, device, args.backward, "forward", "radiance", False, False)
            dt_p = timed_steps(sp, 2, 10, barrier, D, device)
            sp.close()
            leg = {"forward": {"ms_per_step": 1e3 * dt_p / 10, "rays_per_s": B * world * 10 / dt_p,
                               "roofline": with_pass_ceiling({"kernel": "nerfmlp_fwd_kernel", "bound": "mfma", "achieved": ach, "peak": PEAK_MFMA_16BIT / 1e12,
                                                              "unit": "TFLOP/s", "frac": ach * 1e12 / PEAK_MFMA_16BIT, "avg_launch_ms": ms_k,
                                                              "algorithmic_flop_per_launch": mlp_flops}, 1, (sustained or {}).get(pname))}}
            legs[pname] = leg
            legs[pname]["_model"] = (mp, vp)
        # the single-pass TRAIN step: f16 forward (hi plane = the operand) + f16 backward; bf16 has no training forward (its saved operands would
        # be 8-bit): its train leg is the bf16 BACKWARD behind the f16 forward
        for tag, bw in (("f16", "f16"), ("bf16", "bf16")):
            mp, vp = model_with_precision(model, "f16"), models_fresh_variables(pf, device)
            st = Stepper(args, cfg, mp, vp, rays, key, B, world, rank, fine, device, bw, "train", "radiance", args.pipeline, False)
            dt_t = timed_steps(st, 2, 10, barrier, D, device)
            st.close()
            g = grads_of(mp, bw)
            legs[tag]["train"] = {"ms_per_step": 1e3 * dt_t / 10, "rays_per_s": B * world * 10 / dt_t, "forward_precision": "f16", "backward_precision": bw,
                                  "grad_err_rel_max_vs_f16x3": float((g - g_ref).abs().max() / g_ref.abs().max()),
                                  "what": "NerfMLP gradient of one step on the bench batch against the default (f16x3 forward + f16x3 backward, itself held to "
                                          "1e-5 of max|g| vs float64 autograd by tests/test_gpu_backward.py), relative to max|g|"}
            del mp, vp, g
        # the RANGE-SAFE training arithmetic (bf16x3 forward, its bf16 hi plane saved as it is, bf16 backward): what train_step(range_retry=True)
        # re-runs a step in whose f16-based arithmetic met a row outside f16's range — never the headline, never the default
        mp, vp = model_with_precision(model, "bf16x3"), models_fresh_variables(pf, device)
        st = Stepper(args, cfg, mp, vp, rays, key, B, world, rank, fine, device, "bf16", "train", "radiance", args.pipeline, False)
        dt_t = timed_steps(st, 2, 10, barrier, D, device)
        st.close()
        g = grads_of(mp, "bf16")
        # ... and what the opt-in switch itself costs on the DEFAULT step: one host read of the non-finite count per step (no re-run happens here)
        vs = models_fresh_variables(pf, device)
        st = Stepper(args, cfg, model, vs, rays, key, B, world, rank, fine, device, args.backward, "train", "radiance", args.pipeline, False)
        st.flags.range_retry = True                       # decided in place: one host read of the non-finite count per step
        dt_s = timed_steps(st, 2, 10, barrier, D, device)
        retries = st.tstate.range_retries
        st.close()
        vl = models_fresh_variables(pf, device)
        st = Stepper(args, cfg, model, vl, rays, key, B, world, rank, fine, device, args.backward, "train", "radiance", args.pipeline, False)
        st.flags.range_retry = False                      # never re-run (the skip of a non-finite update stays): what the lagged default costs is the difference
        dt_l = timed_steps(st, 2, 10, barrier, D, device)
        retries_l = st.tstate.range_retries
        st.close()
        legs["range_safe_train"] = {"ms_per_step": 1e3 * dt_t / 10, "rays_per_s": B * world * 10 / dt_t, "forward_precision": "bf16x3", "backward_precision": "bf16",
                                    "grad_err_rel_max_vs_f16x3": float((g - g_ref).abs().max() / g_ref.abs().max()),
                                    "default_step_with_range_retry_on": {"ms_per_step": 1e3 * dt_s / 10, "rays_per_s": B * world * 10 / dt_s, "re_runs": retries},
                                    "default_step_with_range_retry_off": {"ms_per_step": 1e3 * dt_l / 10, "rays_per_s": B * world * 10 / dt_l, "re_runs": retries_l},
                                    "what": "the step a range_retry re-run costs, on the bench batch (which is inside f16's range: nothing is re-run here); the headline "
                                            "runs with flags.range_retry = 'lag' (the count of step k - 2 is read after step k is queued: no per-step host read) — beside it "
                                            "the same step with the decision in place (True: one host read per step) and with no re-run at all (False)"}
        del mp, vp, vs, vl, g
        torch.cuda.empty_cache()

        # ---- the range retry WHERE IT FIRES: a coarse network doctored so that ~1 % of the rows of an ordinary batch leave
        # f16's range (Dense_0 unit 7 = relu(200 x), Dense_1[7 -> 3] = 200: 4e4 x, beyond 65504 for x > 1.64 — the first samples of the ~9 % of
        # the rays that start on that side), 50 steps, every tenth batch such an ordinary one, the others drawn from rays that stay inside:
        # what a re-run costs at bench size, in the lagged default and decided in place, and how far the two parameter sets end up apart
        def retry_run(mode, hot_every):
            vv = models_fresh_variables(pf, device)
            fl = U.default_flags(num_coarse_samples=cfg["S"], num_fine_samples=fine, num_path_samples=cfg["P"], white_bkgd=False, bg_weight=0.025,
                                 bg_smooth_weight=0.0, use_online_sparsity=False, randomized=True, near=cfg["near"], far=cfg["far"],
                                 batch_size=B * world, backward_precision="f16x3", stage="radiance", range_retry=mode, lr_delay_steps=0)
            ts = TrainState.create(model, vv, fl)
            lo_c = ts.segments["coarse_mlp"][0]
            ts.theta[lo_c:lo_c + 256] = 0.0
            ts.theta[lo_c + 7] = 200.0
            ts.theta[lo_c + 63 * 256 + 256 + 7 * 256 + 3] = 200.0
            po, pdv = syn.sphere_rays(16 * B, seed=syn.SEED + 77 + rank)
            hot = np.maximum(po[:, 0] + cfg["near"] * pdv[:, 0], po[:, 0] + cfg["far"] * pdv[:, 0]) > 1.55
            cool_idx = np.nonzero(~hot)[0]
            gen = np.random.default_rng(syn.SEED + 78 + rank)
            mk = lambda idx: {"rays": Rays(torch.from_numpy(po[idx]).to(device), None, torch.from_numpy(pdv[idx]).to(device), None),
                              "pixels": torch.from_numpy(gen.uniform(0, 1, (B, 3)).astype(np.float32)).to(device), "annealed_alpha": 0.5}
            cool = [mk(cool_idx[i * B:(i + 1) * B]) for i in range(4)]
            ordinary = mk(np.arange(B))                    # as drawn: ~9 % of its rays start on the hot side
            from samplenerfro_amd.train import flush_range_retry
            rk = key
            n_steps, losses = 50, []
            for k in range(-5, n_steps):
                if k == 0:
                    barrier(); t0 = time.perf_counter()
                bt = ordinary if (hot_every and k >= 0 and k % hot_every == hot_every // 2) else cool[k % 4]
                ts, stt, rk = train_step(model, rk, ts, bt, fl)
                losses.append(stt.loss)
            flush_range_retry(model, ts)
            barrier()
            dt_r = time.perf_counter() - t0
            fin = [bool(torch.isfinite(x)) for x in losses[5:]]
            out = {"ms_per_step": 1e3 * dt_r / n_steps, "re_runs": ts.range_retries, "re_runs_failed": ts.range_retry_failures,
                   "steps_whose_first_attempt_was_skipped": fin.count(False), "parameters_finite": bool(torch.isfinite(ts.theta).all())}
            th = ts.theta.clone()
            del ts, vv
            return out, th

        live_lag, th_lag = retry_run("lag", 10)
        live_in, th_in = retry_run(True, 10)
        live_none, _ = retry_run("lag", 0)
        legs["range_retry_live"] = {"lagged_default": live_lag, "decided_in_place": live_in, "same_loop_without_a_batch_out_of_range": live_none,
                                    "ms_per_re_run": (live_lag["ms_per_step"] - live_none["ms_per_step"]) * 50 / max(live_lag["re_runs"], 1),
                                    "max_abs_theta_lagged_vs_in_place": float((th_lag - th_in).abs().max()),
                                    "what": "50 steps of 4096 x 128; every tenth batch has ~1 % of its rows outside f16's range: its update is skipped on the "
                                            "device and the batch re-run in bf16x3 + bf16 (two steps later by default, or in place); the two orders differ by "
                                            "the position of 5 updates (|dtheta| of a few learning rates); trajectory against the float64 loop: "
                                            "tests/test_gpu_train.py::test_range_retry_trajectory_follows_the_float64_loop_over_50_steps"}
        del th_lag, th_in, g_ref
        torch.cuda.empty_cache()

    traffic, sq, pmc_meta = pmc_lookup(args.workload, fine, B, args.mode, args.backward, rnerf_cus, prec_fwd_name)

    def traffic_of(prefix):
        for k, v in traffic.items():
            if k.startswith(prefix):
                return v
        return None

    # the PMC passes of the forward workload were taken in the DEFAULT eval precision: with another one (--eval-precision f16x3) the profile holds
    # that precision's kernel only as the gated-off fallback launch (a few KB of traffic) — no counter fields then
    pmc_matches_precision = True       # (pmc_lookup's file name carries the precision)

    def counters_of(prefix):
        for k, v in sq.items():
            if k.startswith(prefix):
                return v
        return None

    # ---- the harder variants of the same metric, in the same driver-run record ------------------------------------
    variants = None
    if train and args.extra and args.stage == "radiance" and args.workload == "ship_straight" and args.rays is None and args.fine is None and args.precision == "f16x3":
        variants = {}

        def run_variant(tag, vcfg, vmodel, vvars, vfine, vB, note, vstage="radiance"):
            ov, dv = syn.sphere_rays(vB, seed=syn.SEED + rank)
            vrays = Rays(torch.from_numpy(ov).to(device), None, torch.from_numpy(dv).to(device), None)
            sv = Stepper(args, vcfg, vmodel, vvars, vrays, key, vB, world, rank, vfine, device, args.backward, "train", vstage,
                         args.pipeline and vstage == "radiance", args.graph and vstage == "radiance")
            nv = 5 if vB >= 4096 else 20          # sub-millisecond steps: 5 timed steps are ~5 ms of wall clock — too few to be a steady state
            dtv = timed_steps(sv, 2 if vB >= 4096 else 4, nv, barrier, D, device)
            sv.close()
            torch.cuda.empty_cache()
            variants[tag] = {"ms_per_step": 1e3 * dtv / nv, "rays_per_s": vB * world * nv / dtv, "rays_per_gpu": vB, "steps": nv, "coarse": vcfg["S"],
                             "fine": vfine, "what": note}

        vm = models_with_fine(model, cfg, 256, device, args.precision)
        vR = args.variant_rays        # (4096; tests/test_gpu_bench_world8.py puts eight ranks on one device and shrinks the 35 GB-per-rank hierarchical legs)
        run_variant("ship_straight_128+256", cfg, vm[0], vm[1], 256, vR, "BASELINE configs[1], hierarchical (N_f = 2S: 512 MLP rows per ray)")
        del vm
        del model, variables
        torch.cuda.empty_cache()
        rcfg = dict(syn.CONFIGS["ship_refractive"])
        t0 = time.perf_counter()
        rmodel, rvars, _ = build_scene(rcfg, device, args.precision, 0, "radiance")
        torch.cuda.synchronize()
        t_refr = time.perf_counter() - t0
        run_variant("ship_refractive_128", rcfg, rmodel, rvars, 0, vR, "BASELINE configs[2]: 512^3 sphere grid after the (9, 3.0) prefilter; the speculative "
                    "march mispredicts where rays bend")
        vm = models_with_fine(rmodel, rcfg, 256, device, args.precision)
        run_variant("ship_refractive_128+256", rcfg, vm[0], vm[1], 256, vR, "BASELINE configs[2], hierarchical")
        del vm, rmodel, rvars
        torch.cuda.empty_cache()
        amodel, avars, _ = build_scene(rcfg, device, args.precision, 0, "all")
        run_variant("ship_refractive_128_stage_all", rcfg, amodel, avars, 0, vR, "stage all* (train.py:302-310): so3_mlp bends the gradient inside the march "
                    "(evaluated by four waves per 16-ray workgroup at every node of the boundary shell) and is trained through the march's adjoint; the "
                    "step includes the deterministic (node, ray) order of the shell pairs — one sort + two gathers, ~0.35 ms of device time — and "
                    "the shell-coherent ray order of the batch (a coarse pre-march + a sort, ~0.3 ms)", "all")
        del amodel, avars
        torch.cuda.empty_cache()
        dcfg = dict(syn.CONFIGS["dolphin_train"])
        dmodel, dvars, _ = build_scene(dcfg, device, args.precision, dcfg["F"], "radiance")
        for vB, note in ((4096, "BASELINE configs[3] on one GPU: 64 + 128 samples, 256^3 grid, global batch 4096"),
                         (1024, "the reference's default batch (configs/example.yaml:20)"), (512, "one GPU's shard of 4096 rays over 8 GPUs"),
                         (128, "one GPU's shard of the reference's default batch (1024 rays) over 8 GPUs")):
            run_variant("dolphin_train_%d" % vB, dcfg, dmodel, dvars, dcfg["F"], vB, note)
        if world > 1:
            # BASELINE configs[3] as written — STRONG scaling: the global batch split over the ranks that ran (train.py:196)
            for gB in (4096, 1024):
                if gB % world == 0:
                    run_variant("dolphin_train_global%d_strong" % gB, dcfg, dmodel, dvars, dcfg["F"], gB // world,
                                "global batch %d = %d rays per GPU on %d GPUs, gradient all-reduce included (rays_per_s is the whole job's)" % (gB, gB // world, world))
        del dmodel, dvars
        torch.cuda.empty_cache()
        variants["scene_build_s"] = {"ship_straight (table only)": t_scene, "ship_refractive (sphere + (9, 3.0) prefilter + table, 512^3)": t_refr}
        model, variables, pf = build_scene(cfg, device, args.precision, fine, args.stage, args.eval_precision)

    frame = None
    if args.frame:
        # ms/frame @ 800x800 (BASELINE.json metric 2): pinhole rays of the example camera looking at the volume, rendered in
        # chunks; each rank renders its own full frame here (the sharded variant is distributed.render_image_sharded)
        from samplenerfro_amd import utils as U
        H = W = 800
        focal = 0.5 * W / np.tan(0.5 * 0.6911112070083618)          # example_data/transforms_train.json camera_angle_x
        c2w = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 4.0]], np.float32)                  # camera on +z at distance 4
        o_w, _, v_w = ops.generate_rays(c2w, H, W, device, focal=focal)                           # rays are generated on the device
        fr = Rays(o_w, None, v_w, None)
        fn = lambda k0, k1, r, path=None: model.apply(variables, k0, k1, r, False, path=path)
        model.release_reserved_cus()
        chunk = 8192 * 4
        U.render_image(fn, fr, key, False, chunk=chunk)
        barrier()
        t1 = time.perf_counter()
        rgb_img, _, _ = U.render_image(fn, fr, key, False, chunk=chunk)
        barrier()
        frame = {"ms_per_frame": 1e3 * D.max_over_ranks(time.perf_counter() - t1, device), "height": H, "width": W, "samples": cfg["S"] + fine,
                 "chunk": chunk, "finite": bool(torch.isfinite(rgb_img).all()), "precision": args.eval_precision,
                 "what": "every rank renders the WHOLE frame (one GPU's time)"}

        def sharded_frame(mdl, vrs, rays_hw, ck):
            """BASELINE configs[4]'s form: the image rows sharded over the ranks (eval.py:95-105, rnerf/utils.py:353-370), no collective;
            the frame is done when the slowest rank is."""
            f2 = lambda k0, k1, r, path=None: mdl.apply(vrs, k0, k1, r, False, path=path)
            D.render_image_sharded(f2, rays_hw, key, False, chunk=ck, gather=False)
            barrier()
            t2 = time.perf_counter()
            blk = D.render_image_sharded(f2, rays_hw, key, False, chunk=ck, gather=False)[0]
            barrier()
            return 1e3 * D.max_over_ranks(time.perf_counter() - t2, device), blk

        if world > 1:
            ms_sh, blk = sharded_frame(model, variables, fr, chunk)
            lo, hi = D.shard_bounds(H, world, rank)
            frame["ms_per_frame_sharded"] = ms_sh
            frame["sharded"] = {"ranks": world, "rows_per_rank": hi - lo, "block_equals_full_frame_rows": bool(torch.equal(blk, rgb_img[lo:hi])),
                                "what": "contiguous row blocks per rank, no collective (distributed.render_image_sharded)"}
        if args.extra and args.stage == "radiance" and args.workload == "ship_straight" and args.rays is None and args.fine is None:
            # BASELINE configs[4]: glass, 800 x 800, 256 samples per ray, P = 24 -> N = 6144 eikonal steps, 384^3 grid after (5, 3.0)
            gcfg = dict(syn.CONFIGS["glass_frame"])
            del rgb_img
            gm, gv, _ = build_scene(gcfg, device, args.precision, 0, "radiance", args.eval_precision)
            gfn = lambda k0, k1, r, path=None: gm.apply(gv, k0, k1, r, False, path=path)
            gchunk = 16384
            if world > 1:
                ms_g, _blk = sharded_frame(gm, gv, fr, gchunk)
            else:
                U.render_image(gfn, fr, key, False, chunk=gchunk)
                barrier()
                t1 = time.perf_counter()
                g_img, _, _ = U.render_image(gfn, fr, key, False, chunk=gchunk)
                barrier()
                ms_g = 1e3 * (time.perf_counter() - t1)
            frame["glass_frame"] = {("ms_per_frame_sharded" if world > 1 else "ms_per_frame"): ms_g, "ranks": world, "samples": gcfg["S"],
                                    "eikonal_steps": gcfg["S"] * gcfg["P"], "grid": gcfg["G"], "chunk": gchunk,
                                    "what": "BASELINE configs[4]: 800 x 800 x 256 samples, N = 6144, rows sharded over the ranks that ran"}
            del gm, gv
            torch.cuda.empty_cache()
            frame["glass_frame"]["precision"] = args.eval_precision
            rgb_img = None
        if args.extra and args.stage == "radiance":
            # the same frame in the other render arithmetic — f16f8 (opt-in since round 6) beside the default f16x3, or the training arithmetic
            # beside an --eval-precision of the caller's: its time and the largest colour difference between the two frames (same weights:
            # build_scene is seeded) — and the default frame at the reference's own chunk size (rnerf/utils.py:241-244: 8192 rays)
            other = "f16f8" if args.eval_precision == args.precision == "f16x3" else (args.precision if args.eval_precision != args.precision else None)
            if other is not None:
                if rgb_img is None:
                    rgb_img = U.render_image(fn, fr, key, False, chunk=chunk)[0]
                m8, v8, _ = build_scene(cfg, device, args.precision, fine, args.stage, other)
                fn8 = lambda k0, k1, r, path=None: m8.apply(v8, k0, k1, r, False, path=path)
                U.render_image(fn8, fr, key, False, chunk=chunk)
                barrier()
                t1 = time.perf_counter()
                rgb8, _, _ = U.render_image(fn8, fr, key, False, chunk=chunk)
                barrier()
                frame[other] = {"ms_per_frame": 1e3 * D.max_over_ranks(time.perf_counter() - t1, device),
                                "max_abs_rgb_vs_the_frame_above": float((rgb8 - rgb_img).abs().max()), "finite": bool(torch.isfinite(rgb8).all()),
                                "what": ("opt-in render arithmetic (f16 main term + fp8 cross terms): within 1e-4 RGB of the oracle on these glorot-initialised "
                                         "weights, 2e-4 on trained-like ones — not the default" if other == "f16f8" else "the training arithmetic")}
                del m8, v8, rgb8
                torch.cuda.empty_cache()
            U.render_image(fn, fr, key, False, chunk=8192)
            barrier()
            t1 = time.perf_counter()
            U.render_image(fn, fr, key, False, chunk=8192)
            barrier()
            frame["ms_per_frame_chunk8192"] = 1e3 * D.max_over_ranks(time.perf_counter() - t1, device)
    if rank == 0:
        total_rays = B * args.steps * world
        rows_per_ray = S + (S + fine if fine > 0 else 0)
        flop_per_ray_fwd = rows_per_ray * MLP_FLOP_PER_ROW + BKGD_FLOP_PER_RAY
        line = {
            "metric": "rays/sec (train step)" if train else "rays/sec (forward render pass)", "value": total_rays / dt, "unit": "rays/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": dtype_label(args.precision if train else args.eval_precision, args.backward, train), "data": "synthetic",
            "config": {"workload": f"{args.workload}: {'train step (forward + backward + grad all-reduce + Adam)' if train else 'forward render pass'}, "
                                   f"{B} rays/GPU x {S} coarse + {fine} fine samples, "
                                   f"P={cfg['P']} (N={N} eikonal steps), grid {cfg['G']}^3", "rays_per_gpu": B,
                       "mlp_rows_per_ray": rows_per_ray, "pass": args.mode, "stage": args.stage,
                       "precision": prec_fwd_name + ": forward — " + PRECISION_NOTES.get(prec_fwd_name, "single 16-bit MFMA per product, fp32 accumulate"),
                       "eval_precision": args.eval_precision,
                       "backward_precision": (None if not train else args.backward),
                       "host_gc": "gc.collect() + gc.freeze() after set-up, before the warm-up steps: a generation-2 collection of the whole import graph takes 45 ms "
                                  "of host time and used to land in one 20-step window per run (bench.py; --trace-steps shows it)",
                       "backward_precision_note": (None if not train else {
                           "f16x3": "row-normalised f16 hi + lo parts of every saved activation and gradient (22 bits), 3 MFMAs per product: within 1e-5 of max|g| vs float64",
                           "f16x3lo8": "f16x3 with the lo planes of the saved activations / gradients stored as e4m3 bytes and decoded in the wgrad (11 + 4 significand "
                                       "bits per operand, 3/4 of the operand bytes): 1.6e-5 of max|g| vs float64 at 581 rows — outside the default's 1e-5: a labelled mode",
                           "f16": "row-normalised f16 parts (11-bit significand), 1-2 MFMAs per product: ~1e-3 of max|g| on small batches, ~1e-5 at this size",
                           "bf16": "bf16 parts (8-bit significand), round 1's arithmetic: ~6e-3 of max|g| on small batches"}[args.backward]),
                       "launch": ("one hipGraph launch per step (key split, march branch, forward, backward, Adam: csrc/pipeline.hip)" if graph_used
                                  else ("two whole-path C calls per step (rnerf_train_forward_backward + rnerf_adam_update, csrc/pipeline.hip): "
                                        "every kernel of the step is sequenced in librnerf.so, none decided on the host" if train
                                        else "one whole-path C call per batch (rnerf_forward)")),
                       "pipeline": ("none" if not args.pipeline else ("march(k+1) on a side branch beside the tail of step k" if train
                                                                          else "march(k+1) on a side stream overlaps MLP(k)"))},
            "roofline": {"kernel": "nerfmlp_fwd_kernel", "bound": "mfma", "achieved": mlp_achieved / 1e12, "peak": PEAK_MFMA_16BIT / 1e12,
                         "unit": "TFLOP/s", "frac": mlp_achieved / PEAK_MFMA_16BIT, "traffic": traffic_of("nerfmlp_fwd_kernel<%d, 0, 0," % prec_fwd) if pmc_matches_precision else None,
                         "avg_launch_ms": mlp_ms, "algorithmic_flop_per_launch": mlp_flops,
                         # computed, not a counter: MFMA flops issued (3 passes in the x3 modes) / (launch time x 2.5 PF)
                         "precision": prec_fwd_name,
                         "mfma_issue_frac_computed": {"f16x3": 3, "bf16x3": 3, "f16x2": 2, "f16f8": 3}.get(prec_fwd_name, 1) * mlp_achieved / PEAK_MFMA_16BIT,
                         "counters": counters_of("nerfmlp_fwd_kernel<%d, 0, 0," % prec_fwd) if pmc_matches_precision else None,
                         "passes": MFMA_PASSES.get(prec_fwd_name, 1), "sustained_mfma_tflops": (sustained or {}).get("bf16" if prec_fwd_name.startswith("bf16") else "f16"),
                         "frac_of_pass_ceiling": (MFMA_PASSES.get(prec_fwd_name, 1) * mlp_achieved / 1e12 / (sustained or {}).get("bf16" if prec_fwd_name.startswith("bf16") else "f16"))
                         if sustained else None},
            "roofline_march": {"kernel": "march_kernel", "bound": "hbm", "achieved": march_achieved / 1e9, "peak": PEAK_HBM / 1e9,
                               "unit": "GB/s", "frac": march_achieved / PEAK_HBM, "traffic": traffic_of("march_kernel"), "avg_launch_ms": march_ms,
                               "algorithmic_bytes_per_launch": march_bytes},
        }
        if train:
            # the dominant kernel of a train step: the longest of training forward / dgrad / wgrad (algorithmic FLOP against the MFMA peak)
            fk = "nerfmlp_fwd_kernel<%d, 0, %d," % (_lib.PRECISIONS[args.precision], {"f16x3": 2, "f16x3lo8": 3}.get(args.backward, 1))      # (+ the tile-size argument)
            for tk, pref in zip(train_kernels, (fk, "nerfmlp_dgrad_kernel", "nerfmlp_wgrad")):
                tk["traffic"] = traffic_of(pref)
                tk["counters"] = counters_of(pref)
            line["roofline_forward_kernel"] = line["roofline"]
            # plain maximum of the measured stand-alone launch times; the three are within a few % of each other, so the line also says how
            # close the runner-up is (in the step itself the wgrad hosts the next batch's march and is the longest: profiles/r04/train_step_timeline.txt)
            by_ms = sorted(train_kernels, key=lambda t: -t["avg_launch_ms"])
            line["roofline"] = dict(by_ms[0])
            line["roofline"]["tie_within_frac"] = 1.0 - by_ms[1]["avg_launch_ms"] / by_ms[0]["avg_launch_ms"]
            line["roofline"]["runner_up"] = by_ms[1]["kernel"]
            line["roofline_train_kernels"] = train_kernels
            # the whole step against the MFMA peak: SURVEY §8(d)'s algorithmic FLOP (forward + dgrad + wgrad ~ 3 x forward) / ms_per_step
            step_flop = 3.0 * flop_per_ray_fwd * B
            line["roofline_step"] = {"bound": "mfma", "algorithmic_flop_per_step": step_flop, "achieved": step_flop / (dt / args.steps) / 1e12,
                                     "peak": PEAK_MFMA_16BIT / 1e12, "unit": "TFLOP/s", "frac": step_flop / (dt / args.steps) / PEAK_MFMA_16BIT,
                                     "note": "3 x (mlp_rows_per_ray x 1 186 816 + 112 896) FLOP per ray; the fp32-grade modes issue 3 MFMAs per product, "
                                             "so MFMA issue is ~3 x this fraction (DESIGN.md §4: the 3-pass floor)"}
        line["sustained_mfma"] = {"tflops": sustained, "what": "back-to-back v_mfma_f32_32x32x16 on every CU, no memory traffic, measured in this run "
                                  "(csrc/ubench/mfma_rate.hip): the matrix pipe's ceiling under this chip's power limit; `frac_of_pass_ceiling` of a "
                                  "roofline object = achieved x passes / this"}
        line["collectives"] = {"backend": (dist.get_backend() if dist.is_initialized() else None), "ranks": world,
                               "per_step": ("none" if not (train and D.active()) else
                                            "all-reduce(mean) of the flat gradient + stats buffer between the backward and the update; the NerfMLP "
                                            "segments (95 % of the bytes) start on a side stream right behind the last wgrad, beside the step's tail")}
        if coll:
            line["collectives"].update(coll)
        if replicas:
            line["collectives"]["replicas"] = replicas
        if curve:
            line["scaling_curve"] = curve
        if stability:
            line["stability"] = stability
        if pmc_meta is not None:
            line["pmc_profile"] = pmc_meta
        if other_modes:
            line["other_backward_modes"] = other_modes
        if graph_replay:
            line["graph_replay"] = graph_replay
        if variants:
            line["variants"] = variants
        if train and args.workload == "ship_straight" and fine == 0 and B == 4096:
            # BASELINE.json north_star, as numbers: >= 1e7 rays/s whole-node on 8 GPUs at 128 samples per ray = 1.25e6 per GPU
            share = 1.25e6
            per_gpu = total_rays / dt / world
            ns = {"target_rays_per_s_per_gpu": share, "this_line_rays_per_s_per_gpu": per_gpu, "frac_of_target": per_gpu / share,
                  "arithmetic_of_this_line": dtype_label(args.precision, args.backward, True),
                  "what": "north_star asks for >= 1e7 rays/s on 8 x MI355X (1.25e6 per GPU) in bf16-class MFMA arithmetic with <= 1e-4 RGB error; the "
                          "headline is fp32-grade (3 MFMAs per product), the single-pass leg is the arithmetic north_star names"}
            if legs and "train" in legs.get("f16", {}):
                ns["single_pass_f16_leg_rays_per_s_per_gpu"] = legs["f16"]["train"]["rays_per_s"] / world
                ns["single_pass_f16_leg_frac_of_target"] = legs["f16"]["train"]["rays_per_s"] / world / share
            line["north_star"] = ns
        if legs:
            line["precision_legs"] = legs
            legs["what"] = ("the arithmetic north_star names (one 16-bit MFMA per product) on the headline workload, next to the fp32-grade default: "
                            "never the headline (narrower than the reference's fp32); parity_vs_oracle is filled when the CPU-baseline leg runs")
        if frame is not None:
            line["frame"] = frame
        if not args.no_cpu_baseline and args.stage == "radiance":      # (the oracle legs below restate the radiance stage)
            cpu_rps, cpu_dt = cpu_baseline(cfg, pf, fine, args.cpu_rays, syn.SEED, train)
            what = ("one train step: numpy fp32 oracle for march/sampling + torch CPU fp32 autograd of train.py's loss_fn + Adam"
                    if train else "one pass of the numpy fp32 oracle")
            try:      # threads actually used by the heavy part: torch intra-op threads (train) / the BLAS pool behind numpy (forward)
                from threadpoolctl import threadpool_info
                blas = max([int(t.get("num_threads", 1)) for t in threadpool_info() if t.get("user_api") == "blas"] or [1])
            except Exception:
                blas = os.cpu_count()
            used = torch.get_num_threads() if train else blas
            par = parity_vs_oracle(model, pf, cfg, fine, device, others={k: v["_model"][0] for k, v in (legs or {}).items() if isinstance(v, dict) and "_model" in v})
            for k in [k for k, v in (legs or {}).items() if isinstance(v, dict) and "_model" in v]:
                legs[k]["forward"]["parity_vs_oracle"] = par.pop(k)
            line["parity"] = par
            line["cpu_baseline"] = {"value": cpu_rps, "unit": "rays/s", "cores": used, "host_cpus": os.cpu_count(), "kind": "port",
                                    "sample": f"{args.cpu_rays} rays of the same workload, {what} (threaded BLAS); {CPU_WARMUP} warm-up + median of "
                                              f"{CPU_TIMED} timed passes, {cpu_dt:.1f} s each"}
        for v in (legs or {}).values():
            if isinstance(v, dict):
                v.pop("_model", None)
        print(json.dumps(line))
    if dist.is_initialized():
        dist.destroy_process_group()


def model_with_precision(model, precision):
    """The same scene (shared IoR table) with every NerfMLP pass in `precision`."""
    import copy
    from samplenerfro_amd import _lib
    m2 = copy.copy(model)
    m2.precision = m2.eval_precision = _lib.PRECISIONS[precision]
    m2._packed, m2._jit_cache, m2._ws, m2._key_cache, m2._u_lin, m2._side = {}, {}, {}, {}, None, None
    return m2


def models_fresh_variables(pf, device):
    import torch
    from samplenerfro_amd import models
    return models.make_variables({k: torch.from_numpy(v).to(device) for k, v in pf.items()})


def models_with_fine(model, cfg, fine, device, precision):
    """The same scene with num_fine_samples = `fine`: shares the (up to 2 GB) IoR table of `model` instead of rebuilding it."""
    import copy
    import torch
    from samplenerfro_amd import models, synthetic as syn
    m2 = copy.copy(model)
    m2.num_fine_samples = fine
    m2.fine_step_size = (m2.far - m2.near) / (m2.num_coarse_samples + fine)
    m2._packed, m2._jit_cache, m2._ws, m2._key_cache, m2._u_lin, m2._side = {}, {}, {}, {}, None, None
    pf = syn.init_params_flat(0, fine=True)
    return m2, models.make_variables({k: torch.from_numpy(v).to(device) for k, v in pf.items()})


if __name__ == "__main__":
    main()
