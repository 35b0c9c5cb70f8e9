"""Build librnerf.so (hand-written HIP for gfx950) in-tree with hipcc.

`python -m samplenerfro_amd.build` or `samplenerfro_amd.build.build()`.  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "librnerf.so")
LIB_EXPERIMENTS = os.path.join(LIBDIR, "librnerf_experiments.so")
LIB_UBENCH = os.path.join(LIBDIR, "librnerf_ubench.so")       # csrc/ubench/mfma_rate.hip: the measured MFMA ceiling bench.py quotes (not the product)
SOURCES = ["grid.hip", "march.hip", "render.hip", "mlp.hip", "mlp_f32.hip", "bkgd16.hip", "pipeline.hip"]
# -ffp-contract=off + correctly rounded div/sqrt: the march/lookup/resample kernels reproduce the reference's
# individually rounded fp32 op order so that integer indices are bit-exact against the oracle.
# -fno-slp-vectorize: a performance choice first — hipcc's SLP pass packs adjacent scalar fp32 ops into v_pk_{mul,add,fma}_f32, which beside
# MFMAs are slower than the scalar forms (MI355X_MICROARCH.md: +22..26 cycles per gap).  It also keeps the schedule away from a hazard class
# LLVM cannot see: the one-line asm VALU statements in the MFMA shadows are opaque to its hazard recognizer, so no wait states are inserted
# between them and a dependent MFMA.  Round 2 met wrong values in a few lanes, run-to-run different, in one SLP-on schedule of the f16
# dgrad; tools/hazard_check.py scans the ISA of a build for the two hazard patterns (tests/test_hazards.py runs it on the product flags:
# 0 candidates) and tests/test_gpu_backward.py::test_backward_kernels_are_bit_stable_from_run_to_run checks the device.  (With the pass on,
# this ROCm's clang currently crashes on csrc/mlp.hip, so the SLP-on schedule itself can no longer be inspected.)
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-fhip-fp32-correctly-rounded-divide-sqrt", "-Wno-unused-value", "-fno-slp-vectorize"]


def _hipcc() -> str:
    for c in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (need ROCm >= 7.0 for gfx950)")


def _stale(out: str, deps) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(job) -> None:
    cmd, verbose = job
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)


def build(force: bool = False, verbose: bool = False, experiments: bool = True) -> str:
    """Compile every translation unit (in parallel) and link librnerf.so; returns its path.

    experiments=True also builds librnerf_experiments.so from the same sources with -DRNERF_EXPERIMENTS: the experiment / test switches
    (RNERF_* environment variables, csrc/common.h RNERF_ENV) exist only there — the product library reads no environment."""
    from concurrent.futures import ThreadPoolExecutor
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    headers = [os.path.join(CSRC, "common.h"), os.path.join(HERE, "..", "include", "rnerf.h"),
               os.path.join(CSRC, "ior_train_kernels.inc"), os.path.join(CSRC, "ior_train_api.inc"), os.path.join(CSRC, "nerfmlp_layout.h"),
               os.path.join(CSRC, "mfma_ops.h"), os.path.join(CSRC, "bkgd_layout.h"), os.path.join(CSRC, "so3_layout.h")]
    variants = [(LIB, LIBDIR, [])]
    if experiments:
        variants.append((LIB_EXPERIMENTS, os.path.join(LIBDIR, "exp"), ["-DRNERF_EXPERIMENTS"]))
    jobs, links = [], []
    for lib, objdir, extra in variants:
        os.makedirs(objdir, exist_ok=True)
        objs = []
        for src in SOURCES:
            s = os.path.join(CSRC, src)
            o = os.path.join(objdir, src.replace(".hip", ".o"))
            objs.append(o)
            if force or _stale(o, [s] + headers):
                jobs.append(([hipcc] + FLAGS + extra + ["-c", s, "-o", o], verbose))
        links.append((lib, objs))
    if jobs:
        # longest first (mlp.hip takes ~2 min, the rest seconds): the two mlp.o builds run side by side
        jobs.sort(key=lambda j: -os.path.getsize(j[0][-3]))
        with ThreadPoolExecutor(max_workers=min(len(jobs), max(2, (os.cpu_count() or 2) // 2))) as ex:
            list(ex.map(_compile, jobs))
    ub = os.path.join(CSRC, "ubench", "mfma_rate.hip")
    if force or _stale(LIB_UBENCH, [ub]):
        _compile(([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", ub, "-o", LIB_UBENCH], verbose))
    for lib, objs in links:
        if force or _stale(lib, objs):
            _compile(([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib], verbose))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, experiments="--no-experiments" not in sys.argv))
