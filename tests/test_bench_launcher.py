"""bench.py's --gpus contract on the CPU side: the flag is checked against the launcher's world size before anything touches a GPU, and a
bare `--gpus N` starts N ranks (each of which then stops at "needs a ROCm GPU" in this container)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLEAN = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}


def test_world_size_must_equal_gpus():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8"], env=dict(CLEAN, WORLD_SIZE="1", RANK="0"),
                         capture_output=True, text=True, timeout=120)
    assert out.returncode != 0 and "--gpus 8" in out.stderr and "WORLD_SIZE=1" in out.stderr


def test_bare_gpus_flag_starts_that_many_ranks():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("CPU-side check (the GPU-side one is tests/test_gpu_bench_world2.py)")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=CLEAN,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode != 0
    assert "bench.py needs a ROCm GPU" in out.stderr and " of 2]" in out.stderr, out.stderr[-2000:]      # a rank of a 2-rank launch
