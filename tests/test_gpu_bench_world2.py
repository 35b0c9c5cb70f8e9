"""bench.py's multi-rank branch end to end: two ranks launched exactly like the driver does (torch.distributed.run, --gpus 2), both on
cuda:0 with the gloo backend (--dist-backend gloo; RCCL refuses two ranks on one device).  Covers process-group init, per-rank keys and
rays, the in-step gradient all-reduce, the barrier + max-over-ranks timing and the single JSON line of rank 0."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_bench_two_ranks(scaling):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--steps", "3", "--warmup", "1", "--workload", "example", "--rays", "512", "--no-frame",
           "--no-cpu-baseline", "--scaling", scaling] + (["--graph"] if scaling == "strong" else [])      # both launch forms with two ranks
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=560)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]                       # rank 0 prints the one line
    d = json.loads(lines[0])
    per = 512 if scaling == "weak" else 256
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == scaling and d["config"]["rays_per_gpu"] == per
    assert d["metric"] == "rays/sec (train step)" and d["unit"] == "rays/s" and d["value"] > 0
    assert abs(d["value"] - 2 * per * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]      # whole-job rays / max-over-ranks time
    assert d["roofline"]["frac"] > 0 and d["config"]["backward_precision"] == "f16x3"
    c = d["collectives"]                       # the step's one exchange, timed
    assert c["ranks"] == 2 and c["allreduce_us"] > 0 and c["allreduce_bytes"] > 4 * 1_000_000 and "exposed_us" in c
    r = c["replicas"]                          # every rank applied the same update to the same reduced gradient; own key and rays per rank
    assert r["parameters_bit_identical"] is True and r["distinct_rank_keys"] == 2 and r["distinct_rank_batches"] == 2
    sc = d["scaling_curve"]                    # the same step on the first 1 and 2 ranks of this launch
    assert sc["n"] == [1, 2] and all(v > 0 for v in sc["rays_per_s"])
    assert d["stability"]["windows"] == 5 and d["stability"]["min_ms"] <= d["stability"]["median_ms"] <= d["stability"]["max_ms"]
    assert d["dtype"] == "f16x3/fp32-acc"


@pytest.mark.timeout(600)
def test_bench_frame_leg_is_sharded_over_the_ranks():
    """BASELINE configs[4]'s form with two ranks: every rank renders its contiguous block of image rows (eval.py:95-105,
    rnerf/utils.py:353-370; no collective) and the blocks equal the rows of the single-rank frame bit for bit."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--steps", "2", "--warmup", "1", "--workload", "example", "--rays", "512", "--mode", "forward",
           "--no-cpu-baseline", "--no-extra"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=560)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][0])
    f = d["frame"]
    assert f["ms_per_frame_sharded"] > 0 and f["sharded"]["ranks"] == 2 and f["sharded"]["rows_per_rank"] == 400
    assert f["sharded"]["block_equals_full_frame_rows"] is True and f["finite"]


@pytest.mark.timeout(600)
def test_bench_gpus_flag_launches_the_ranks_itself():
    """`python bench.py --gpus 2` WITHOUT a launcher (how the driver starts N = 1, and what a bare --gpus 8 must not silently turn into one
    rank): bench.py becomes the launcher, starts two ranks as children and relays rank 0's line."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--steps", "2", "--warmup", "1", "--workload", "example", "--rays", "256",
           "--no-frame", "--no-cpu-baseline", "--no-extra"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=560)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["rays_per_gpu"] == 256 and d["value"] > 0
