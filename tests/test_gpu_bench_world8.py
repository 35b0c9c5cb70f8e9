"""The first 8-GPU run, rehearsed on one device: bench.py launched exactly like the driver launches it
(torch.distributed.run, --gpus 8), eight ranks on cuda:0 over gloo (RCCL refuses two ranks on one device).  What an 8-GPU node adds is
RCCL / xGMI underneath the same calls; everything above the backend — the launcher environment, one key and one ray batch per rank, the
in-step gradient all-reduce, the replicas' bits, BASELINE configs[3] as written (global batch 4096 = 512 rays per rank, strong), configs[4]
as written (800 rows = 100 rows per rank, no collective), the in-run scaling curve, a rank failure — runs here."""
import json
import os
import socket
import subprocess
import sys
import time

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _torchrun(n, extra, limit=900):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--dist-backend", "gloo"] + extra
    t0 = time.time()
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=limit)
    return out, time.time() - t0


def _line(out):
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]               # rank 0 prints the one line
    return json.loads(lines[0])


@pytest.mark.timeout(900)
@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_eight_ranks_small(scaling):
    out, _ = _torchrun(8, ["--steps", "3", "--warmup", "1", "--workload", "example", "--rays", "512", "--no-frame", "--no-cpu-baseline", "--no-extra",
                           "--scaling", scaling])
    d = _line(out)
    per = 512 if scaling == "weak" else 64
    assert d["n_gpus"] == 8 and d["scaling"] == scaling and d["config"]["rays_per_gpu"] == per and d["value"] > 0
    assert abs(d["value"] - 8 * per * 3 / (d["ms_per_step"] * 3e-3)) < 1e-6 * d["value"]
    c = d["collectives"]
    assert c["ranks"] == 8 and c["backend"] == "gloo" and c["allreduce_us"] > 0 and "exposed_us" in c
    r = c["replicas"]
    assert r["ranks"] == 8 and r["parameters_bit_identical"] is True and r["after_steps"] >= 3
    assert r["distinct_rank_keys"] == 8 and r["distinct_rank_batches"] == 8
    sc = d["scaling_curve"]
    assert sc["n"] == [1, 2, 4, 8] and len(sc["rays_per_s"]) == 4 and all(v > 0 for v in sc["rays_per_s"])


@pytest.mark.timeout(1500)
def test_eight_ranks_the_drivers_command():
    """`bench.py --gpus 8 --steps K --warmup W` with every default leg on at full size (4096 rays x 128 samples per rank on the 512^3 table, the
    variants, both frames); only K is shortened, the rank-0 CPU baseline left out and the hierarchical ship_* variants (35 GB of saved operands per
    rank at 4096 x 512 rows — a GPU's worth, but here eight ranks share one device's 288 GB) run 1024 rays per rank."""
    out, took = _torchrun(8, ["--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--variant-rays", "1024"], limit=1400)
    d = _line(out)
    assert d["n_gpus"] == 8 and d["config"]["rays_per_gpu"] == 4096 and d["scaling"] == "weak" and d["metric"] == "rays/sec (train step)"
    assert d["collectives"]["ranks"] == 8 and d["collectives"]["replicas"]["parameters_bit_identical"] is True
    assert d["scaling_curve"]["n"] == [1, 2, 4, 8]
    v = d["variants"]
    g = v["dolphin_train_global4096_strong"]            # BASELINE configs[3] as written
    assert g["rays_per_gpu"] == 512 and g["rays_per_s"] > 0 and v["dolphin_train_global1024_strong"]["rays_per_gpu"] == 128
    f = d["frame"]                                        # BASELINE configs[4]'s form: rows sharded, no collective
    assert f["sharded"]["ranks"] == 8 and f["sharded"]["rows_per_rank"] == 100 and f["sharded"]["block_equals_full_frame_rows"] is True
    assert f["glass_frame"]["ranks"] == 8 and f["glass_frame"]["ms_per_frame_sharded"] > 0 and f["glass_frame"]["eikonal_steps"] == 6144
    print("eight ranks on one device, the driver's command: %.0f s" % took)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("mode", ["exit", "hang"])
def test_a_failed_rank_takes_the_bench_down(mode):
    """One of two ranks dies (or stops responding) after its scene is built: the launch must end non-zero within the collective timeout and
    print no result line.  A restart is the same command again — a fresh set of child processes."""
    out, took = _torchrun(2, ["--steps", "3", "--warmup", "1", "--workload", "example", "--rays", "256", "--no-frame", "--no-cpu-baseline", "--no-extra",
                              "--fail-rank", "1", "--fail-mode", mode, "--dist-timeout", "15"], limit=500)
    assert out.returncode != 0
    assert not [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert took < 200, took
