"""A rank that dies or stops responding must take the job down with a non-zero exit code inside the collective timeout — never hang it
(the reference's pmap has no such path: a lost device is a fatal XLA error).  CPU, gloo, launched like the driver
launches bench.py (torch.distributed.run).  A restart is a fresh launch: nothing here (or in the product) ever re-execs a process."""
import os
import socket
import subprocess
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "helpers", "dist_worker.py")


def _launch(world, steps, fail_rank, mode, timeout_s, limit):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(port), WORKER, str(steps), str(fail_rank), mode, str(timeout_s)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    t0 = time.time()
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=limit)
    return out, time.time() - t0


@pytest.mark.timeout(300)
def test_healthy_job_exits_zero():
    out, _ = _launch(4, 5, -1, "none", 30, 240)
    assert out.returncode == 0, out.stderr[-2000:]
    assert sum(("rank %d done" % r) in out.stdout for r in range(4)) == 4


@pytest.mark.timeout(300)
def test_a_rank_that_dies_fails_the_job():
    out, took = _launch(4, 50, 2, "exit", 30, 240)
    assert out.returncode != 0
    assert "rank 2 done" not in out.stdout and took < 120
    assert "exitcode" in out.stderr or "exit code" in out.stderr.lower() or "17" in out.stderr       # the launcher names the failed rank


@pytest.mark.timeout(300)
def test_a_rank_that_hangs_fails_the_job_within_the_timeout():
    timeout_s = 8
    out, took = _launch(2, 50, 1, "hang", timeout_s, 240)
    assert out.returncode != 0
    assert took < timeout_s + 90, took             # the survivor's all-reduce raised after timeout_s; the launcher then stopped the sleeper
    assert "rank 0 done" not in out.stdout


@pytest.mark.timeout(300)
def test_a_restart_after_a_failure_is_a_fresh_launch():
    out, _ = _launch(2, 50, 0, "exit", 20, 240)
    assert out.returncode != 0
    out, _ = _launch(2, 5, -1, "none", 20, 240)     # same command line minus the fault: new processes, new rendezvous port
    assert out.returncode == 0, out.stderr[-2000:]
