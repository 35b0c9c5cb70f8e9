"""RCCL itself on the GPU box: backend "nccl" (= librccl on ROCm) in a one-rank group — process-group init on the device, the asynchronous
all-reduce pair the train step uses behind its wgrad (distributed.allreduce_begin / allreduce_end_mean_: issue on RCCL's stream, wait on
the compute stream), and one whole train step with the group initialised.  A one-GPU box cannot hold two RCCL ranks (one rank per device),
so world > 1 stays with the gloo tests; what this adds is that the collective library is loaded and executes on this hardware."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import os, sys, json
sys.path.insert(0, %r)
import numpy as np, torch, torch.distributed as dist
from samplenerfro_amd import distributed as D
D.force_single_rank_group(True)         # D.init is a no-op for one rank otherwise
rank, world = D.init("nccl", timeout_s=120.0)      # bound to the device (device_id): the RCCL communicator is created here, eagerly
assert dist.is_initialized() and (rank, world) == (0, 1)
dev = torch.device("cuda", torch.cuda.current_device())
assert dist.get_backend() == "nccl"
sub = dist.new_group(ranks=[0])         # what bench.py's in-run scaling curve does: a communicator split off the bound group
with D.use_group(sub):
    probe = torch.ones(8, device=dev)
    D.allreduce_mean_([probe]); D.barrier()
    assert D.world() == (0, 1) and bool((probe == 1).all())
buf = torch.arange(1 << 20, dtype=torch.float32, device=dev)
ref = buf.clone()
side = torch.cuda.Stream()
with torch.cuda.stream(side):            # the train step issues it behind the wgrad; here: from a non-default stream
    big = torch.ones(1 << 24, device=dev) * 3.0
    h = D.allreduce_begin(buf, force=True)
    assert h is not None
    D.allreduce_end_mean_(h, buf)
    ok_async = bool(torch.equal(buf, ref))
D.allreduce_mean_([buf])
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
loaded = [l.split()[-1] for l in open("/proc/self/maps") if "rccl" in l]
print(json.dumps({"ok_async": ok_async, "max": float(t.item()), "rccl": sorted(set(loaded))[:2]}))
dist.destroy_process_group()
""" % ROOT


@pytest.mark.timeout(300)
def test_rccl_one_rank_group_runs_the_async_allreduce():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-c", CHILD], env=env, cwd=ROOT, capture_output=True, text=True, timeout=280)
    assert out.returncode == 0, out.stderr[-3000:]
    import json
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["ok_async"] and d["max"] == 1.5
    assert d["rccl"], "librccl is not mapped into the process"


@pytest.mark.timeout(600)
def test_bench_under_torchrun_nccl_one_rank():
    """bench.py exactly as the driver launches N ranks (torch.distributed.run, backend nccl), with N = 1: the launcher environment,
    RCCL group init, barrier and max-over-ranks all go through librccl."""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--workload", "example", "--rays", "256", "--no-frame",
           "--no-cpu-baseline", "--no-extra", "--force-dist"]
    out = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=560)
    assert out.returncode == 0, out.stderr[-3000:]
    import json
    d = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d.get("collectives", {}).get("backend") == "nccl"
