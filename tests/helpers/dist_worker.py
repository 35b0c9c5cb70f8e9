"""Worker of tests/test_rank_failure.py: a data-parallel loop with the path's collectives (distributed.allreduce_mean_ / barrier) over
gloo on CPU tensors.  argv: <steps> <fail_rank> <fail_mode: none|exit|hang> <timeout_s>.  Started by torch.distributed.run."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from samplenerfro_amd import distributed as D

steps, fail_rank, mode, timeout_s = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], float(sys.argv[4])
rank, world = D.init("gloo", timeout_s=timeout_s)
g = torch.full((1 << 16,), float(rank + 1))
for i in range(steps):
    if i == 2 and rank == fail_rank:
        if mode == "exit":
            os._exit(17)                 # dies without a goodbye (no destroy_process_group, no exception)
        if mode == "hang":
            time.sleep(3600)             # alive but never joins the collective again
    buf = g.clone()
    D.allreduce_mean_([buf])
    assert abs(float(buf[0]) - (world + 1) / 2.0) < 1e-6
D.barrier()
print(f"rank {rank} done", flush=True)
torch.distributed.destroy_process_group()
