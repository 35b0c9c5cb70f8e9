#!/bin/bash
# the driver's 8-GPU command with eight ranks on ONE device over gloo (what tests/test_gpu_bench_world8.py::test_eight_ranks_the_drivers_command runs)
# -> gpurun_out/r05/rehearsal_8_ranks_one_device.json
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 8 --dist-backend gloo \
  --steps 3 --warmup 1 --no-cpu-baseline --variant-rays 1024 2> gpurun_out/r05/rehearsal.err | grep '^{' > gpurun_out/r05/rehearsal_8_ranks_one_device.json
echo "rc=$? bytes=$(wc -c < gpurun_out/r05/rehearsal_8_ranks_one_device.json)"
