#!/bin/bash
# the single-pass f16 step (f16 forward + f16 backward with the one-pass dgrad) and the tests that hold its gradients
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_backward.py tests/test_gpu_train.py tests/test_gpu_whole_path.py -q -x -m gpu 2>&1 | tail -5
A="--precision f16 --backward f16 --no-frame --no-cpu-baseline --no-extra --steps 40 --warmup 5"
python3 bench.py $A 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('f16 + f16', d['value'], d['ms_per_step'], [ (k['kernel'], round(k['avg_launch_ms'],3)) for k in d['roofline_train_kernels']])"
