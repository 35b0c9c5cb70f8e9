#!/bin/bash
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04/t2; mkdir -p $O; cd $R
timeout 1500 python3 -m pytest tests/test_gpu_checkpoint.py tests/test_gpu_bench_world2.py "tests/test_gpu_parity.py::test_f16x3_against_the_exact_fp32_arbiter" "tests/test_gpu_parity.py::test_nerf_mlp" tests/test_gpu_train.py tests/test_gpu_whole_path.py tests/test_gpu_rccl.py tests/test_gpu_train_2rank.py -q -s > $O/tests.log 2>&1; echo "tests rc=$?"; grep -E "passed|failed|FAILED|arbiter|level|\[x" $O/tests.log | head -40
