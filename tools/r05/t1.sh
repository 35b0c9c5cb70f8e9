set -x
mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_rccl.py tests/test_gpu_bench_world2.py tests/test_gpu_bench_world8.py -x -q -m gpu --durations=10 -s 2>&1 | tail -40 > gpurun_out/r05/t1.log
free -g | head -2 >> gpurun_out/r05/t1.log; nproc >> gpurun_out/r05/t1.log
