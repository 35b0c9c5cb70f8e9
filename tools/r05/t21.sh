mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_train.py tests/test_gpu_parity.py tests/test_gpu_backward.py tests/test_gpu_whole_path.py tests/test_gpu_composite_lanes.py tests/test_golden_configs.py tests/test_gpu_c_host.py -x -q -m gpu 2>&1 | tail -12 > gpurun_out/r05/t21.log
