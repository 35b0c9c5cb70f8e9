mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_parity.py tests/test_gpu_whole_path.py tests/test_gpu_checkpoint.py tests/test_gpu_half_tiles.py tests/test_gpu_train.py -x -q -m gpu 2>&1 | tail -30 > gpurun_out/r05/t6.log
python bench.py --mode forward --steps 10 --no-cpu-baseline --no-extra > gpurun_out/r05/b_fwd.json 2>/dev/null
