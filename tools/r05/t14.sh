mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_example_scene.py -x -q -m gpu -s -k single_pass 2>&1 | grep "means: {\|passed\|failed" > gpurun_out/r05/t14.log
