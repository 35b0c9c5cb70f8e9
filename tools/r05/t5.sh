mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_example_scene.py -x -q -m gpu -s --durations=5 2>&1 | tail -40 > gpurun_out/r05/t5.log
