mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_train.py -x -q -m gpu -s -k teacher 2>&1 | grep "means\|passed\|failed\|Error\|assert" > gpurun_out/r05/t19.log
