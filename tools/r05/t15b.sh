mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_train_all.py -x -q -m gpu 2>&1 | grep -i "error\|Error" | head -8 > gpurun_out/r05/t15.log
