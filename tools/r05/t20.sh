mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_parity.py tests/test_gpu_train.py tests/test_gpu_whole_path.py tests/test_gpu_backward.py tests/test_gpu_example_scene.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r05/t20.log
bash tools/r05/final.sh 3 > gpurun_out/r05/final3.log 2>&1
