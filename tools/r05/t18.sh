mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_example_scene.py -x -q -m gpu -s -k raises_the_psnr 2>&1 | grep "opacity\|dRGB\|passed\|failed\|Error" > gpurun_out/r05/t18.log
