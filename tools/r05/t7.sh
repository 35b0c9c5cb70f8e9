mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_train_all.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -8 > gpurun_out/r05/t7.log
bash tools/r05/prof_step.sh all1 march_all_kernel --workload ship_refractive --stage all
