mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_train_all.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r05/t15.log
python tools/r05/ab_all.py 2>/dev/null | tail -3 >> gpurun_out/r05/t15.log
bash tools/r05/prof_step.sh all3 march_all_kernel --workload ship_refractive --stage all
grep "so3\|input_grad\|adjoint\|bkgd_wgrad_kernel<1>" gpurun_out/r05/all3_timeline.txt >> gpurun_out/r05/t15.log
