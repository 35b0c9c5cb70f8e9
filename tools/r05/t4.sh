mkdir -p gpurun_out/r05
python bench.py --steps 10 --warmup 3 > gpurun_out/r05/b_default.json 2> gpurun_out/r05/b_default.err; echo rc=$? >> gpurun_out/r05/b_default.err
python bench.py --steps 10 --warmup 3 --precision f16 --backward f16 --no-cpu-baseline --no-frame > gpurun_out/r05/b_train_f16.json 2> gpurun_out/r05/b_train_f16.err; echo rc=$? >> gpurun_out/r05/b_train_f16.err
python bench.py --steps 10 --warmup 3 --mode forward --precision bf16 --no-cpu-baseline > gpurun_out/r05/b_fwd_bf16.json 2> gpurun_out/r05/b_fwd_bf16.err; echo rc=$? >> gpurun_out/r05/b_fwd_bf16.err
python -m pytest tests/test_gpu_train.py tests/test_gpu_backward.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r05/t4.log
