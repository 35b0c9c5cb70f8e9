mkdir -p gpurun_out/r05
( time python bench.py --gpus 2 --steps 2 --warmup 1 > gpurun_out/r05/nccl2.out 2> gpurun_out/r05/nccl2.err; echo "rc=$?" ) > gpurun_out/r05/t13.log 2>&1
grep -v amdgpu gpurun_out/r05/nccl2.err | grep -i "bench.py\|error" | head -5 >> gpurun_out/r05/t13.log
