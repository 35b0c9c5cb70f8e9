mkdir -p gpurun_out/r05
export HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 8 --dist-backend gloo --steps 3 --warmup 1 --no-cpu-baseline --variant-rays 1024 > gpurun_out/r05/t2.out 2> gpurun_out/r05/t2.err
echo rc=$? >> gpurun_out/r05/t2.out
grep -v "amdgpu.ids" gpurun_out/r05/t2.err | head -150 > gpurun_out/r05/t2.err.head
