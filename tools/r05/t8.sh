mkdir -p gpurun_out/r05
python bench.py --workload ship_refractive --stage all --no-cpu-baseline --no-frame --no-extra --steps 20 > gpurun_out/r05/b_all.json 2>gpurun_out/r05/b_all.err
