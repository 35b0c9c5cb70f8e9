mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_train_all.py -x -q -m gpu 2>&1 | tail -5 > gpurun_out/r05/t9.log
python bench.py --workload ship_refractive --stage all --no-cpu-baseline --no-frame --no-extra --steps 20 > gpurun_out/r05/b_all2.json 2>gpurun_out/r05/b_all2.err
python - <<'P' >> gpurun_out/r05/t9.log 2>&1
import sys; sys.path.insert(0, ".")
from samplenerfro_amd import train
train._ALL_CHAIN_BESIDE_WGRAD = False
sys.argv = ["bench.py", "--workload", "ship_refractive", "--stage", "all", "--no-cpu-baseline", "--no-frame", "--no-extra", "--steps", "20"]
import runpy
runpy.run_path("bench.py", run_name="__main__")
P
bash tools/r05/prof_step.sh all2 march_all_kernel --workload ship_refractive --stage all
