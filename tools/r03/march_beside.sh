X="--no-extra --no-frame --no-cpu-baseline --steps 30 --warmup 5"
for V in 0 1; do
  export RNERF_MARCH_BESIDE_WGRAD=$V
  python bench.py $X 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('beside_wgrad=$V', round(d['ms_per_step'],3),'ms', int(d['value']),'rays/s', [(k['kernel'][:22], round(k['avg_launch_ms'],3)) for k in d['roofline_train_kernels']])"
done
