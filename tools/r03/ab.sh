# usage: bash tools/r03/ab.sh [bench args]  -> one line with ms/step and the three big kernels (same box for A/B comparisons)
X="--no-extra --no-frame --no-cpu-baseline --steps 30 --warmup 5"
python bench.py $X "$@" 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$*', round(d['ms_per_step'],3),'ms', int(d['value']),'rays/s', [(k['kernel'][:22], round(k['avg_launch_ms'],3)) for k in d.get('roofline_train_kernels',[])])"
