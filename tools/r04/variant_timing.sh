X="--no-extra --no-frame --no-cpu-baseline --workload dolphin_train --rays 128"
for sw in "5 2" "5 2" "30 5" "100 10"; do set -- $sw; python bench.py $X --steps $1 --warmup $2 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('steps $1 warmup $2:', round(d['ms_per_step'],3),'ms')"; done
