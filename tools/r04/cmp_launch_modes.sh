mkdir -p gpurun_out/r04
X="--no-extra --no-frame --no-cpu-baseline --steps 30 --warmup 5"
for W in "--workload dolphin_train --rays 1024" "--workload dolphin_train --rays 512" ""; do
  for M in "graph:--graph" "eager_whole:" "staged:"; do
    name=${M%%:*}; fl=${M#*:}
    if [ "$name" = staged ]; then export RNERF_STAGED=1; else unset RNERF_STAGED; fi
    python bench.py $W $X $fl 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$name', '$W', round(d['ms_per_step'],3),'ms', int(d['value']),'rays/s')"
  done
done
