mkdir -p gpurun_out/r05
python bench.py --no-frame --no-cpu-baseline --steps 10 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
print(d['precision_legs']['backward_modes_behind_the_f16x3_forward'], d['other_backward_modes'])
print(d['value'], d['north_star']['frac_of_target'])" > gpurun_out/r05/t16.log 2>&1
