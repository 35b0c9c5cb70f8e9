mkdir -p gpurun_out/r05
python bench.py --no-frame --no-cpu-baseline --no-extra --steps 20 2>/dev/null | python -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][0])
for t in d['roofline_train_kernels']: print(t['kernel'], round(t['avg_launch_ms'],3), t.get('avg_launch_ms_alone'), t.get('launched'), round(t['frac'],4))
print(d['roofline']['kernel'], d['roofline']['tie_within_frac'], d['value'], d['ms_per_step'])" > gpurun_out/r05/t17.log 2>&1
