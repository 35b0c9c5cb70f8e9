mkdir -p gpurun_out/r05
for i in 1 2 3; do for L in reference bricks; do
python bench.py --table-layout $L --no-extra --no-cpu-baseline --no-frame --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$L', round(d['value']), round(d['ms_per_step'],4))"
done; done > gpurun_out/r05/t12.log 2>&1
for L in reference bricks; do
python bench.py --workload ship_refractive --table-layout $L --no-extra --no-cpu-baseline --no-frame --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('refractive $L', round(d['value']), round(d['ms_per_step'],4))"
done >> gpurun_out/r05/t12.log 2>&1
