mkdir -p gpurun_out/r05
for i in 1 2; do
python bench.py --mode forward --precision f16 --no-extra --no-cpu-baseline --no-frame --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('fwd f16', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'])"
python bench.py --precision f16 --backward f16 --no-extra --no-cpu-baseline --no-frame --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('train f16', d['value'], d['ms_per_step'], [round(t['avg_launch_ms'],3) for t in d['roofline_train_kernels']])"
done > gpurun_out/r05/t11.log 2>&1
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "nerf_mlp" 2>&1 | tail -2 >> gpurun_out/r05/t11.log
