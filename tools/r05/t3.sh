mkdir -p gpurun_out/r05
python -m pytest tests -x -q -m gpu --durations=15 2>&1 | tail -40 > gpurun_out/r05/t3.log
