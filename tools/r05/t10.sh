mkdir -p gpurun_out/r05
python -m pytest tests/test_gpu_train.py tests/test_gpu_bench_world2.py "tests/test_gpu_bench_world8.py::test_eight_ranks_small" -x -q -m gpu -s 2>&1 | grep -v "^$" | tail -25 > gpurun_out/r05/t10.log
( time python bench.py > gpurun_out/r05/b_time.json 2>/dev/null ) 2>> gpurun_out/r05/t10.log
python -c "import __graft_entry__ as g; g.smoke()" >> gpurun_out/r05/t10.log 2>&1
