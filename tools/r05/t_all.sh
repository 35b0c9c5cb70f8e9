#!/bin/bash
# full GPU suite + smoke + a default bench line (no profiles)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
(time timeout 1500 python -m pytest tests -q -x -m gpu 2>&1 | tail -15) > gpurun_out/r05/t_all.log 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> gpurun_out/r05/t_all.log 2>&1
python3 bench.py > gpurun_out/r05/t_all_bench.json 2> gpurun_out/r05/t_all_bench.err; echo "bench rc=$?" >> gpurun_out/r05/t_all.log
