#!/bin/bash
# full GPU suite + smoke + a default bench line + the stage-all* line (no profiles)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
(time timeout 1500 python -m pytest tests -q -x -m gpu 2>&1 | tail -6) > gpurun_out/r05/t_all.log 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> gpurun_out/r05/t_all.log 2>&1
python3 bench.py > gpurun_out/r05/t_all_bench.json 2> gpurun_out/r05/t_all_bench.err; echo "bench rc=$?" >> gpurun_out/r05/t_all.log
python3 bench.py --workload ship_refractive --stage all --no-extra --no-frame --no-cpu-baseline > gpurun_out/r05/t_all_stage_all.json 2>/dev/null; echo "stage all rc=$?" >> gpurun_out/r05/t_all.log
