#!/bin/bash
# after a kernel-source change: the full GPU suite + smoke, then the PMC passes (final.sh 3).  Copy gpurun_out/r05/final/pmc_* to profiles/r05
# BEFORE running refresh12.sh (the committed lines look their counter fields up there).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
(time timeout 1500 python -m pytest tests -q -x -m gpu 2>&1 | tail -6) > gpurun_out/r05/t_all.log 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" >> gpurun_out/r05/t_all.log 2>&1
bash tools/r05/final.sh 3 > gpurun_out/r05/refresh3.log 2>&1
