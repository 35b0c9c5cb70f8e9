#!/bin/bash
# round-2 baseline on the GPU box: gpu tests, MFMA microbenchmarks (with clocks), SQ counters of the MLP kernels, bench line
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02; mkdir -p $O
cd $R
timeout 900 python -m pytest tests -m gpu -x -q > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log
tail -3 $O/gputest.log
for u in mfma_ub mfma_fill; do
  hipcc --offload-arch=gfx950 -O3 tools/ubench/$u.hip -o /tmp/$u && timeout 120 /tmp/$u > $O/ubench_$u.txt 2>&1
done
cat $O/ubench_mfma_ub.txt
(rocm-smi --showclocks 2>/dev/null | head -30) > $O/clocks_idle.txt
bash tools/pmc_sq.sh nerfmlp -- $R/tools/bwd_time.py > $O/sq_bwd_time.txt 2>&1
cat $O/sq_bwd_time.txt
timeout 600 python bench.py > $O/bench_train.json 2> $O/bench_train.err
cut -c1-400 $O/bench_train.json
