#!/bin/bash
# usage (GPU box, repo root): bash tools/r06/final.sh [part]  -> gpurun_out/r06/final/*: every number DESIGN.md section 4 / profiles/r06 quote.
# part 2 = rocprof stats + timelines, 3 = PMC passes, 1 = bench lines (run LAST, after the PMC files of part 3 were copied to profiles/r06:
# a line's `traffic` / `counters` are looked up there).  Default: 2 then 3.
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r06/final; mkdir -p $O
cd $R
P=${1:-23}
if [[ $P == *2* ]]; then
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o default -- python3 $R/bench.py > $O/prof_default_bench.json 2> $O/prof_default.err)
cp $O/prof/default_kernel_stats.csv $O/default_bench_kernel_stats.csv; rm -rf $O/prof
bash tools/r06/prof_step.sh final_step rng_forward
cp $R/gpurun_out/r06/final_step_kernel_stats.csv $O/train_step_kernel_stats.csv; cp $R/gpurun_out/r06/final_step_timeline.txt $O/train_step_timeline.txt
bash tools/r06/prof_step.sh final_step_lo8 rng_forward --backward f16x3lo8
cp $R/gpurun_out/r06/final_step_lo8_kernel_stats.csv $O/train_step_lo8_kernel_stats.csv; cp $R/gpurun_out/r06/final_step_lo8_timeline.txt $O/train_step_lo8_timeline.txt
bash tools/r06/prof_step.sh final_forward rng_forward --mode forward
cp $R/gpurun_out/r06/final_forward_kernel_stats.csv $O/forward_kernel_stats.csv; cp $R/gpurun_out/r06/final_forward_timeline.txt $O/forward_timeline.txt
bash tools/r06/prof_step.sh final_all march_all_kernel --workload ship_refractive --stage all
cp $R/gpurun_out/r06/final_all_kernel_stats.csv $O/stage_all_kernel_stats.csv; cp $R/gpurun_out/r06/final_all_timeline.txt $O/stage_all_timeline.txt
fi
if [[ $P == *3* ]]; then
bash tools/r06/pmc_all.sh ship_straight_f0_train_f16x3 > $O/pmc_train.txt 2>&1
bash tools/r06/pmc_all.sh ship_straight_f0_train_f16x3lo8 --backward f16x3lo8 > $O/pmc_train_lo8.txt 2>&1
bash tools/r06/pmc_all.sh ship_straight_f0_forward --mode forward > $O/pmc_forward.txt 2>&1
cp $R/gpurun_out/r06/pmc_*.json $O/
rm -rf $R/gpurun_out/r06/pmc
fi
if [[ $P == *1* ]]; then
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"
python3 bench.py --mode forward > $O/bench_forward.json 2> $O/bench_forward.err; echo "forward rc=$?"
python3 bench.py --backward f16x3lo8 --no-frame --no-extra > $O/bench_train_lo8.json 2>/dev/null
python3 bench.py --mode forward --eval-precision f16f8 --no-extra > $O/bench_forward_f16f8.json 2>/dev/null
python3 bench.py --precision f16 --backward f16 --no-frame > $O/bench_train_f16.json 2>/dev/null
python3 bench.py --workload ship_refractive --no-extra --no-frame > $O/bench_ship_refractive.json 2>/dev/null
python3 bench.py --workload dolphin_train --no-extra --no-frame --no-cpu-baseline > $O/bench_dolphin_train.json 2>/dev/null
python3 bench.py --workload ship_refractive --stage all --no-extra --no-frame --no-cpu-baseline > $O/bench_stage_all.json 2>/dev/null
fi
ls $O
