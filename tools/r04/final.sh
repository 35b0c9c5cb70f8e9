#!/bin/bash
# usage (GPU box, repo root): bash tools/r04/final.sh [part]  -> gpurun_out/r04/final/*: every number DESIGN.md section 4 / profiles/r04 quote.
# part 1 = bench lines, 2 = rocprof stats + timelines, 3 = PMC passes, 4 = small tools (default: all)
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r04/final; mkdir -p $O
cd $R
P=${1:-all}
if [ $P = all ] || [ $P = 1 ]; then
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"
python3 bench.py --mode forward > $O/bench_forward.json 2> $O/bench_forward.err; echo "forward rc=$?"
python3 bench.py --mode forward --eval-precision f16x3 --no-cpu-baseline --no-extra > $O/bench_forward_f16x3.json 2>/dev/null
python3 bench.py --backward tf32 --no-extra --no-frame --no-cpu-baseline > $O/bench_train_tf32.json 2>/dev/null
python3 bench.py --workload ship_refractive --no-extra --no-frame > $O/bench_ship_refractive.json 2>/dev/null
python3 bench.py --workload dolphin_train --no-extra --no-frame --no-cpu-baseline > $O/bench_dolphin_train.json 2>/dev/null
python3 bench.py --workload ship_refractive --stage all --no-extra --no-frame --no-cpu-baseline > $O/bench_stage_all.json 2>/dev/null
fi
if [ $P = all ] || [ $P = 2 ]; then
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o default -- python3 $R/bench.py > $O/prof_default_bench.json 2> $O/prof_default.err)
cp $O/prof/default_kernel_stats.csv $O/default_bench_kernel_stats.csv
bash tools/r04/prof_step.sh final_step > $O/step_stats_and_timeline.txt 2>&1
cp $R/gpurun_out/r04/final_step_kernel_stats.csv $O/train_step_kernel_stats.csv; cp $R/gpurun_out/r04/final_step_timeline.txt $O/train_step_timeline.txt
bash tools/r04/prof_step.sh final_step_dolphin512 --workload dolphin_train --rays 512 > $O/step_dolphin512.txt 2>&1
cp $R/gpurun_out/r04/final_step_dolphin512_timeline.txt $O/train_step_dolphin512_timeline.txt
bash tools/r04/prof_step.sh final_forward --mode forward > $O/step_forward.txt 2>&1
cp $R/gpurun_out/r04/final_forward_kernel_stats.csv $O/forward_kernel_stats.csv
fi
if [ $P = all ] || [ $P = 3 ]; then
bash tools/r04/pmc_all.sh ship_straight_f0_train_f32 > $O/pmc_train.txt 2>&1
bash tools/r04/pmc_all.sh ship_straight_f0_forward --mode forward > $O/pmc_forward.txt 2>&1
bash tools/r04/pmc_all.sh dolphin_train_f128_train_f32 --workload dolphin_train > $O/pmc_dolphin.txt 2>&1
cp $R/gpurun_out/r04/pmc_*.json $O/
fi
if [ $P = all ] || [ $P = 4 ]; then
RNERF_WGRAD_TRACE=1 python3 tools/bwd_time.py > $O/bwd_time_trace.txt 2>&1
python3 tools/march_time.py > $O/march_time.txt 2>&1
bash tools/r04/cmp_launch_modes.sh > $O/launch_modes.txt 2>&1
fi
rm -rf $O/prof $R/gpurun_out/r04/pmc $R/gpurun_out/r04/prof
ls $O
