#!/bin/bash
# usage (GPU box, repo root): bash tools/r03/final.sh  -> gpurun_out/r03/final/* : every number DESIGN.md §4 / profiles/r03 quote, one box, one call
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r03/final; mkdir -p $O
cd $R
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "default rc=$?"
python3 bench.py --mode forward > $O/bench_forward.json 2> $O/bench_forward.err; echo "forward rc=$?"
python3 bench.py --backward tf32 --no-extra --no-frame --no-cpu-baseline > $O/bench_train_tf32.json 2>/dev/null
python3 bench.py --mode forward --precision f16x2 --no-cpu-baseline > $O/bench_forward_f16x2.json 2>/dev/null
python3 bench.py --mode forward --precision f16f8 --no-cpu-baseline > $O/bench_forward_f16f8.json 2>/dev/null
python3 bench.py --workload ship_refractive --no-extra --no-frame > $O/bench_ship_refractive.json 2>/dev/null
python3 bench.py --workload dolphin_train --no-extra --no-frame --no-cpu-baseline > $O/bench_dolphin_train.json 2>/dev/null
python3 bench.py --workload ship_refractive --stage all --no-extra --no-frame --no-cpu-baseline > $O/bench_stage_all.json 2>/dev/null
bash tools/r03/cmp_launch_modes.sh > $O/launch_modes.txt 2>&1
# rocprofv3 --kernel-trace --stats of the default command (the whole bench.py run, all legs)
(cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o default -- python3 $R/bench.py > $O/prof_default_bench.json 2> $O/prof_default.err)
cp $O/prof/default_kernel_stats.csv $O/default_bench_kernel_stats.csv
# one train step on the timeline + stats of a no-extras run
bash tools/r03/prof_step.sh final_step > $O/step_stats_and_timeline.txt 2>&1
cp $R/gpurun_out/r03/final_step_kernel_stats.csv $O/train_step_kernel_stats.csv; cp $R/gpurun_out/r03/final_step_timeline.txt $O/train_step_timeline.txt
bash tools/r03/prof_step.sh final_step_dolphin1024 --workload dolphin_train --rays 1024 > $O/step_dolphin1024.txt 2>&1
cp $R/gpurun_out/r03/final_step_dolphin1024_timeline.txt $O/train_step_dolphin1024_timeline.txt
# the stage-all* step: kernel stats of a no-extras run, and the march alone with its shell statistics
bash tools/r03/prof_step.sh final_stage_all --workload ship_refractive --stage all > $O/step_stage_all.txt 2>&1
cp $R/gpurun_out/r03/final_stage_all_kernel_stats.csv $O/kernel_stats_stage_all.csv
python3 tools/march_all_time.py ship_refractive 4096 > $O/march_all_time.txt 2>&1
# PMC passes (separate rocprofv3 runs per counter group)
bash tools/r03/pmc_all.sh ship_straight_f0_train_f32 > $O/pmc_train.txt 2>&1
bash tools/r03/pmc_all.sh ship_straight_f0_forward --mode forward > $O/pmc_forward.txt 2>&1
bash tools/r03/pmc_all.sh ship_refractive_f0_train_f32 --workload ship_refractive > $O/pmc_refr.txt 2>&1
bash tools/r03/pmc_all.sh dolphin_train_f128_train_f32 --workload dolphin_train > $O/pmc_dolphin.txt 2>&1
cp $R/gpurun_out/r03/pmc_*.json $O/
RNERF_WGRAD_TRACE=1 python3 tools/bwd_time.py > $O/bwd_time_trace.txt 2>&1
python3 tools/march_time.py > $O/march_time.txt 2>&1
rm -rf $O/prof $R/gpurun_out/r03/pmc $R/gpurun_out/r03/prof
ls $O
