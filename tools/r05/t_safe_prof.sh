#!/bin/bash
# the range-safe training step (bf16x3 forward + bf16 backward) as a bench command of its own + its rocprofv3 stats / timeline
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05
python3 bench.py --precision bf16x3 --backward bf16 --no-extra --no-frame --no-cpu-baseline > gpurun_out/r05/bench_train_range_safe.json 2> gpurun_out/r05/bench_train_range_safe.err; echo "rc=$?"
bash tools/r05/prof_step.sh range_safe_step rng_forward --precision bf16x3 --backward bf16 > gpurun_out/r05/range_safe_prof.log 2>&1; echo "prof rc=$?"
ls gpurun_out/r05 | grep range_safe
