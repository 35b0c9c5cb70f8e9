#!/bin/bash
# usage (GPU box, repo root): bash tools/r06/prof_step.sh <tag> <marker kernel> [bench args...] -> gpurun_out/r06/<tag>_kernel_stats.csv, <tag>_timeline.txt
tag=$1; marker=$2; shift; shift
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r06; mkdir -p $O/prof
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o $tag -- python3 $R/bench.py --no-cpu-baseline --no-frame --no-extra --steps 10 "$@" > $O/$tag.json 2> $O/$tag.err
cd $R
cp $O/prof/${tag}_kernel_stats.csv $O/${tag}_kernel_stats.csv
python3 tools/r06/timeline.py $O/prof/${tag}_kernel_trace.csv "$marker" > $O/${tag}_timeline.txt
python3 tools/kstats.py $O/${tag}_kernel_stats.csv 20 > $O/${tag}_kstats.txt
rm -rf $O/prof
