#!/bin/bash
# usage (on the GPU box, from the repo root): bash tools/prof_train.sh <tag> [bench args...]
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/c
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/c/prof -o $tag -- python3 $R/bench.py --no-cpu-baseline --no-frame --steps 10 "$@" > $R/gpurun_out/c/$tag.json 2> $R/gpurun_out/c/$tag.err
cd $R
cut -c1-260 gpurun_out/c/$tag.json
python3 tools/kstats.py gpurun_out/c/prof/${tag}_kernel_stats.csv 12
