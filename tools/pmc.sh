#!/bin/bash
# usage (GPU box, repo root): bash tools/pmc.sh <tag> [bench args...]   -> gpurun_out/c/<tag>_pmc.json
# Separate passes per counter group (FETCH_SIZE and WRITE_SIZE do not fit one pass); --pmc only with --kernel-trace.
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out/c
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/c/pmc -o ${tag}_$c -- python3 $R/bench.py --no-cpu-baseline --no-frame --steps 5 --warmup 2 "$@" > /dev/null 2>&1
done
cd $R
python3 tools/pmc_summary.py gpurun_out/c/${tag}_pmc.json "rocprofv3 --kernel-trace --pmc <counter>, one counter per pass; bench.py --steps 5 --warmup 2 $*; FETCH_SIZE/WRITE_SIZE in KiB as reported" gpurun_out/c/pmc/${tag}_FETCH_SIZE_counter_collection.csv gpurun_out/c/pmc/${tag}_WRITE_SIZE_counter_collection.csv
