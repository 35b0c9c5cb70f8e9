#!/bin/bash
# usage (GPU box): bash tools/r04/bkgd_prof.sh [rows]: kernel durations of the background-MLP kernels (rocprofv3 kernel trace), both arithmetics
n=${1:-20480}
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/r04
for mode in 1 0; do
  export RNERF_BKGD_EXACT=$mode
  rm -rf /tmp/bkprof; (cd /tmp && rocprofv3 --kernel-trace --stats -d /tmp/bkprof -o bk --output-format csv -- python3 $R/tools/r04/bkgd_time.py $n > /dev/null 2>&1)
  f=$(find /tmp/bkprof -name '*kernel_stats.csv' | head -1)
  echo "RNERF_BKGD_EXACT=$mode rows $n"; python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "bkgd" in r["Name"]:
        print(f"  {r['Name'][:70]:70s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:8.1f} us  min {float(r['MinNs'])/1e3:8.1f}")
PY
done | tee $R/gpurun_out/r04/bkgd_prof_$n.txt
