#!/bin/bash
# usage (GPU box): bash tools/r04/check.sh <tag> -> what the driver runs at round end: build check, GPU test suite, smoke(), the default bench line
R=${GRAFT_REPO_ROOT:-$PWD}; T=${1:-check}; O=$R/gpurun_out/r04/$T; mkdir -p $O; cd $R
timeout 1800 python3 -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 $O/smoke.log
(time timeout 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err) 2> $O/bench_time.txt; echo "bench rc=$?"; cat $O/bench_time.txt | tail -3
python3 - <<PY
import json
d = json.load(open("$O/bench_default.json"))
print({k: d[k] for k in ("value", "ms_per_step", "dtype")}, "traffic", d["roofline"].get("traffic"), "counters", d["roofline"].get("counters"))
PY
