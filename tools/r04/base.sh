#!/bin/bash
# usage (GPU box, repo root): bash tools/r04/base.sh <tag>  -> gpurun_out/r04/<tag>/: GPU test suite + the default bench line on one box
R=${GRAFT_REPO_ROOT:-$PWD}
T=${1:-base}
O=$R/gpurun_out/r04/$T; mkdir -p $O
cd $R
timeout 1500 python3 -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "gpu tests rc=$?"; tail -3 $O/gpu_tests.log
timeout 600 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python3 - <<PY
import json
d = json.load(open("$O/bench_default.json"))
print({k: d[k] for k in ("value", "ms_per_step", "dtype") if k in d}, d.get("roofline"))
PY
