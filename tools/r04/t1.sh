#!/bin/bash
# usage (GPU box): bash tools/r04/t1.sh  -> new tests of this round + the full default bench line
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04/t1; mkdir -p $O; cd $R
timeout 900 python3 -m pytest tests/test_gpu_checkpoint.py tests/test_gpu_bench_world2.py "tests/test_gpu_parity.py::test_f16x3_against_the_exact_fp32_arbiter" "tests/test_gpu_parity.py::test_nerf_mlp" tests/test_gpu_train.py tests/test_gpu_whole_path.py -x -q -s > $O/tests.log 2>&1; echo "tests rc=$?"; tail -30 $O/tests.log
timeout 900 python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -3 $O/bench_default.err
python3 - <<PY
import json
d = json.load(open("$O/bench_default.json"))
print({k: d[k] for k in ("value", "ms_per_step", "dtype")}, d.get("stability"), d["frame"])
print({k: (v.get("ms_per_step") if isinstance(v, dict) else v) for k, v in d["variants"].items()})
PY
