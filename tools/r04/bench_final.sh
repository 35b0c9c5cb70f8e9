#!/bin/bash
# usage (GPU box): bash tools/r04/bench_final.sh -> gpurun_out/r04/bench_final.json, bench_forward_final.json: the two headline lines with the PMC-derived fields filled
mkdir -p gpurun_out/r04
python3 bench.py > gpurun_out/r04/bench_final.json 2>/dev/null
python3 bench.py --mode forward > gpurun_out/r04/bench_forward_final.json 2>/dev/null
python3 - <<'PY'
import json
for f in ("bench_final", "bench_forward_final"):
    d = json.load(open("gpurun_out/r04/" + f + ".json"))
    print(f, round(d["value"]), round(d["ms_per_step"], 3), d["roofline"]["kernel"], d["roofline"].get("traffic"), d["roofline"].get("counters"))
PY
