#!/bin/bash
# sustained train-step rate: 20 steps (the default), then 5000 steps (~31 s) of the same command, then 20 again
cd $GRAFT_REPO_ROOT
show='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]), round(d["ms_per_step"],4), "peak GiB", d.get("memory",{}).get("peak_allocated_gib"))'
A="--no-frame --no-cpu-baseline --no-extra"
python3 bench.py $A 2>/dev/null | python3 -c "$show" "20 steps"
python3 bench.py $A --steps 5000 --warmup 50 2>/dev/null | python3 -c "$show" "5000 steps"
python3 bench.py $A 2>/dev/null | python3 -c "$show" "20 steps"
rocm-smi --showtemp --showpower 2>/dev/null | grep -i "junction\|Average Graphics\|socket" | head -4
