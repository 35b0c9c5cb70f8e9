#!/bin/bash
# usage (GPU box): bash tools/r04/variant_probe.sh: the small-batch `variants` of the default bench line under a few switches — why they differed from
# a direct run of the same workload: HIP streams share a few hardware queues (GPU_MAX_HW_QUEUES, default 4), and a process that has created
# many streams (the bench builds several models) can land two streams of one model on one queue
X="--no-frame --no-cpu-baseline --steps 3 --warmup 1"
for e in "A=1" "RNERF_AUX2_STREAM=1" "GPU_MAX_HW_QUEUES=8" "A=2"; do
  env $e python bench.py $X 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); v=d['variants']; print('$e', round(d['ms_per_step'],3), {k: round(v[k]['ms_per_step'],3) for k in v if k.startswith('dolphin')})"
done
