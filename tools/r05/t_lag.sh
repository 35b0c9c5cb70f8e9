#!/bin/bash
cd $GRAFT_REPO_ROOT
for i in 1 2; do
python3 bench.py --no-frame --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); L=d['precision_legs']['range_safe_train']
print('default', round(d['ms_per_step'],3), 'sync switch', round(L['default_step_with_range_retry_on']['ms_per_step'],3), 'lag', round(L['default_step_with_range_retry_lag']['ms_per_step'],3), L['default_step_with_range_retry_lag']['re_runs'])
"
done
