#!/bin/bash
# usage (GPU box): bash tools/pmc_sq.sh <kernel-substring> -- <python script args...>   SQ stall/busy counters per kernel
pat=$1; shift; shift
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sq && rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM --output-format csv -d /tmp/sq -o p -- python3 "$@" > /tmp/sq.log 2>&1
python3 - "$pat" <<'PY'
import csv, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
seen = set()
for r in csv.DictReader(open("/tmp/sq/p_counter_collection.csv")):
    if sys.argv[1] in r["Kernel_Name"]:
        k = r["Kernel_Name"].split("(")[0][-48:]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if (k, r["Dispatch_Id"]) not in seen: seen.add((k, r["Dispatch_Id"])); n[k] += 1
for k, cs in acc.items():
    print(k, "launches", n[k], {c: round(v / n[k] / 1e6, 1) for c, v in cs.items()})
PY
