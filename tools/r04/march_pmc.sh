#!/bin/bash
# usage (GPU box): bash tools/r04/march_pmc.sh  -> gpurun_out/r04/march_pmc.txt: cache / TLB counters of march_kernel per launch, 64^3 vs 512^3, both layouts
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r04; mkdir -p $O/mpmc; cd /tmp; export TMPDIR=/tmp
: > $O/march_pmc.txt
for cfg in "64 reference" "512 reference" "512 bricks"; do
  set -- $cfg
  for ctr in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum"; do
    rm -rf $O/mpmc/run
    rocprofv3 --pmc $ctr --output-format csv -d $O/mpmc/run -o m -- python3 $R/tools/r04/march_one.py $1 $2 > /dev/null 2>&1
    f=$(find $O/mpmc/run -name "*counter_collection.csv" | head -1)
    python3 - "$f" "$1 $2" >> $O/march_pmc.txt <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "march_kernel" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in acc.items():
    print(f"G/layout {sys.argv[2]:14s} {k:40s} per launch {sum(v)/len(v):16.0f}  (launches {len(v)})")
PY
  done
done
rm -rf $O/mpmc
cat $O/march_pmc.txt
