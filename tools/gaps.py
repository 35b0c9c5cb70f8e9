#!/usr/bin/env python3
"""Idle time between kernels from a rocprofv3 kernel_trace.csv: python tools/gaps.py trace.csv [first_kernel_substring]"""
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
key = sys.argv[2] if len(sys.argv) > 2 else "march_kernel"
starts = [i for i, r in enumerate(rows) if key in r[2]]
# take the last full step: between the last two occurrences of the key kernel
a, b = starts[-3], starts[-2]
seg = rows[a:b]
t0 = seg[0][0]; busy = 0; end = seg[0][0]
print(f"{len(seg)} kernels, step span {(rows[b][0]-t0)/1e3:.1f} us")
for s, e, n in seg:
    gap = s - end
    busy += e - s
    if gap > 3000 or e - s > 50000:
        print(f"  +{(s-t0)/1e3:8.1f} us  gap {gap/1e3:7.1f}  dur {(e-s)/1e3:8.1f}  {n[:70]}")
    end = max(end, e)
print(f"busy {busy/1e3:.1f} us, idle {(rows[b][0]-t0-busy)/1e3:.1f} us")
