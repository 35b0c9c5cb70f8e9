#!/usr/bin/env python3
"""One train step on the GPU timeline from a rocprofv3 kernel_trace.csv: every kernel of the step between two rng_split3 kernels
(start offset, gap to the previous kernel's end, duration).  usage: timeline.py trace.csv [nth-from-last step, default 3]"""
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
key = "rng_forward"      # one per step in both launch forms (the key split runs on the host without the launch graph)
starts = [i for i, r in enumerate(rows) if key in r[2]]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 3
a, b = starts[-n - 1], starts[-n]
seg = rows[a:b]
t0 = seg[0][0]; end = t0; busy = 0
print(f"{len(seg)} kernels, step span {(rows[b][0]-t0)/1e3:.1f} us")
for s, e, name in seg:
    nm = name.replace("void rnerf::", "").split("(")[0][:64]
    print(f"  +{(s-t0)/1e3:8.1f} us  gap {(s-end)/1e3:7.1f}  dur {(e-s)/1e3:8.1f}  {nm}")
    busy += e - s
    end = max(end, e)
print(f"sum of durations {busy/1e3:.1f} us")
