#!/usr/bin/env python3
"""One step on the GPU timeline from a rocprofv3 kernel_trace.csv: every kernel between two occurrences of a marker kernel (start offset,
gap to the end of everything before it, duration, stream/queue).  usage: timeline.py trace.csv [marker substring] [nth-from-last, default 3]"""
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "")) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
key = sys.argv[2] if len(sys.argv) > 2 else "rng_forward"
starts = [i for i, r in enumerate(rows) if key in r[2]]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 3
a, b = starts[-n - 1], starts[-n]
seg = rows[a:b]
t0 = seg[0][0]; end = t0; busy = 0
print(f"{len(seg)} kernels, step span {(rows[b][0]-t0)/1e3:.1f} us (marker: {key})")
for s, e, name, q in seg:
    nm = name.replace("void rnerf::", "").replace("void ", "").split("(")[0][:72]
    print(f"  +{(s-t0)/1e3:8.1f} us  gap {(s-end)/1e3:7.1f}  dur {(e-s)/1e3:8.1f}  q{q:>3} {nm}")
    busy += e - s
    end = max(end, e)
print(f"sum of durations {busy/1e3:.1f} us")
