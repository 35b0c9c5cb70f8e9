#!/usr/bin/env python3
"""Print the top rows of a rocprofv3 kernel_stats.csv: name, calls, average µs, total ms."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 14
for r in rows[:n]:
    print(f"{r['Name'][:58]:58s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  total {float(r['TotalDurationNs'])/1e6:8.2f} ms {r['Percentage']}%")
