#!/usr/bin/env python3
"""Collapse rocprofv3 --pmc counter_collection CSVs into {kernel: {counter: {launches, mean}}} JSON.
usage: pmc_summary.py out.json note a_counter_collection.csv [b_counter_collection.csv ...]"""
import collections, csv, json, re, sys
out, note, files = sys.argv[1], sys.argv[2], sys.argv[3:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    # one row per (dispatch, counter[, dimension instance]): sum the instances of a dispatch, then average over dispatches
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        k = re.sub(r"^void ", "", r["Kernel_Name"]).split("(")[0]
        per[(k, r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    for (k, c, _), v in per.items():
        acc[k][c].append(v)
res = {"note": note, "counters": {k: {c: {"launches": len(v), "mean": sum(v) / len(v)} for c, v in cs.items()} for k, cs in acc.items() if "rnerf" in k}}
json.dump(res, open(out, "w"), indent=1)
for k, cs in res["counters"].items():
    print(k[:60], {c: round(v["mean"], 1) for c, v in cs.items()})
