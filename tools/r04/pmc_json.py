#!/usr/bin/env python3
"""(round 3: stamps every csrc file and reads the commit from the tracked VERSION file when .git is absent)
Collapse rocprofv3 --pmc passes into {kernel: {counter: mean per launch, "avg_ns": kernel-trace mean}} JSON, stamped with the repo
state.  usage: pmc_json.py out.json note *_counter_collection.csv *_kernel_trace.csv"""
import collections, csv, hashlib, json, os, re, subprocess, sys
out, note, files = sys.argv[1], sys.argv[2], sys.argv[3:]
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CSRC = os.path.join(ROOT, "samplenerfro_amd", "csrc")
acc = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(list)
name_of = lambda n: re.sub(r"^void ", "", n).split("(")[0]
for f in files:
    if f.endswith("_kernel_trace.csv"):
        for r in csv.DictReader(open(f)):
            dur[name_of(r["Kernel_Name"])].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
        continue
    per = collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        per[(name_of(r["Kernel_Name"]), r["Counter_Name"], r["Dispatch_Id"])] += float(r["Counter_Value"])
    for (k, c, _), v in per.items():
        acc[k][c].append(v)
sha = lambda p: hashlib.sha256(open(p, "rb").read()).hexdigest()[:16]
try:
    head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
except Exception:
    head = ""
version = open(os.path.join(ROOT, "VERSION")).read().strip() if os.path.exists(os.path.join(ROOT, "VERSION")) else ""
res = {"note": note, "bench_py_sha16": sha(os.path.join(ROOT, "bench.py")),
       "csrc_sha16": {f: sha(os.path.join(CSRC, f)) for f in sorted(os.listdir(CSRC)) if f.endswith((".hip", ".inc", ".h"))},
       "head": head or ("VERSION: " + version if version else "(snapshot without .git and without VERSION)"),
       "xcd_instances": 8,
       "units": "GRBM_GUI_ACTIVE / GRBM_COUNT are summed over the 8 XCDs' GRBM instances (divide by 8 for cycles); FETCH_SIZE / WRITE_SIZE in KiB as reported by rocprofv3 (gfx950: wide 16 B/lane reads are counted at half their size); SQ_* as reported "
                "(SQ_WAVE_CYCLES, SQ_WAIT_*, SQ_ACTIVE_INST_* in quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES in cycles summed over SIMDs); avg_ns from --kernel-trace of the same passes",
       "counters": {}}
for k in sorted(set(acc) | set(dur)):
    if "rnerf" not in k:
        continue
    ent = {c: {"launches": len(v), "mean": sum(v) / len(v)} for c, v in acc[k].items()}
    if dur[k]:
        ent["avg_ns"] = {"launches": len(dur[k]), "mean": sum(dur[k]) / len(dur[k])}
    res["counters"][k] = ent
json.dump(res, open(out, "w"), indent=1)
for k, cs in res["counters"].items():
    if "avg_ns" in cs and cs["avg_ns"]["mean"] > 50000:
        busy = cs.get("SQ_VALU_MFMA_BUSY_CYCLES", {}).get("mean"); gui = cs.get("GRBM_GUI_ACTIVE", {}).get("mean")
        extra = ""
        if busy and gui:        # GRBM_GUI_ACTIVE is reported summed over the 8 XCDs' GRBM instances
            extra = f" mfma_busy/(1024 SIMD x GRBM_GUI_ACTIVE/8) = {busy / (1024 * gui / 8):.3f}, eff. clock {gui / 8 / cs['avg_ns']['mean']:.2f} GHz"
        print(k[-60:], round(cs["avg_ns"]["mean"] / 1e3, 1), "us", extra)
