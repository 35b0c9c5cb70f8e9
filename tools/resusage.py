#!/usr/bin/env python3
"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` remarks: python tools/resusage.py remarks.txt [name-substring]"""
import re, sys
txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
KEYS = [("VGPR", r"    VGPRs"), ("AGPR", r"AGPRs"), ("spill", r"VGPRs Spill"), ("scratch", r"ScratchSize \[bytes/lane\]"),
        ("occ", r"Occupancy \[waves/SIMD\]"), ("SGPR", r"TotalSGPRs")]
for b in re.split(r'remark: Function Name: ', txt)[1:]:
    name = b.split()[0]
    if pat not in name:
        continue
    short = re.sub(r'^_ZN5rnerf\d+', '', name)[:44]
    vals = []
    for label, k in KEYS:
        m = re.search(k + r': (\d+)', b)
        vals.append(f"{label} {m.group(1) if m else '?':>4s}")
    print(f"{short:46s} " + "  ".join(vals))
