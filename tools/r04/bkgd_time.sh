#!/bin/bash
# The two arithmetics of the background MLP side by side: time and difference (one box).   usage: bash tools/r04/bkgd_time.sh [rows ...]
set -u
mkdir -p gpurun_out/r04
OUT=gpurun_out/r04/bkgd_time.txt
: > $OUT
for n in "${@:-20480}"; do
  RNERF_BKGD_EXACT=1 python tools/r04/bkgd_time.py $n /tmp/bk_exact.npy >> $OUT 2>&1
  RNERF_BKGD_EXACT=0 python tools/r04/bkgd_time.py $n /tmp/bk_f16.npy >> $OUT 2>&1
  python - >> $OUT 2>&1 <<PY
import numpy as np
a = np.load("/tmp/bk_exact.npy", allow_pickle=True).item(); b = np.load("/tmp/bk_f16.npy", allow_pickle=True).item()
for k in ("rgb", "save", "grads"):
    d = np.abs(a[k] - b[k]); print(f"  rows $n  {k:5s} max |f16x3 - exact| = {d.max():.3e}   (max |exact| {np.abs(a[k]).max():.3e}, nonfinite {int((~np.isfinite(b[k])).sum())})")
PY
done
cat $OUT
