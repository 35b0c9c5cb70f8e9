#!/bin/bash
# usage (GPU box): bash tools/r04/bkgd_var.sh <variant> [rows]: background-MLP kernels of a variant library against the product's
V=$1; n=${2:-20480}
RNERF_BKGD_EXACT=1 python tools/r04/bkgd_time.py $n /tmp/bk_exact.npy
python tools/r04/bkgd_time.py $n /tmp/bk_prod.npy
RNERF_LIB=$PWD/samplenerfro_amd/lib/var/librnerf_$V.so python tools/r04/bkgd_time.py $n /tmp/bk_var.npy
python - <<PY
import numpy as np
a = np.load("/tmp/bk_exact.npy", allow_pickle=True).item()
for name in ("prod", "var"):
    b = np.load(f"/tmp/bk_{name}.npy", allow_pickle=True).item()
    for k in ("rgb", "save", "grads"):
        d = np.abs(a[k] - b[k]); print(f"  {name:5s} {k:5s} max |f16x3 - exact| = {np.nanmax(d):.3e}  nonfinite {int((~np.isfinite(b[k])).sum())}")
PY
