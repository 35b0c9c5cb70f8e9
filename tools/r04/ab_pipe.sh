#!/bin/bash
# usage (GPU box): bash tools/r04/ab_pipe.sh <variant>: product vs variant library — the three big kernels alone (tools/bwd_time.py) + eval forward, alternating
V=${1:-oldpipe}
for rep in 1 2; do
  unset RNERF_LIB; echo "== product"; python3 tools/mlp_ablate.py f16x3 2>/dev/null; python3 tools/mlp_ablate.py f16f8 2>/dev/null; python3 tools/bwd_time.py 2>/dev/null | tail -4
  export RNERF_LIB=$PWD/samplenerfro_amd/lib/var/librnerf_$V.so; echo "== $V"; python3 tools/mlp_ablate.py f16x3 2>/dev/null; python3 tools/mlp_ablate.py f16f8 2>/dev/null; python3 tools/bwd_time.py 2>/dev/null | tail -4
done
