#!/bin/bash
# usage (GPU box): bash tools/r04/ab_fwd.sh <variant> [precision ...]: the NerfMLP forward kernel alone, product library vs variant, alternating on one box
V=$1; shift
for rep in 1 2 3; do
  for p in "${@:-f16x3}"; do
    unset RNERF_LIB; echo -n "product  "; python3 tools/mlp_ablate.py $p
    export RNERF_LIB=$PWD/samplenerfro_amd/lib/var/librnerf_$V.so; echo -n "$V  "; python3 tools/mlp_ablate.py $p
  done
done
