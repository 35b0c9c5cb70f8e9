#!/bin/bash
# usage (GPU box): bash tools/r04/step_ab.sh <variant> [rays ...]: train step of the product library vs a variant, alternating on one box
V=$1; shift
for rep in 1 2 3; do
  for r in "${@:-4096}"; do
    W="--workload dolphin_train --rays $r"; [ "$r" = 4096 ] && W=""
    unset RNERF_LIB; echo -n "product rays $r: "; bash tools/r03/ab.sh $W
    export RNERF_LIB=$PWD/samplenerfro_amd/lib/var/librnerf_$V.so; echo -n "$V rays $r: "; bash tools/r03/ab.sh $W
  done
done
