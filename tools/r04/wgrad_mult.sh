#!/bin/bash
# usage (GPU box): bash tools/r04/wgrad_mult.sh [rays ...]: rounds of one-per-CU wgrad workgroups (RNERF_WGRAD_MULT; default 2) at small batches
for rep in 1 2; do
  for r in "${@:-512}"; do
    W="--workload dolphin_train --rays $r"; [ "$r" = 4096 ] && W=""
    for m in 2 1 3; do echo -n "rays $r mult $m: "; RNERF_WGRAD_MULT=$m bash tools/r03/ab.sh $W; done
  done
done
