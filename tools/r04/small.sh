#!/bin/bash
# usage (GPU box): bash tools/r04/small.sh -> dolphin_train at 4096 / 1024 / 512 / 128 rays, RNERF_NO_AUX_STREAM=1 (levels in sequence) vs default (levels side by side)
for rep in 1 2; do
for r in 4096 1024 512 128; do
  echo -n "rays $r sequential : "; RNERF_NO_AUX_STREAM=1 bash tools/r03/ab.sh --workload dolphin_train --rays $r
  echo -n "rays $r side by side: "; bash tools/r03/ab.sh --workload dolphin_train --rays $r
done; done
