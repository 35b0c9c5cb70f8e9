#!/bin/bash
# usage: bash tools/r04/build_head.sh [rev] -> samplenerfro_amd/lib/var/librnerf_head.so: the library of a committed revision (default HEAD),
# for A/B runs of the working tree against it on one box (RNERF_LIB=...; tools/r04/step_ab.sh).
set -e
rev=${1:-HEAD}
R=$(cd "$(dirname "$0")/../.." && pwd)
W=/tmp/rnerf_head_src; rm -rf $W; mkdir -p $W $R/samplenerfro_amd/lib/var
(cd $R && git archive $rev samplenerfro_amd/csrc include) | tar -x -C $W
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-value -fno-slp-vectorize -I$W/include"
objs=""
for f in $W/samplenerfro_amd/csrc/*.hip; do
  o=$W/$(basename $f .hip).o
  /opt/rocm/bin/hipcc $FLAGS -c $f -o $o &
  objs="$objs $o"
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $R/samplenerfro_amd/lib/var/librnerf_head.so
echo $R/samplenerfro_amd/lib/var/librnerf_head.so
