#!/bin/bash
# usage: bash tools/r03/build_variant.sh <name> <file.hip> <extra hipcc flags...>  -> samplenerfro_amd/lib/var/librnerf_<name>.so
# Rebuilds ONE source with extra -D switches (compile-time ablations / experiments) and links it with the product's other objects.
# Use with RNERF_LIB=samplenerfro_amd/lib/var/librnerf_<name>.so (tools/bwd_time.py, tools/march_time.py, bench.py).
set -e
name=$1; src=$2; shift 2
R=$(cd "$(dirname "$0")/../.." && pwd)
L=$R/samplenerfro_amd/lib; mkdir -p $L/var
python3 -m samplenerfro_amd.build > /dev/null
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-value -fno-slp-vectorize"
base=$(basename $src .hip)
/opt/rocm/bin/hipcc $FLAGS "$@" -c $R/samplenerfro_amd/csrc/$src -o $L/var/${base}_$name.o
objs=""
for o in grid march render mlp mlp_f32 bkgd16 pipeline; do if [ $o = $base ]; then objs="$objs $L/var/${base}_$name.o"; else objs="$objs $L/$o.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $L/var/librnerf_$name.so
echo $L/var/librnerf_$name.so
