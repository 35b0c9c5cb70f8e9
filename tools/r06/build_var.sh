#!/bin/bash
# usage: bash tools/r06/build_var.sh <name> <extra hipcc flags...>  -> samplenerfro_amd/lib/var/librnerf_<name>.so
# csrc/mlp.hip rebuilt with extra -D switches (compile-time experiments), linked with the product's other objects (which must be built).
# Load with samplenerfro_amd._lib.load(path) first thing in the process (tools/r06/ab_wgrad.py <path>).
set -e
name=$1; shift
R=$(cd "$(dirname "$0")/../.." && pwd)
L=$R/samplenerfro_amd/lib; mkdir -p $L/var
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-value -fno-slp-vectorize"
/opt/rocm/bin/hipcc $FLAGS "$@" -c $R/samplenerfro_amd/csrc/mlp.hip -o $L/var/mlp_$name.o
objs=""
for o in grid march render mlp mlp_f32 bkgd16 pipeline; do if [ $o = mlp ]; then objs="$objs $L/var/mlp_$name.o"; else objs="$objs $L/$o.o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $L/var/librnerf_$name.so
echo $L/var/librnerf_$name.so
