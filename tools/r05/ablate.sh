#!/bin/bash
# usage (GPU box): bash tools/r05/ablate.sh [precision] -> where a slab's time goes in the forward engine (profiling build, tools/r03/build_variant.sh ablate mlp.hip -DRNERF_MLP_ABLATE -DRNERF_EXPERIMENTS)
export RNERF_LIB=$PWD/samplenerfro_amd/lib/var/librnerf_ablate.so
P=${1:-f16}
for d in 0 64 256 1 2 4 8 16; do RNERF_MLP_DEBUG=$d python3 tools/mlp_ablate.py $P 2>/dev/null; done
