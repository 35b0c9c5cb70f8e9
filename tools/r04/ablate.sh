#!/bin/bash
# usage (GPU box): bash tools/r04/ablate.sh -> where a slab's time goes in the forward engine (profiling build)
export RNERF_LIB=$PWD/samplenerfro_amd/lib/var/librnerf_ablate.so
for d in 0 64 256 1 2 4 8 16; do RNERF_MLP_DEBUG=$d python3 tools/mlp_ablate.py f16x3 2>/dev/null; done
