#!/bin/bash
# profiling build of the MLP kernels with the RNERF_MLP_DEBUG ablations / per-phase clocks compiled in -> samplenerfro_amd/lib/var/librnerf_ablate.so
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
mkdir -p $R/build/var $R/samplenerfro_amd/lib/var
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -Wno-unused-value -fno-slp-vectorize"
/opt/rocm/bin/hipcc $F -DRNERF_MLP_ABLATE -DRNERF_DGRAD_PROFILE -c $R/samplenerfro_amd/csrc/mlp.hip -o $R/build/var/mlp_ab.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $R/samplenerfro_amd/lib/grid.o $R/samplenerfro_amd/lib/march.o $R/samplenerfro_amd/lib/render.o $R/build/var/mlp_ab.o -o $R/samplenerfro_amd/lib/var/librnerf_ablate.so
echo $R/samplenerfro_amd/lib/var/librnerf_ablate.so
