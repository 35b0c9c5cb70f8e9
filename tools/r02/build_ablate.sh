#!/bin/bash
# profiling build of the MLP kernels with the RNERF_MLP_DEBUG ablations / per-phase clocks compiled in -> samplenerfro_amd/lib/var/librnerf_ablate.so
set -e
R=$(cd "$(dirname "$0")/../.." && pwd)
bash $R/tools/r03/build_variant.sh ablate mlp.hip -DRNERF_MLP_ABLATE -DRNERF_DGRAD_PROFILE
