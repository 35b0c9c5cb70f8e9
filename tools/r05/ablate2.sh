export RNERF_LIB=$PWD/samplenerfro_amd/lib/var/librnerf_ablate.so
mkdir -p gpurun_out/r05
for d in 0 5 13 29 16 48; do RNERF_MLP_DEBUG=$d python3 tools/mlp_ablate.py f16 2>/dev/null; done > gpurun_out/r05/ablate2_f16.txt
