mkdir -p gpurun_out/r05
bash tools/r05/ablate.sh f16 > gpurun_out/r05/ablate_f16.txt 2>&1
bash tools/r05/ablate.sh f16x3 > gpurun_out/r05/ablate_f16x3.txt 2>&1
