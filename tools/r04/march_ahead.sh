#!/bin/bash
# usage (GPU box): bash tools/r04/march_ahead.sh  -> march_kernel at 64^3 / 256^3 / 512^3, both table layouts, speculation depth 2 (product) / 3 / 4 / 6
for v in product ahead3 ahead4 ahead6; do
  if [ $v = product ]; then unset RNERF_LIB; else export RNERF_LIB=$PWD/samplenerfro_amd/lib/var/librnerf_$v.so; fi
  echo "== $v"; python3 tools/march_time.py 2>/dev/null | grep -v refractive
done
