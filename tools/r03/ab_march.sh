# usage: bash tools/r03/ab_march.sh [variant ...]: tools/march_time.py with the product library and with samplenerfro_amd/lib/var/librnerf_<variant>.so, on ONE box
unset RNERF_LIB; echo "== product"; python tools/march_time.py 2>&1 | tail -6
for v in "$@"; do
  export RNERF_LIB=$PWD/samplenerfro_amd/lib/var/librnerf_$v.so; echo "== $v"; python tools/march_time.py 2>&1 | tail -6
done
