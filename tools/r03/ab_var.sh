# usage: bash tools/r03/ab_var.sh <variant> [rows]: product library vs samplenerfro_amd/lib/var/librnerf_<variant>.so, alternating on ONE box:
# the NerfMLP kernels alone (tools/bwd_time.py).  `old` = a copy of the previous product build kept by hand.
v=$1; shift
for rep in 1 2; do
  unset RNERF_LIB; echo "== product"; python tools/bwd_time.py "$@" 2>&1 | tail -4
  export RNERF_LIB=$PWD/samplenerfro_amd/lib/var/librnerf_$v.so; echo "== $v"; python tools/bwd_time.py "$@" 2>&1 | tail -4
done
