# usage: bash tools/r03/ab_lib.sh <variant-name> [bench args]: product library vs samplenerfro_amd/lib/var/librnerf_<variant>.so, alternating, same box
for rep in 1 2; do
  unset RNERF_LIB; echo -n "product : "; bash tools/r03/ab.sh "${@:2}"
  export RNERF_LIB=$PWD/samplenerfro_amd/lib/var/librnerf_$1.so; echo -n "$1 : "; bash tools/r03/ab.sh "${@:2}"
done
