# usage: bash tools/r03/ab_env.sh VAR=VALUE [bench args]: default environment vs VAR=VALUE, alternating, same box
kv=$1; shift
for rep in 1 2; do
  echo -n "default : "; bash tools/r03/ab.sh "$@"
  echo -n "$kv : "; env $kv bash tools/r03/ab.sh "$@"
done
