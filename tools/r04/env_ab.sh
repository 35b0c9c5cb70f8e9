#!/bin/bash
# usage (GPU box): bash tools/r04/env_ab.sh <ENV_NAME> "<values>" [rays ...]: train step under each value of an environment switch, alternating
E=$1; VALS=$2; shift 2
for rep in 1 2 3; do
  for r in "${@:-512}"; do
    W="--workload dolphin_train --rays $r"; [ "$r" = 4096 ] && W=""
    for v in $VALS; do echo -n "rays $r $E=$v: "; env $E=$v bash tools/r03/ab.sh $W; done
  done
done
