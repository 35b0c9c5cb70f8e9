#!/bin/bash
# forward pass: the next batch's march beside the MLP on reserved CUs (--pipeline --reserve-cus n) against the strict sequence (--no-pipeline)
cd $GRAFT_REPO_ROOT
show='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(sys.argv[1], round(d["value"]), round(d["ms_per_step"],4))'
for P in "" "--precision f16"; do
A="--mode forward --no-frame --no-cpu-baseline --no-extra --steps 60 --warmup 10 $P"
for rep in 1 2; do
python3 bench.py $A --no-pipeline 2>/dev/null | python3 -c "$show" "$P sequence"
for n in 32 40 48 64 96; do
python3 bench.py $A --pipeline --reserve-cus $n 2>/dev/null | python3 -c "$show" "$P pipeline reserve $n"
done
done
done
