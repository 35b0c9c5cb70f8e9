# the committed bench lines of profiles/r06 (run AFTER the PMC passes are in profiles/r06: traffic / counters are looked up there)
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out/r06/lines; mkdir -p $O; cd $R
python3 bench.py > $O/bench_default.json 2>/dev/null; echo "default rc=$?"
python3 bench.py --mode forward > $O/bench_forward.json 2>/dev/null
python3 bench.py --precision f16 --backward f16 --no-frame > $O/bench_train_f16.json 2>/dev/null
python3 bench.py --mode forward --precision f16 --no-extra > $O/bench_forward_f16.json 2>/dev/null
python3 bench.py --mode forward --precision bf16 --no-extra > $O/bench_forward_bf16.json 2>/dev/null
