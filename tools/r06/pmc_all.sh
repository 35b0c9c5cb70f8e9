#!/bin/bash
# usage (GPU box, repo root): bash tools/r06/pmc_all.sh <tag> [bench args...]  -> gpurun_out/r06/pmc_<tag>.json
# One rocprofv3 --pmc pass per counter group (FETCH_SIZE and WRITE_SIZE do not fit one pass; --pmc only with --kernel-trace), the program
# itself after `--` (python3 bench.py, no wrapper).  <tag> = <workload>_f<fine>_<forward|train_<backward>>[_p<precision>] is what bench.py looks up.
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r06; mkdir -p $O/pmc
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F16" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc -o ${tag}_g$i -- python3 $R/bench.py --no-cpu-baseline --no-frame --no-extra --steps 5 --warmup 2 "$@" > $O/pmc/${tag}_g$i.log 2>&1 || echo "group $i failed: $grp"
done
cd $R
python3 tools/r04/pmc_json.py $O/pmc_${tag}.json "$tag: bench.py --no-cpu-baseline --no-frame --no-extra --steps 5 --warmup 2 $*" $O/pmc/${tag}_g*_counter_collection.csv $O/pmc/${tag}_g*_kernel_trace.csv
