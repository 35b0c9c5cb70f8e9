#!/bin/bash
# usage (GPU box, repo root): bash tools/r02/pmc_all.sh <tag> [bench args...]  -> gpurun_out/r02/pmc_<tag>.json
# One rocprofv3 --pmc pass per counter group (FETCH_SIZE and WRITE_SIZE do not fit one pass; --pmc only with --kernel-trace).
tag=$1; shift
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r02; mkdir -p $O/pmc
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "(SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*|GRBM_GUI_ACTIVE|SQ_BUSY_CYCLES|SQ_BUSY_CU_CYCLES)" | sort -u > $O/pmc/counters_available.txt
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_MFMA_MOPS_F16" "GRBM_GUI_ACTIVE GRBM_COUNT" "SQ_INSTS_VALU_MFMA_F16 SQ_INSTS_VALU_MFMA_BF16 SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc -o ${tag}_g$i -- python3 $R/bench.py --no-cpu-baseline --no-frame --no-extra --steps 5 --warmup 2 "$@" > $O/pmc/${tag}_g$i.log 2>&1 || echo "group $i failed: $grp"
done
cd $R
python3 tools/r02/pmc_json.py $O/pmc_${tag}.json "$tag: bench.py --steps 5 --warmup 2 $*" $O/pmc/${tag}_g*_counter_collection.csv $O/pmc/${tag}_g*_kernel_trace.csv
