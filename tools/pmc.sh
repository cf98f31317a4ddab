#!/bin/bash
# Collect PMC counters for tools/kbench.py in separate passes (gpurun refuses --pmc with other traces).
# usage: tools/pmc.sh <outdir> [kbench args]      (PMC_SCRIPT=path overrides tools/kbench.py)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; shift
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 ${PMC_SCRIPT:-$R/tools/kbench.py} "$@" > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; }
done
python3 $R/tools/pmc_summary.py $OUT
