#!/bin/bash
# Collect PMC counters in separate rocprofv3 passes (gpurun refuses --pmc combined with other traces; FETCH_SIZE and
# WRITE_SIZE do not fit one pass) and summarise them per kernel.
# usage: tools/pmc.sh <outdir under gpurun_out/> [script args]
#   PMC_SCRIPT=tools/xyz.py  the program to profile (default tools/kbench.py: whole-flow kernel + per-layer chain)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/$1; shift
SCRIPT=${PMC_SCRIPT:-tools/kbench.py}
case "$SCRIPT" in /*) ;; *) SCRIPT=$R/$SCRIPT ;; esac
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_VALU_MFMA_COEXEC_CYCLES SQ_IFETCH" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" ; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 $SCRIPT "$@" > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -5 $OUT/p$i.log; }
done
python3 $R/tools/pmc_summary.py $OUT
