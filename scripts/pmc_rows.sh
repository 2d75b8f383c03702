#!/bin/bash
# usage (GPU box, repo root): bash scripts/pmc_rows.sh <tag>   -- counter passes over the stand-alone row kernels
set -e
T=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
B="python3 $R/scripts/time_row_kernels.py"
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_I8"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $R/gpurun_out/pmc_rows_${T}_$i -- $B > $R/gpurun_out/pmc_rows_${T}_$i.log 2>&1 || echo "pass $i failed"
  echo "pass $i done"
done
