#!/bin/bash
# usage (GPU box, repo root): bash scripts/pmc_one.sh <tag> <kernel-name-substring> -- python3 script.py args...
# Kernel stats + SQ counter passes of one command (counters in their own runs, kernel-trace only: MI355X_MICROARCH.md);
# prints per-kernel averages of every counter for kernels whose name contains the substring.
set -eu
T=$1; K=$2; shift 3
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/pmc_$T; mkdir -p $O
cd /tmp; export TMPDIR=/tmp
run() { # name, counters...
  local n=$1; shift
  rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/$n -- "${CMD[@]}" > $O/$n.log 2>&1 || { tail -20 $O/$n.log; exit 1; }
}
CMD=("$@")
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- "${CMD[@]}" > $O/stats.log 2>&1 || { tail -20 $O/stats.log; exit 1; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_LDS SQ_ACTIVE_INST_LDS
run sq2 SQ_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SALU
run grbm GRBM_GUI_ACTIVE
python3 $R/scripts/pmc_one_summary.py $O "$K"
