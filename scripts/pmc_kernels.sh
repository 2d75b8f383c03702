#!/bin/bash
# usage: pmc_kernels.sh <tag> <script args...>   counter passes over scripts/time_kernels.py (attention / LN / GELU)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
T=$1; shift
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/$T/p1 -- python3 $R/scripts/time_kernels.py "$@" > $R/gpurun_out/$T.p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU --output-format csv -d $R/gpurun_out/$T/p2 -- python3 $R/scripts/time_kernels.py "$@" > $R/gpurun_out/$T.p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES --output-format csv -d $R/gpurun_out/$T/p3 -- python3 $R/scripts/time_kernels.py "$@" > $R/gpurun_out/$T.p3.log 2>&1
echo done
