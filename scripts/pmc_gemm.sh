#!/bin/bash
# usage: pmc_gemm.sh <shape> <flags> <tag>   (run from repo root on the GPU box)
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
S=$1; F=$2; T=$3
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/$T/p1 -- python3 $R/scripts/gemm_one.py $S $F > $R/gpurun_out/$T.p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VALU_MFMA_COEXEC_CYCLES --output-format csv -d $R/gpurun_out/$T/p2 -- python3 $R/scripts/gemm_one.py $S $F > $R/gpurun_out/$T.p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM --output-format csv -d $R/gpurun_out/$T/p3 -- python3 $R/scripts/gemm_one.py $S $F > $R/gpurun_out/$T.p3.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCC_HIT_sum TCC_MISS_sum --output-format csv -d $R/gpurun_out/$T/p4 -- python3 $R/scripts/gemm_one.py $S $F > $R/gpurun_out/$T.p4.log 2>&1
echo done
