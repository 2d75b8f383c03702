#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_WAVES SQ_INSTS_VALU_MFMA_I8 --output-format csv -d $R/gpurun_out/pmc_r04l_valu -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-extras > $R/gpurun_out/pmc_r04l_valu.log 2>&1
echo done
