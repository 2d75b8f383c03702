#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04e; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gemm" > $O/gemm_tests.log 2>&1 || { tail -30 $O/gemm_tests.log; exit 1; }
tail -2 $O/gemm_tests.log
timeout -k 10 300 python scripts/gemm_ab.py --frags16 --resid proj fc2 -- 0:0 0:1024 > $O/ab_resload.txt 2>&1; grep frags16 $O/ab_resload.txt
