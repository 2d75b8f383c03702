#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03l; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python scripts/gemm_ab.py --frags16 --resid proj fc2 -- 0:0 2048:0 > $O/ab_split.txt 2>&1; grep frags16 $O/ab_split.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -60 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
