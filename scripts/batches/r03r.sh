#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03r; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gelu_fused" > $O/gelu_tests.log 2>&1 || { tail -30 $O/gelu_tests.log; exit 1; }
tail -3 $O/gelu_tests.log
timeout -k 10 200 python scripts/gelu_fused_ab.py > $O/gelu_ab.txt 2>&1 || { tail -30 $O/gelu_ab.txt; exit 1; }
cat $O/gelu_ab.txt
