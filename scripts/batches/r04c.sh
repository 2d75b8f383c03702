#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04c; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -60 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
