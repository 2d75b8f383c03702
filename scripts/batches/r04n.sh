#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04n; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_model.py -m gpu -x -q -k "repeated" > $O/t.log 2>&1 || { tail -30 $O/t.log; exit 1; }
tail -2 $O/t.log
