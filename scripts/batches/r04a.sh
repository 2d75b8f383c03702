#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04a; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_modules.py -m gpu -x -q -k "carries_int8 or ibert" > $O/t.log 2>&1 || { tail -60 $O/t.log; exit 1; }
tail -3 $O/t.log
