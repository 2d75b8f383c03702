#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04h; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_modules.py -m gpu -x -q -k "reference_attention or drives_the_modules or hip_graph" > $O/t.log 2>&1 || { tail -40 $O/t.log; exit 1; }
tail -3 $O/t.log
