#!/bin/bash
# round 4, batch v: natural-scale 16-bit LayerNorm with row sums in registers - parity (Swin suite), then A/B
set -eu
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4v; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_swin.py -m gpu -x -q > $O/swin_tests.log 2>&1 || { tail -60 $O/swin_tests.log; exit 1; }
tail -3 $O/swin_tests.log
timeout -k 10 300 python scripts/time_swin_kernels.py lnc > $O/lnc.txt 2>&1 || { tail -30 $O/lnc.txt; exit 1; }
cat $O/lnc.txt
