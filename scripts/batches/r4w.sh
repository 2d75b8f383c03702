#!/bin/bash
# round 4, batch w: natural-scale window attention from the band table - Swin parity suite, then literal / band / power-of-two timing
set -eu
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4w; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_swin.py -m gpu -x -q > $O/swin_tests.log 2>&1 || { tail -60 $O/swin_tests.log; exit 1; }
tail -3 $O/swin_tests.log
timeout -k 10 300 python scripts/time_swin_kernels.py attnc > $O/attnc.txt 2>&1 || { tail -30 $O/attnc.txt; exit 1; }
cat $O/attnc.txt
