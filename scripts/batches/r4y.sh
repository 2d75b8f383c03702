#!/bin/bash
# round 4, batch y: outer-order register sums of the natural-scale 16-bit LayerNorm - Swin parity suite, timing, which Shiftmax form each block takes
set -eu
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4y; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_swin.py -m gpu -x -q > $O/swin_tests.log 2>&1 || { tail -60 $O/swin_tests.log; exit 1; }
tail -3 $O/swin_tests.log
timeout -k 10 300 python scripts/time_swin_kernels.py lnc > $O/lnc.txt 2>&1 || { tail -30 $O/lnc.txt; exit 1; }
cat $O/lnc.txt
timeout -k 10 300 python scripts/bench_configs.py 5 15 > $O/configs.jsonl 2> $O/configs.err || { tail -20 $O/configs.err; exit 1; }
cat $O/configs.jsonl
