#!/bin/bash
# round 4, batch p: GEMM with unconditional prefetch / wait pairs (ISA lint): parity + timings at the config shapes; stress run
set -eu
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4p; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "gemm or golden or headline" > $O/gemm_tests.log 2>&1 || { tail -60 $O/gemm_tests.log; exit 1; }
tail -3 $O/gemm_tests.log
timeout -k 10 600 python scripts/gemm_narrow_ab.py > $O/gemm_ab.txt 2>&1 || { tail -40 $O/gemm_ab.txt; exit 1; }
sed -n 5,9p $O/gemm_ab.txt
timeout -k 10 600 python scripts/stress_parity.py > $O/stress.txt 2>&1 || { tail -20 $O/stress.txt; exit 1; }
tail -4 $O/stress.txt
