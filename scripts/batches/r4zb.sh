#!/bin/bash
# round 4, batch zb: wave-pipelined GEMM (gemm_wp.h) - parity against the two-workgroup form and the oracle, then A/B timing
set -eu
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4zb; mkdir -p $O
timeout -k 10 240 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "wave_pipelined" > $O/wp_tests.log 2>&1 || { tail -40 $O/wp_tests.log; exit 1; }
tail -3 $O/wp_tests.log
timeout -k 10 400 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gemm" > $O/gemm_tests.log 2>&1 || { tail -40 $O/gemm_tests.log; exit 1; }
tail -3 $O/gemm_tests.log
timeout -k 10 300 python scripts/gemm_ab.py --frags16 fc1 qkv -- 0:0 0:32768 > $O/ab_rq.txt 2>&1 || { tail -20 $O/ab_rq.txt; exit 1; }
grep frags16 $O/ab_rq.txt
