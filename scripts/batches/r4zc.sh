#!/bin/bash
# round 4, batch zc: the wave-pipelined GEMM as a lab form (flags2 bit 15) - parity, then timing with its ablations
set -eu
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4zc; mkdir -p $O
timeout -k 10 400 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gemm" > $O/gemm_tests.log 2>&1 || { tail -40 $O/gemm_tests.log; exit 1; }
tail -3 $O/gemm_tests.log
timeout -k 10 300 python scripts/gemm_ab.py --frags16 fc1 qkv -- 0:0 0:32768 0:98304 0:163840 0:229376 > $O/ab_wp.txt 2>&1 || { tail -20 $O/ab_wp.txt; exit 1; }
grep frags16 $O/ab_wp.txt
