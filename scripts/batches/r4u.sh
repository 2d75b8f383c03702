#!/bin/bash
# round 4, batch u: accumulators preloaded by LDS reads instead of moves - GEMM parity, then the A/B (flags2 16384 = former form)
set -eu
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4u; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "gemm" > $O/gemm_tests.log 2>&1 || { tail -60 $O/gemm_tests.log; exit 1; }
tail -3 $O/gemm_tests.log
timeout -k 10 400 python scripts/gemm_ab.py --frags16 --resid proj fc2 -- 0:0 0:16384 > $O/ab_resid.txt 2>&1 || { tail -20 $O/ab_resid.txt; exit 1; }
grep frags16 $O/ab_resid.txt
timeout -k 10 400 python scripts/gemm_ab.py --frags16 fc1 qkv -- 0:0 0:16384 > $O/ab_rq.txt 2>&1 || { tail -20 $O/ab_rq.txt; exit 1; }
grep frags16 $O/ab_rq.txt
