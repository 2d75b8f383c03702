#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04k; mkdir -p $O
timeout -k 10 400 python scripts/gemm_ab.py --frags16 --resid proj fc2 -- 0:0 0:2048 > $O/ab_prio.txt 2>&1; grep frags16 $O/ab_prio.txt
