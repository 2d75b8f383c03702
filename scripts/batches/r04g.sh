#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04g; mkdir -p $O
timeout -k 10 400 python scripts/gemm_ab.py --frags16 --resid proj fc2 -- 0:0 262144:0 524288:0 1048576:0 1572864:0 > $O/ab_stagger_resid.txt 2>&1; grep frags16 $O/ab_stagger_resid.txt
timeout -k 10 400 python scripts/gemm_ab.py --frags16 fc1 qkv -- 0:0 262144:0 524288:0 1048576:0 > $O/ab_stagger.txt 2>&1; grep frags16 $O/ab_stagger.txt
