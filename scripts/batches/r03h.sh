#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03h; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python scripts/wreg_timeline.py proj fc2 --s16 --resid > $O/timeline_s16_resid.txt 2>&1; grep -v "workgroup \|amdgpu" $O/timeline_s16_resid.txt
timeout -k 10 300 python scripts/gemm_ab.py --frags16 --resid proj fc2 -- 0 > $O/ab_resid.txt 2>&1; cat $O/ab_resid.txt
