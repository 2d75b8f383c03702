#!/bin/bash
# round 4, batch l: A/B of narrow vs full work items of the weights-in-registers GEMM at the config shapes
set -eu
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4l; mkdir -p $O
timeout -k 10 600 python scripts/gemm_narrow_ab.py > $O/gemm_narrow_ab.txt 2>&1 || { tail -40 $O/gemm_narrow_ab.txt; exit 1; }
cat $O/gemm_narrow_ab.txt
