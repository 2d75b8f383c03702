#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03q; mkdir -p $O
timeout -k 10 300 python scripts/debug/gelu_fused_diag.py > $O/diag.txt 2>&1 || { tail -30 $O/diag.txt; exit 1; }
head -40 $O/diag.txt
