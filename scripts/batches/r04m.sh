#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04m; mkdir -p $O
timeout -k 10 500 python scripts/stress_parity.py 400 > $O/stress.txt 2>&1 || { tail -20 $O/stress.txt; exit 1; }
cat $O/stress.txt
