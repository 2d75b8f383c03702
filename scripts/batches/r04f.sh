#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04f; mkdir -p $O
timeout -k 10 300 python scripts/wreg_timeline.py proj fc2 --s16 --resid > $O/timeline.txt 2>&1 || { tail -30 $O/timeline.txt; exit 1; }
grep -v "^workgroup [0-9]" $O/timeline.txt
