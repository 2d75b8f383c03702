#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03u; mkdir -p $O
timeout -k 10 200 python scripts/dual_stream_probe.py 2 > $O/dual2.txt 2>&1 || { tail -20 $O/dual2.txt; exit 1; }
cat $O/dual2.txt
timeout -k 10 200 python scripts/dual_stream_probe.py 2 >> $O/dual2b.txt 2>&1; cat $O/dual2b.txt
