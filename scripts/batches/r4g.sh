#!/bin/bash
# round 4, batch g: wave timeline of the streaming LayerNorm
set -eu
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4g; mkdir -p $O
timeout -k 10 300 python scripts/ln_timeline.py > $O/ln_timeline.txt 2>&1 || { tail -30 $O/ln_timeline.txt; exit 1; }
cat $O/ln_timeline.txt
