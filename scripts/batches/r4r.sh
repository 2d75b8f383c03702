#!/bin/bash
# round 4, batch r: LayerNorm A/B with progress-based wave priority
set -eu
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4r; mkdir -p $O
timeout -k 10 600 python scripts/ln_ab.py --headline > $O/ln_ab.txt 2>&1 || { tail -40 $O/ln_ab.txt; exit 1; }
cat $O/ln_ab.txt
