#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04d; mkdir -p $O
timeout -k 10 200 python scripts/attn_ablate.py 0 64 0 64 0 64 > $O/attn_onelut.txt 2>&1 || { tail -20 $O/attn_onelut.txt; exit 1; }
cat $O/attn_onelut.txt
