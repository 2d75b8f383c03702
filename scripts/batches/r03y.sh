#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03y; mkdir -p $O
timeout -k 10 300 python scripts/attn_occ_insitu.py > $O/attn_occ_insitu.txt 2>&1 || { tail -20 $O/attn_occ_insitu.txt; exit 1; }
cat $O/attn_occ_insitu.txt
