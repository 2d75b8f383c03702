#!/bin/bash
# round 4, batch n: attention micro-optimisations: parity (attention + model tests), ablation with A/B of the score requantisation
set -eu
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4n; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "attention or golden or headline or geometries" > $O/attn_tests.log 2>&1 || { tail -60 $O/attn_tests.log; exit 1; }
tail -3 $O/attn_tests.log
timeout -k 10 300 python scripts/attn_ablate.py > $O/attn_ablate.txt 2>&1 || { tail -30 $O/attn_ablate.txt; exit 1; }
cat $O/attn_ablate.txt
