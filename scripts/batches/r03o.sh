#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03o; mkdir -p $O
bash scripts/profile_bench.sh r03o
PMC_CFGS="2 5" bash scripts/profile_configs.sh r03o 1 2 4 5
cd "$GRAFT_REPO_ROOT"
python3 scripts/summarize_profiles.py r03o > $O/summary.txt 2>&1; tail -20 $O/summary.txt
python3 scripts/summarize_configs.py r03o > $O/configs.txt 2>&1; tail -30 $O/configs.txt
