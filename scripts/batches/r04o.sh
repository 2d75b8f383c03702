#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04o; mkdir -p $O
timeout -k 10 400 python scripts/bench_configs.py 5 15 3 13 > $O/natural_vs_pow2.jsonl 2>&1; cat $O/natural_vs_pow2.jsonl
