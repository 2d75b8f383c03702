#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04b; mkdir -p $O
timeout -k 10 400 python scripts/bench_module_path.py 256 > $O/module_path_b256.txt 2>&1 || { tail -30 $O/module_path_b256.txt; exit 1; }
cat $O/module_path_b256.txt
