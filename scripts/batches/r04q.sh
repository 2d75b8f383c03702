#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04q; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "layernorm or compat or swin or natural" > $O/t.log 2>&1 || { tail -40 $O/t.log; exit 1; }
tail -2 $O/t.log
timeout -k 10 300 python scripts/bench_configs.py 15 5 > $O/cfg15.jsonl 2>&1; cat $O/cfg15.jsonl
