#!/bin/bash
# round 4, batch z: half-wave LayerNorm for small launches (fewer row pairs per wave, rows before the table, sqrt shortcut): parity + A/B + DeiT-S
set -eu
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4z; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "layernorm and not i16" > $O/ln_tests.log 2>&1 || { tail -60 $O/ln_tests.log; exit 1; }
tail -3 $O/ln_tests.log
timeout -k 10 300 python scripts/ln_ab.py --small > $O/ln_small.txt 2>&1 || { tail -30 $O/ln_small.txt; exit 1; }
cat $O/ln_small.txt
timeout -k 10 300 python scripts/bench_configs.py 1 2 > $O/configs.jsonl 2> $O/configs.err || { tail -20 $O/configs.err; exit 1; }
cat $O/configs.jsonl
