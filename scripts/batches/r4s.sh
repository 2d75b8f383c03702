#!/bin/bash
# round 4, batch s: progress priority as the default - LayerNorm parity (all forms) then the A/B
set -eu
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4s; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "layernorm or producers_write_block_layout" > $O/ln_tests.log 2>&1 || { tail -60 $O/ln_tests.log; exit 1; }
tail -3 $O/ln_tests.log
timeout -k 10 300 python scripts/ln_ab.py --headline > $O/ln_ab.txt 2>&1 || { tail -40 $O/ln_ab.txt; exit 1; }
cat $O/ln_ab.txt
timeout -k 10 200 python bench.py --steps 20 --warmup 5 > $O/bench.log 2>&1 || { tail -40 $O/bench.log; exit 1; }
tail -1 $O/bench.log
