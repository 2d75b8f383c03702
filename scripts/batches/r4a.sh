#!/bin/bash
# round 4, batch a: parity of the streaming LayerNorm kernel (every LN test, both libraries) + A/B timing of its variants
set -eu
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4a; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "layernorm or producers_write_block_layout" > $O/ln_tests.log 2>&1 || { tail -60 $O/ln_tests.log; exit 1; }
tail -3 $O/ln_tests.log
timeout -k 10 600 python scripts/ln_ab.py > $O/ln_ab.txt 2>&1 || { tail -40 $O/ln_ab.txt; exit 1; }
cat $O/ln_ab.txt
