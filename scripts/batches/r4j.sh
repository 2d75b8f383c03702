#!/bin/bash
# round 4, batch j: LayerNorm parity with every kernel form forced (incl. the Newton-shortcut rows), plain and COMPAT
set -eu
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4j; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q -k "layernorm or producers_write_block_layout" > $O/ln_tests.log 2>&1 || { tail -60 $O/ln_tests.log; exit 1; }
tail -3 $O/ln_tests.log
