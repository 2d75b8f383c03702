#!/bin/bash
# round 4, batch zf: the whole GPU suite + smoke
set -eu
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4zf; mkdir -p $O
timeout -k 10 1100 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -60 $O/gpu_tests.log; exit 1; }
tail -4 $O/gpu_tests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -2 $O/smoke.log
