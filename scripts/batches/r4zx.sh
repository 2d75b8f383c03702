#!/bin/bash
# round 4, batch zx: generic wave reductions (module-level kernels) without LDS round trips: whole GPU suite, smoke, module-path timing
set -eu
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4zx; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -3 $O/tests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 400 python3 scripts/bench_module_path.py > $O/module_path.txt 2>&1 || { tail -20 $O/module_path.txt; exit 1; }
tail -8 $O/module_path.txt
