#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03t; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -60 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -2 $O/smoke.log
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
cat $O/bench.json
