#!/bin/bash
# round 4, batch o: whole GPU suite (new goldens: vit_large, swin_small, both regimes), bench line
set -eu
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4o; mkdir -p $O
timeout -k 10 1150 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -60 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
timeout -k 10 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }
cat $O/bench.json
