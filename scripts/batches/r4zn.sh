#!/bin/bash
# round 4, batch zn: ShiftGELU table pass with unconditional loads (row-major path) + prefetch for short rows: whole GPU suite, smoke, timing, configs
set -eu
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4zn; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -60 $O/gpu_tests.log; exit 1; }
tail -4 $O/gpu_tests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -2 $O/smoke.log
timeout -k 10 300 python scripts/time_swin_kernels.py gelu > $O/gelu.txt 2>&1 || { tail -30 $O/gelu.txt; exit 1; }
cat $O/gelu.txt
timeout -k 10 400 python3 scripts/bench_configs.py --graph 2 3 5 15 > $O/configs.jsonl 2> $O/configs.err || { tail -5 $O/configs.err; exit 1; }
cut -c1-200 $O/configs.jsonl
