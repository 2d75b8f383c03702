#!/bin/bash
# round 4, batch zo: ShiftGELU short rows without the prefetch as the product form: whole GPU suite, smoke, timing, rocprofv3 of configs 5 and 2
set -eu
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4zo; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -3 $O/tests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 300 python scripts/time_swin_kernels.py gelu pn > $O/gelu.txt 2>&1 || { tail -30 $O/gelu.txt; exit 1; }
cat $O/gelu.txt
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4zo/prof5 -- python3 $R/scripts/bench_configs.py 5 > $R/gpurun_out/r4zo/prof5.log 2>&1 || { tail -5 $R/gpurun_out/r4zo/prof5.log; exit 1; }
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4zo/prof2 -- python3 $R/scripts/bench_configs.py 2 > $R/gpurun_out/r4zo/prof2.log 2>&1 || { tail -5 $R/gpurun_out/r4zo/prof2.log; exit 1; }
cd $R
find gpurun_out/r4zo -name "*kernel_stats.csv" | head
