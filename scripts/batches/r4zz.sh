#!/bin/bash
# round 4, batch zz: patchify with 32-bit index arithmetic: whole GPU suite, smoke, rocprofv3 of config 3 (patchify duration)
set -eu
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4zz; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -3 $O/tests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4zz/prof3 -- python3 $R/scripts/bench_configs.py 3 > $R/gpurun_out/r4zz/prof3.log 2>&1 || { tail -5 $R/gpurun_out/r4zz/prof3.log; exit 1; }
cd $R
grep -h "patchify\|embed_kernel" gpurun_out/r4zz/prof3/*/*kernel_stats.csv | cut -c1-200
