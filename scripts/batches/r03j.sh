#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03j; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -60 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extras > $O/bench.json 2>> $O/bench.err; cat $O/bench.json
cd /tmp && timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof -o deitb -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --probe-forwards 0 > $GRAFT_REPO_ROOT/$O/prof_bench.json 2> $GRAFT_REPO_ROOT/$O/prof.err
cd $GRAFT_REPO_ROOT; ls $O/prof | head; f=$(ls $O/prof/*kernel_stats.csv | head -1); head -16 $f | cut -c1-200
