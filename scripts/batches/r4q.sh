#!/bin/bash
# round 4, batch q: rocprofv3 kernel stats + FETCH / WRITE / MFMA-busy passes of the bench command (-> profiles/r04h_*), natural-scale
# kernel stats, kernel stats of configs 2, 4, 5
set -eu
cd "$GRAFT_REPO_ROOT"
bash scripts/profile_bench.sh r04h
BENCH_ARGS="--natural-only" bash -c 'true'
R=$GRAFT_REPO_ROOT
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r04h_natural -- python3 $R/scripts/bench_configs.py 13 > $R/gpurun_out/prof_r04h_natural.log 2>&1 ) || { tail -20 gpurun_out/prof_r04h_natural.log; exit 1; }
echo "natural stats done"
PMC_CFGS="" bash scripts/profile_configs.sh r04h 2 4 5
