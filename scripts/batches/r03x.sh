#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03x; mkdir -p $O
bash scripts/profile_bench.sh r03x
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r03x_natural -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --natural-scales > $R/gpurun_out/prof_r03x_natural.log 2>&1
echo "natural stats done"
