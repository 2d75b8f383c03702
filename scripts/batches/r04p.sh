#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r04p_cfg15 -- python3 $R/scripts/bench_configs.py 15 > $R/gpurun_out/prof_r04p_cfg15.log 2>&1
echo done
