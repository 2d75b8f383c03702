#!/bin/bash
# round 4, batch zy: final per-config table (configs 1 2 3 4 5 13 15, eager + graph) and the natural-scale bench under rocprofv3 --stats, final binaries
set -eu
cd "$GRAFT_REPO_ROOT"
R=$PWD
mkdir -p gpurun_out/r4zy
timeout -k 10 500 python3 scripts/bench_configs.py --graph 1 2 3 4 5 13 15 > gpurun_out/r4zy/configs.jsonl 2> gpurun_out/r4zy/configs.err || { tail -5 gpurun_out/r4zy/configs.err; exit 1; }
cut -c1-170 gpurun_out/r4zy/configs.jsonl
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4zy/prof_natural -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --natural-scales > $R/gpurun_out/r4zy/prof_natural.log 2>&1 || { tail -5 $R/gpurun_out/r4zy/prof_natural.log; exit 1; }
cd $R
grep '^{' gpurun_out/r4zy/prof_natural.log | cut -c1-300
