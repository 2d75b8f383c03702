#!/bin/bash
# round 4, batch ze: final profiles of the round - bench command (stats + counter passes), natural-scale bench, configs 2 4 5 15, full bench line
set -eu
cd "$GRAFT_REPO_ROOT"
R=$PWD
[ -d gpurun_out/prof_r04q ] || timeout -k 10 900 bash scripts/profile_bench.sh r04q
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r04q_natural -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --natural-scales > $R/gpurun_out/prof_r04q_natural.log 2>&1 || { tail -5 $R/gpurun_out/prof_r04q_natural.log; exit 1; }
cd $R
for c in 2 4 5 15; do
  ( cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_r04q_cfg$c -- python3 $R/scripts/bench_configs.py $c > $R/gpurun_out/prof_r04q_cfg$c.log 2>&1 ) || { tail -5 gpurun_out/prof_r04q_cfg$c.log; exit 1; }
  echo "config $c done"
done
timeout -k 10 400 python3 scripts/bench_configs.py --graph 1 2 3 4 5 13 15 > gpurun_out/r04q_configs.jsonl 2> gpurun_out/r04q_configs.err || { tail -5 gpurun_out/r04q_configs.err; exit 1; }
cat gpurun_out/r04q_configs.jsonl
timeout -k 10 600 python3 bench.py > gpurun_out/r04q_bench.log 2>&1 || { tail -20 gpurun_out/r04q_bench.log; exit 1; }
tail -1 gpurun_out/r04q_bench.log
