#!/bin/bash
# round 4, batch x: kernel stats of Swin-T b128 at power-of-two (5) and natural (15) scales, DeiT-S b64 (2), + throughput lines
set -eu
cd "$GRAFT_REPO_ROOT"
R=$PWD
O=$R/gpurun_out/r4x; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in 5 15 2; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg$c -- python3 $R/scripts/bench_configs.py $c > $O/prof_cfg$c.log 2>&1 || { tail -20 $O/prof_cfg$c.log; exit 1; }
  echo "config $c done"
done
cd $R
timeout -k 10 300 python3 scripts/bench_configs.py 2 5 15 > $O/configs.jsonl 2> $O/configs.err || { tail -20 $O/configs.err; exit 1; }
cat $O/configs.jsonl
find $O -name '*kernel_stats.csv' | head
