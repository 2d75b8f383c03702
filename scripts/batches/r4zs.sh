#!/bin/bash
# round 4, batch zs: skinny-K GEMM with K / 32 at compile time: GEMM + Swin parity, rocprofv3 of config 5, configs 5 / 15
set -eu
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4zs; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_swin.py -m gpu -x -q -k "gemm or swin or golden or engine" > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -3 $O/tests.log
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4zs/prof5 -- python3 $R/scripts/bench_configs.py 5 > $R/gpurun_out/r4zs/prof5.log 2>&1 || { tail -5 $R/gpurun_out/r4zs/prof5.log; exit 1; }
cd $R
python3 - <<'P'
import csv, glob
rows = list(csv.DictReader(open(glob.glob('gpurun_out/r4zs/prof5/*/*kernel_stats.csv')[0])))
for r in rows:
    if 'skinny' in r['Name'] or 'gemm_i8' in r['Name']:
        print(r['Name'].replace('(anonymous namespace)::', '')[:80].ljust(80), r['Calls'], round(float(r['AverageNs']) / 1e3, 1))
P
timeout -k 10 300 python3 scripts/bench_configs.py --graph 5 15 > $O/configs.jsonl 2> $O/configs.err || { tail -5 $O/configs.err; exit 1; }
cut -c1-200 $O/configs.jsonl
