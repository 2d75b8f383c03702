#!/bin/bash
# round 4, batch zl: skinny-K GEMM (Swin stage 0) - parity (GEMM + Swin suites), config 5 kernel stats
set -eu
cd "$GRAFT_REPO_ROOT"
R=$PWD
O=$R/gpurun_out/r4zl; mkdir -p $O
timeout -k 10 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "skinny" > $O/sk_tests.log 2>&1 || { tail -40 $O/sk_tests.log; exit 1; }
tail -2 $O/sk_tests.log
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_swin.py -m gpu -x -q -k "gemm or swin" > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -3 $O/tests.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg5 -- python3 $R/scripts/bench_configs.py 5 > $O/prof_cfg5.log 2>&1 || { tail -20 $O/prof_cfg5.log; exit 1; }
cd $R
python3 - <<PY
import csv,glob
f=glob.glob('$O/prof_cfg5/*/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('config 5 kernel sum per forward ms',round(tot/1e6/25,3))
for r in rows[:18]:
    print(f"  {r['Name'][:100]:100s} {int(r['Calls'])/25:5.1f} {float(r['AverageNs'])/1e3:7.1f} us")
PY
timeout -k 10 300 python3 scripts/bench_configs.py --graph 5 15 > $O/configs.jsonl 2> $O/configs.err || { tail -5 $O/configs.err; exit 1; }
cut -c1-200 $O/configs.jsonl
