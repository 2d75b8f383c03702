#!/bin/bash
# round 4, batch zg: all-half-tiles launches of the weights-in-registers GEMM (<= 256 tiles) - parity (GEMM, models), kernel durations in configs 2 and 5
set -eu
cd "$GRAFT_REPO_ROOT"
R=$PWD
O=$R/gpurun_out/r4zg; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "gemm or model or swin or modules" > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -3 $O/tests.log
cd /tmp && export TMPDIR=/tmp
for c in 2 5; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg$c -- python3 $R/scripts/bench_configs.py $c > $O/prof_cfg$c.log 2>&1 || { tail -20 $O/prof_cfg$c.log; exit 1; }
done
cd $R
python3 - <<PY
import csv,glob
for c in (2,5):
    f=glob.glob('$O/prof_cfg%d/*/*kernel_stats.csv'%c)[0]
    rows=list(csv.DictReader(open(f)))
    tot=sum(float(r['TotalDurationNs']) for r in rows)
    print('config',c,'kernel sum per forward ms',round(tot/1e6/25,3))
    for r in rows[:9]:
        print(f"  {r['Name'][:100]:100s} {int(r['Calls'])/25:5.1f} {float(r['AverageNs'])/1e3:7.1f} us")
PY
timeout -k 10 300 python3 scripts/bench_configs.py --graph 2 5 > $O/configs.jsonl 2> $O/configs.err || { tail -5 $O/configs.err; exit 1; }
cat $O/configs.jsonl
