#!/bin/bash
# round 4, batch zj: outer-order tie rows of the 8-bit natural-scale LayerNorm in parallel - parity (compat + swin), config 15 kernel stats
set -eu
cd "$GRAFT_REPO_ROOT"
R=$PWD
O=$R/gpurun_out/r4zj; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_compat.py tests/test_gpu_swin.py -m gpu -x -q > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -3 $O/tests.log
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_cfg15 -- python3 $R/scripts/bench_configs.py 15 > $O/prof_cfg15.log 2>&1 || { tail -20 $O/prof_cfg15.log; exit 1; }
cd $R
python3 - <<PY
import csv,glob
f=glob.glob('$O/prof_cfg15/*/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('config 15 kernel sum per forward ms',round(tot/1e6/25,3))
for r in rows[:14]:
    print(f"  {r['Name'][:100]:100s} {int(r['Calls'])/25:5.1f} {float(r['AverageNs'])/1e3:7.1f} us")
PY
