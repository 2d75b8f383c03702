#!/bin/bash
# round 4, batch zu: attention row reductions by permlane swaps instead of ds_bpermute: whole GPU suite, smoke, window attention timing, rocprofv3 of config 3
set -eu
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4zu; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -3 $O/tests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 300 python scripts/time_swin_kernels.py attn > $O/attn.txt 2>&1 || { tail -30 $O/attn.txt; exit 1; }
cat $O/attn.txt
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for c in 3 13; do
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4zu/prof$c -- python3 $R/scripts/bench_configs.py $c > $R/gpurun_out/r4zu/prof$c.log 2>&1 || { tail -5 $R/gpurun_out/r4zu/prof$c.log; exit 1; }
done
cd $R
python3 - <<'P'
import csv, glob
for c in (3, 13):
    rows = list(csv.DictReader(open(glob.glob(f'gpurun_out/r4zu/prof{c}/*/*kernel_stats.csv')[0])))
    print('config', c)
    for r in rows[:7]:
        print('  ', r['Name'].replace('(anonymous namespace)::', '')[:80].ljust(80), r['Calls'], round(float(r['AverageNs']) / 1e3, 1))
P
timeout -k 10 300 python3 scripts/bench_configs.py --graph 3 5 > $O/configs.jsonl 2> $O/configs.err || { tail -5 $O/configs.err; exit 1; }
cut -c1-200 $O/configs.jsonl
