#!/bin/bash
# round 4, batch zv: LDS-free butterflies in the 16-bit LayerNorm and the ShiftGELU row maximum: whole GPU suite, smoke, LN16 / GELU timing, rocprofv3 of configs 5 and 2
set -eu
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4zv; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -3 $O/tests.log
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1 || { tail -20 $O/smoke.log; exit 1; }
tail -1 $O/smoke.log
timeout -k 10 300 python scripts/time_swin_kernels.py ln gelu > $O/ln_gelu.txt 2>&1 || { tail -30 $O/ln_gelu.txt; exit 1; }
cat $O/ln_gelu.txt
R=$GRAFT_REPO_ROOT
cd /tmp; export TMPDIR=/tmp
for c in 5 2; do
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r4zv/prof$c -- python3 $R/scripts/bench_configs.py $c > $R/gpurun_out/r4zv/prof$c.log 2>&1 || { tail -5 $R/gpurun_out/r4zv/prof$c.log; exit 1; }
done
cd $R
python3 - <<'P'
import csv, glob
for c in (5, 2):
    rows = list(csv.DictReader(open(glob.glob(f'gpurun_out/r4zv/prof{c}/*/*kernel_stats.csv')[0])))
    print('config', c, 'kernel time per forward', round(sum(float(r['TotalDurationNs']) for r in rows) / 25 / 1e3, 1), 'us')
    for r in rows:
        if 'layernorm' in r['Name'] or 'gelu' in r['Name'] or 'attention' in r['Name']:
            print('  ', r['Name'].replace('(anonymous namespace)::', '')[:80].ljust(80), r['Calls'], round(float(r['AverageNs']) / 1e3, 1))
P
timeout -k 10 300 python3 scripts/bench_configs.py --graph 2 5 15 > $O/configs.jsonl 2> $O/configs.err || { tail -5 $O/configs.err; exit 1; }
cut -c1-200 $O/configs.jsonl
