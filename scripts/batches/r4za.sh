#!/bin/bash
# round 4, batch za: kernel durations (rocprofv3) of the half-wave LayerNorm variants at the DeiT-S shape
set -eu
cd "$GRAFT_REPO_ROOT"
R=$PWD
O=$R/gpurun_out/r4za; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/scripts/ln_ab.py --small --first > $O/prof.log 2>&1 || { tail -20 $O/prof.log; exit 1; }
python3 - <<PY
import csv,glob
f=glob.glob('$O/prof/**/*kernel_stats.csv',recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:12]:
    print(f"{r['Name'][:110]:110s} {r['Calls']:>6} {float(r['AverageNs'])/1e3:7.2f} us")
PY
