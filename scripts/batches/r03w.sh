#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03w; mkdir -p $O
timeout -k 10 200 python scripts/attn_ablate.py 0 32 0 32 > $O/attn_occ.txt 2>&1 || { tail -20 $O/attn_occ.txt; exit 1; }
cat $O/attn_occ.txt
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "attention or golden or natural or headline or ibert or batch" > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -2 $O/tests.log
for i in 1 2; do timeout -k 10 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras > $O/bench_$i.json 2> $O/bench.err; python -c "
import json;d=json.load(open('$O/bench_$i.json'));print(d['ms_per_step'],d['value'])"; done
timeout -k 10 300 python scripts/bench_configs.py 1 2 3 4 5 13 > $O/configs.jsonl 2>&1; cat $O/configs.jsonl
