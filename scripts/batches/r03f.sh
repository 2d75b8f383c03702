#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03f; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 300 python scripts/wreg_timeline.py proj fc2 --s16 --resid > $O/timeline_s16_resid.txt 2>&1; cat $O/timeline_s16_resid.txt
timeout -k 10 300 python scripts/wreg_timeline.py fc1 qkv --s16 > $O/timeline_s16_rq.txt 2>&1; cat $O/timeline_s16_rq.txt
