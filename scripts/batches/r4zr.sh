#!/bin/bash
# round 4, batch zr: 200 forwards x 6 engines on the final binaries (any forward that differs from the first is a hazard)
set -eu
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r4zr
timeout -k 10 900 python3 scripts/stress_parity.py 200 > gpurun_out/r4zr/stress.txt 2>&1 || { tail -20 gpurun_out/r4zr/stress.txt; exit 1; }
cat gpurun_out/r4zr/stress.txt
