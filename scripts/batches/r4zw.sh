#!/bin/bash
# round 4, batch zq: final profiles of the bench command (stats + three counter passes) and the full bench line, on the final binaries
set -eu
cd "$GRAFT_REPO_ROOT"
R=$PWD
timeout -k 10 900 bash scripts/profile_bench.sh r04zj
cd $R
timeout -k 10 600 python3 bench.py > gpurun_out/r04zj_bench.log 2> gpurun_out/r04zj_bench_progress.txt || { tail -20 gpurun_out/r04zj_bench_progress.txt; exit 1; }
tail -1 gpurun_out/r04zj_bench.log
du -sh gpurun_out/prof_r04zj gpurun_out/pmc_r04zj_* | cat
