#!/bin/bash
# usage (on the GPU box, from the repo root): bash scripts/profile_bench.sh <tag>
# Four separate rocprofv3 runs of the default bench command (kernel stats; FETCH_SIZE; WRITE_SIZE; MFMA busy), as
# MI355X_MICROARCH.md prescribes (counters in their own passes, kernel-trace only).  Outputs under gpurun_out/;
# scripts/summarize_profiles.py <tag> condenses them into profiles/.
set -e
T=$1
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
B="python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras $BENCH_ARGS"   # e.g. BENCH_ARGS="--bitwidth 16 --operators ibert"
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$T -- $B > $R/gpurun_out/prof_$T.log 2>&1
echo "stats done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/pmc_${T}_fetch -- $B > $R/gpurun_out/pmc_${T}_fetch.log 2>&1
echo "fetch done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/pmc_${T}_write -- $B > $R/gpurun_out/pmc_${T}_write.log 2>&1
echo "write done"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/pmc_${T}_mfma -- $B > $R/gpurun_out/pmc_${T}_mfma.log 2>&1
echo "mfma done"
