#!/bin/bash
# usage (GPU box, repo root): bash scripts/profile_configs.sh <tag> [configs...]
# rocprofv3 kernel stats of each BASELINE.json config that fits one GPU (scripts/bench_configs.py: 1 DeiT-T b1, 2 DeiT-S b64,
# 3 DeiT-B b256, 4 ViT-B b128 = one rank's shard of config 4, 5 Swin-T b128), eager and (small batches) HIP-graph replay.
# Outputs under gpurun_out/; scripts/summarize_configs.py <tag> condenses them into profiles/.
set -e
T=$1; shift
CFGS=${@:-1 2 4 5}
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp; export TMPDIR=/tmp
for c in $CFGS; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${T}_cfg$c -- python3 $R/scripts/bench_configs.py $c > $R/gpurun_out/prof_${T}_cfg$c.log 2>&1
  echo "config $c stats done"
done
# separate counter passes (FETCH_SIZE, WRITE_SIZE: one counter per run, kernel-trace only) for the configs in $PMC_CFGS
for c in $PMC_CFGS; do
  for k in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $k --output-format csv -d $R/gpurun_out/pmc_${T}_cfg${c}_$k -- python3 $R/scripts/bench_configs.py $c > $R/gpurun_out/pmc_${T}_cfg${c}_$k.log 2>&1
    echo "config $c $k done"
  done
done
python3 $R/scripts/bench_configs.py --graph 1 2 3 4 5 > $R/gpurun_out/${T}_configs.jsonl 2> $R/gpurun_out/${T}_configs.err
echo "throughput lines done"
