#!/bin/bash
# round 4, batch c: extended VALU price list + ablation of the streaming LayerNorm
set -eu
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4c; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate_probe scripts/probes/valu_rate_probe.hip > $O/build.log 2>&1 || { tail -30 $O/build.log; exit 1; }
timeout -k 10 120 /tmp/valu_rate_probe > $O/valu_rate.txt 2>&1 || { tail -30 $O/valu_rate.txt; exit 1; }
cat $O/valu_rate.txt
timeout -k 10 300 python scripts/ln_ablate.py > $O/ln_ablate.txt 2>&1 || { tail -30 $O/ln_ablate.txt; exit 1; }
cat $O/ln_ablate.txt
