#!/bin/bash
# round 4, batch b: VALU instruction price list (scripts/probes/valu_rate_probe.hip)
set -eu
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4b; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate_probe scripts/probes/valu_rate_probe.hip > $O/build.log 2>&1 || { tail -30 $O/build.log; exit 1; }
timeout -k 10 120 /tmp/valu_rate_probe > $O/valu_rate.txt 2>&1 || { tail -30 $O/valu_rate.txt; exit 1; }
cat $O/valu_rate.txt
