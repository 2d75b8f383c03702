#!/bin/bash
# round 4, batch t: MFMA / VALU co-issue probe (two waves of one SIMD, as the GEMM's main loop beside the other workgroup's epilogue)
set -eu
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4t; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -o /tmp/coissue_probe scripts/probes/coissue_probe.hip
timeout -k 10 300 /tmp/coissue_probe > $O/coissue.txt 2>&1 || { tail -40 $O/coissue.txt; exit 1; }
cat $O/coissue.txt
