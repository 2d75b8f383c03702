#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03i; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o /tmp/epilogue_probe scripts/probes/epilogue_probe.hip
timeout -k 10 120 /tmp/epilogue_probe > $O/epilogue_probe.txt 2>&1; cat $O/epilogue_probe.txt
