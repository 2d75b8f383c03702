#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03s; mkdir -p $O
cd scripts/probes && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/xcc_map_probe xcc_map_probe.hip && cd ../..
timeout -k 10 60 /tmp/xcc_map_probe 512 > $O/xcc_map.txt 2>&1; cat $O/xcc_map.txt
