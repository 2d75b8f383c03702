#!/bin/bash
# round 4, batch m: conversion-and-pack instruction semantics, more instruction prices, attention ablation baseline
set -eu
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4m; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/cvt_pack_probe scripts/probes/cvt_pack_probe.hip > $O/build.log 2>&1 || { tail -30 $O/build.log; exit 1; }
timeout -k 10 60 /tmp/cvt_pack_probe > $O/cvt_pack.txt 2>&1 || { tail -30 $O/cvt_pack.txt; exit 1; }
cat $O/cvt_pack.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rate_probe scripts/probes/valu_rate_probe.hip >> $O/build.log 2>&1 || { tail -30 $O/build.log; exit 1; }
timeout -k 10 120 /tmp/valu_rate_probe > $O/valu_rate.txt 2>&1 || { tail -30 $O/valu_rate.txt; exit 1; }
tail -7 $O/valu_rate.txt
timeout -k 10 300 python scripts/attn_ablate.py > $O/attn_ablate.txt 2>&1 || { tail -30 $O/attn_ablate.txt; exit 1; }
cat $O/attn_ablate.txt
