#!/bin/bash
# round 4, batch e: LDS pattern probe; streaming LayerNorm v2 (4 waves per SIMD, sqrt shortcut): parity, A/B, ablation
set -eu
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4e; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_pattern_probe scripts/probes/lds_pattern_probe.hip > $O/build.log 2>&1 || { tail -30 $O/build.log; exit 1; }
timeout -k 10 120 /tmp/lds_pattern_probe > $O/lds_pattern.txt 2>&1 || { tail -30 $O/lds_pattern.txt; exit 1; }
cat $O/lds_pattern.txt
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "layernorm or producers_write_block_layout" > $O/ln_tests.log 2>&1 || { tail -60 $O/ln_tests.log; exit 1; }
tail -3 $O/ln_tests.log
timeout -k 10 600 python scripts/ln_ab.py --headline > $O/ln_ab.txt 2>&1 || { tail -40 $O/ln_ab.txt; exit 1; }
cat $O/ln_ab.txt
timeout -k 10 300 python scripts/ln_ablate.py > $O/ln_ablate.txt 2>&1 || { tail -30 $O/ln_ablate.txt; exit 1; }
cat $O/ln_ablate.txt
