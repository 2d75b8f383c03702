#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04t; mkdir -p $O
timeout -k 10 300 python scripts/bench_configs.py 13 13 > $O/natural_occ4.jsonl 2>&1; cat $O/natural_occ4.jsonl
sed -i 's/G == 8 \&\& NJ <= 3 ? 4 : NJ <= 1 ? 4/G == 8 \&\& NJ <= 3 ? (COMPAT ? 3 : 4) : NJ <= 1 ? 4/' i-vit_amd/csrc/rowops.hip
grep -c "COMPAT ? 3 : 4" i-vit_amd/csrc/rowops.hip
timeout -k 10 600 make -C i-vit_amd/csrc -s -j8 > $O/make.log 2>&1 || { tail -20 $O/make.log; exit 1; }
timeout -k 10 300 python scripts/bench_configs.py 13 13 > $O/natural_occ3.jsonl 2>&1; cat $O/natural_occ3.jsonl
timeout -k 10 300 python -m pytest tests/test_gpu_compat.py -m gpu -x -q -k "natural_scale_model" > $O/t.log 2>&1; tail -1 $O/t.log
