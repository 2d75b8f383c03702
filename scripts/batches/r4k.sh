#!/bin/bash
# round 4, batch k: narrow (128 x 128) work items of the weights-in-registers GEMM: parity (all GEMM tests, engines), configs timing
set -eu
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4k; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "gemm" > $O/gemm_tests.log 2>&1 || { tail -60 $O/gemm_tests.log; exit 1; }
tail -3 $O/gemm_tests.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q -k "golden or config or headline or batch_invariance or swin" > $O/model_tests.log 2>&1 || { tail -60 $O/model_tests.log; exit 1; }
tail -3 $O/model_tests.log
timeout -k 10 600 python scripts/bench_configs.py > $O/configs.jsonl 2> $O/configs.err || { tail -20 $O/configs.err; exit 1; }
cat $O/configs.jsonl
