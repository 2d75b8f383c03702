#!/bin/bash
# round 4, batch zi: float32 score requantisation in the Swin window attention (power-of-two multipliers) - parity, timing, config 5
set -eu
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4zi; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_swin.py -m gpu -x -q > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -3 $O/tests.log
timeout -k 10 300 python scripts/time_swin_kernels.py attn attnc > $O/attn.txt 2>&1 || { tail -30 $O/attn.txt; exit 1; }
cat $O/attn.txt
timeout -k 10 300 python3 scripts/bench_configs.py --graph 5 15 > $O/configs.jsonl 2> $O/configs.err || { tail -5 $O/configs.err; exit 1; }
cut -c1-200 $O/configs.jsonl
