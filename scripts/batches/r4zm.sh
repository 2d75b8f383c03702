#!/bin/bash
# round 4, batch zm: one-dword half-wave int8 LayerNorm for rows of at most 128 channels (Swin's patch norm) - parity, timing, config 5
set -eu
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r4zm; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_compat.py -m gpu -x -q -k "layernorm" > $O/tests_ln.log 2>&1 || { tail -60 $O/tests_ln.log; exit 1; }
tail -3 $O/tests_ln.log
timeout -k 10 900 python -m pytest tests/test_gpu_swin.py -m gpu -x -q > $O/tests.log 2>&1 || { tail -60 $O/tests.log; exit 1; }
tail -3 $O/tests.log
timeout -k 10 300 python scripts/time_swin_kernels.py pn > $O/pn.txt 2>&1 || { tail -30 $O/pn.txt; exit 1; }
cat $O/pn.txt
timeout -k 10 300 python3 scripts/bench_configs.py --graph 2 5 15 > $O/configs.jsonl 2> $O/configs.err || { tail -5 $O/configs.err; exit 1; }
cut -c1-200 $O/configs.jsonl
