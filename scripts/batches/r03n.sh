#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03n; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_modules.py -m gpu -x -q -k "int8 or carries or module_path" > $O/lazy_tests.log 2>&1 || { tail -80 $O/lazy_tests.log; exit 1; }
tail -3 $O/lazy_tests.log
IVIT_KERNEL_SPLIT=1 timeout -k 10 300 python scripts/bench_module_path.py 256 --family ivit > $O/module_path_b256.txt 2>&1 || { tail -30 $O/module_path_b256.txt; exit 1; }
cat $O/module_path_b256.txt
