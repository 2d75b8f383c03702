#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03m; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_modules.py -m gpu -x -q -k "int8 or carries or module_path" > $O/lazy_tests.log 2>&1 || { tail -80 $O/lazy_tests.log; exit 1; }
tail -3 $O/lazy_tests.log
timeout -k 10 300 python scripts/bench_module_path.py 256 --family ivit > $O/module_path_b256.txt 2>&1 || { tail -30 $O/module_path_b256.txt; exit 1; }
cat $O/module_path_b256.txt
timeout -k 10 600 python -m pytest tests/test_gpu_swin.py -m gpu -x -q > $O/swin_tests.log 2>&1 || { tail -60 $O/swin_tests.log; exit 1; }
tail -3 $O/swin_tests.log
timeout -k 10 200 python scripts/bench_configs.py 5 > $O/cfg5.json 2>&1 && cat $O/cfg5.json
IVIT_FRAGS16=0 timeout -k 10 200 python scripts/bench_configs.py 5 > $O/cfg5_f32.json 2>&1 && cat $O/cfg5_f32.json
