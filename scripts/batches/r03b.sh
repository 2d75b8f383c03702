#!/bin/bash
# round 3, GPU batch b: parity of the 16x16x64 form of the weights-in-registers GEMM, then A/B against the 32x32x32 form
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03b; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "fragment_layout or both_kernels or output_map" > $O/gemm_tests.log 2>&1 || { tail -40 $O/gemm_tests.log; exit 1; }
tail -3 $O/gemm_tests.log
timeout -k 10 300 python scripts/gemm_ab.py --frags16 fc1 qkv proj fc2 -- 0 > $O/ab_s16_rq.txt 2>&1; cat $O/ab_s16_rq.txt
timeout -k 10 300 python scripts/gemm_ab.py --frags16 --resid proj fc2 -- 0 > $O/ab_s16_resid.txt 2>&1; cat $O/ab_s16_resid.txt
timeout -k 10 300 python scripts/gemm_ab.py --frags16 --qkv qkv -- 0 > $O/ab_s16_qkv.txt 2>&1; cat $O/ab_s16_qkv.txt
timeout -k 10 300 python scripts/debug/calib_parity.py > $O/calib_parity.txt 2>&1; cat $O/calib_parity.txt
