#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03k; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "gemm" > $O/gemm_tests.log 2>&1 || { tail -40 $O/gemm_tests.log; exit 1; }
tail -3 $O/gemm_tests.log
timeout -k 10 600 python -m pytest tests/test_gpu_model.py tests/test_gpu_compat.py -m gpu -x -q > $O/model_tests.log 2>&1 || { tail -40 $O/model_tests.log; exit 1; }
tail -3 $O/model_tests.log
timeout -k 10 300 python scripts/wreg_timeline.py proj fc2 --s16 --resid > $O/timeline_s16_resid.txt 2>&1; grep -v "workgroup \|amdgpu" $O/timeline_s16_resid.txt
timeout -k 10 300 python scripts/wreg_timeline.py fc1 --s16 > $O/timeline_s16_rq.txt 2>&1; grep -v "workgroup \|amdgpu" $O/timeline_s16_rq.txt
timeout -k 10 300 python scripts/gemm_ab.py --frags16 --resid proj fc2 -- 0:0 0:512 > $O/ab_resid.txt 2>&1; cat $O/ab_resid.txt
timeout -k 10 300 python scripts/gemm_ab.py --frags16 fc1 qkv -- 0:0 0:512 > $O/ab_rq.txt 2>&1; cat $O/ab_rq.txt
timeout -k 10 300 python scripts/gemm_ab.py --frags16 --qkv qkv -- 0:0 0:512 > $O/ab_qkv.txt 2>&1; cat $O/ab_qkv.txt
timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-extras > $O/bench.json 2>> $O/bench.err; cat $O/bench.json
