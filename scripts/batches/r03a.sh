#!/bin/bash
# round 3, GPU batch a: baseline tests, MFMA-shape probe, cache-policy A/B of the GEMM epilogue, baseline bench line
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03a; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -30 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_shape_probe scripts/probes/mfma_shape_probe.hip
timeout -k 10 120 /tmp/mfma_shape_probe random 2 > $O/mfma_shape_random_2wg.txt 2>&1
timeout -k 10 120 /tmp/mfma_shape_probe random 1 > $O/mfma_shape_random_1wg.txt 2>&1
timeout -k 10 120 /tmp/mfma_shape_probe zero 2 > $O/mfma_shape_zero_2wg.txt 2>&1
cat $O/mfma_shape_*.txt
timeout -k 10 300 python scripts/gemm_ab.py --frags fc1 qkv -- 0:0 0:1 0:2 0:3 > $O/ab_policy_rq.txt 2>&1
cat $O/ab_policy_rq.txt
timeout -k 10 300 python scripts/gemm_ab.py --frags --resid proj fc2 -- 0:0 0:1 0:2 0:3 0:4 0:6 > $O/ab_policy_resid.txt 2>&1
cat $O/ab_policy_resid.txt
timeout -k 10 300 python scripts/gemm_ab.py --frags --qkv qkv -- 0:0 0:1 0:2 0:3 > $O/ab_policy_qkv.txt 2>&1
cat $O/ab_policy_qkv.txt
timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_line.json 2> $O/bench.err
cat $O/bench_line.json
