#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04j; mkdir -p $O
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_torchrun.json 2> $O/bench_torchrun.err || { tail -30 $O/bench_torchrun.err; exit 1; }
cat $O/bench_torchrun.json | cut -c1-600
