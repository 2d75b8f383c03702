#!/bin/bash
# round 3, GPU batch d: Swin natural-scale end-to-end + calibration-trace tests, then the full suite and the bench A/B of the fragment forms
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03d; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_swin.py tests/test_gpu_modules.py -m gpu -x -q -k "natural or calibration" > $O/new_tests.log 2>&1 || { tail -80 $O/new_tests.log; exit 1; }
tail -3 $O/new_tests.log
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1 || { tail -60 $O/gpu_tests.log; exit 1; }
tail -3 $O/gpu_tests.log
for i in 1 2; do
IVIT_FRAGS16=0 timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_f32_$i.json 2>> $O/bench.err
timeout -k 10 300 python bench.py --steps 50 --warmup 10 --no-cpu-baseline > $O/bench_f16_$i.json 2>> $O/bench.err
done
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r03d/bench_f*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["ms_per_step"], d["value"], d["roofline"]["avg_launch_ms_raw"], d["roofline"]["frac"])
PY
