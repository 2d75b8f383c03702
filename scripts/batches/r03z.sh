#!/bin/bash
set -e
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03z; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "attention or geometries or golden" > $O/geom.log 2>&1 || { tail -60 $O/geom.log; exit 1; }
tail -3 $O/geom.log
