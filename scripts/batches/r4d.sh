#!/bin/bash
# round 4, batch d: SQ counters of the streaming LayerNorm and the grouped one at the headline shape
set -eu
cd "$GRAFT_REPO_ROOT"
R=$GRAFT_REPO_ROOT
bash scripts/pmc_one.sh ln_stream layernorm -- python3 $R/scripts/ln_one.py 0 0
bash scripts/pmc_one.sh ln_grouped layernorm -- python3 $R/scripts/ln_one.py 3 0
