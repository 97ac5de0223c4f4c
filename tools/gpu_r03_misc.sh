#!/bin/bash
# GPU box, round 3: the early vote's two forms side by side, the protocol's host-side cost with a one-rank RCCL group, and a
# randomised soak (tests/soak.py is where the oracle is used as the checker: nothing under tools/ touches oracle/)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r03; mkdir -p $OUT; cd $ROOT
tools/ab.sh replay "--opt early_replay=0" "--opt early_replay=1" "--opt early_replay=0" "--opt early_replay=1" || exit 1
timeout -k 10 300 python tools/pipeline_cost.py 2>/dev/null | tee $OUT/pipeline_cost.txt || exit 1
for seed in ${SOAK_SEEDS:-311 312 313}; do timeout -k 10 400 python tests/soak.py $seed 1500 2>&1 | tail -1 | tee -a $OUT/soak.txt || exit 1; done
