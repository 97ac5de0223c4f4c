#!/bin/bash
# GPU box, round 3: the filter's own tests, then the labeler leg of bench.py with and without the fp32 filter (same box, same process order)
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r03; mkdir -p $OUT; cd $ROOT
timeout -k 10 500 python -m pytest tests/test_vote_gpu.py -x -q -k "filter or golden or extreme or certified or tuning or early_vote_on or pixel_accurate or randomised" > $OUT/filter_tests.txt 2>&1 || { tail -30 $OUT/filter_tests.txt; exit 1; }
tail -3 $OUT/filter_tests.txt
for f in 1 0 1 0; do
  timeout -k 10 200 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --render-views 0 --opt filter_project=$f > $OUT/bench_filter_$f.json 2>$OUT/bench_filter_$f.err || { tail -5 $OUT/bench_filter_$f.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("$OUT/bench_filter_$f.json").read().strip().splitlines()[-1])
print("filter_project=$f ms_per_step", d["ms_per_step"], "kernel_resident_ms", d["side"].get("kernel_resident_ms_per_step"), "resident_ms", d["side"].get("resident_ms_per_step"), {k:v["ms_per_launch"] for k,v in d["kernels_ms"].items()})
PY
done
