#!/bin/bash
# GPU box, round 3: rasterizer tests, then views/s by frames in flight in a process that has / has not labelled before (the second
# stream exists or not), with the first extra frame on that stream or on one more stream, with one pre pass per group or per frame
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r03; mkdir -p $OUT; cd $ROOT
timeout -k 10 600 python -m pytest tests/test_render_gpu.py tests/test_vote_gpu.py -x -q -m gpu -k "render or rccl or cli or frames" > $OUT/render_tests.txt 2>&1 || { tail -40 $OUT/render_tests.txt; exit 1; }
tail -3 $OUT/render_tests.txt
for v in "" "early" "multi0" "early multi0" "early share0" "early share0 multi0"; do
  echo "== render_frames_probe $v" | tee -a $OUT/render_frames_r03.txt
  GSX_PROBE_SHARE=$([[ "$v" == *share0* ]] && echo 0 || echo 1) timeout -k 10 300 python tools/render_frames_probe.py $v 2>/dev/null | tee -a $OUT/render_frames_r03.txt || exit 1
done
