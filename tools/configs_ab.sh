#!/bin/bash
# the labeler on the other BASELINE config shapes (per GPU): C2 500k x 16 @720p, C3 3M x 200 @1080p, C5-shape 10M x 125 @4K
set -o pipefail
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/configs_ab; mkdir -p $OUT; rm -f $OUT/ab.txt; cd $ROOT
run() { echo "== $*" >> $OUT/ab.txt; timeout -k 10 500 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --render-views 0 "$@" 2>>$OUT/err.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'], d['config']['visible_fraction'], d['config']['wave_views_culled_fraction'])" >> $OUT/ab.txt || exit 1; }
run --gaussians 500000 --views 16 --width 1280 --height 720
run --gaussians 3000000 --views 200 --width 1920 --height 1080
run --gaussians 3000000 --views 200 --width 3840 --height 2160
run --gaussians 10000000 --views 125 --width 3840 --height 2160
run --gaussians 10000000 --views 125 --width 3840 --height 2160 --seg-cell 1
cat $OUT/ab.txt
