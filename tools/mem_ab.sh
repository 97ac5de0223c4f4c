#!/bin/bash
# memory sensitivity of the vote kernel: same Gaussians/views/classes (same arithmetic), smaller maps (less gather traffic)
set -o pipefail
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/mem_ab; mkdir -p $OUT; rm -f $OUT/ab.txt; cd $ROOT
for wh in "1920 1080" "960 540" "480 270" "3840 2160" "1920 1080"; do
  set -- $wh
  echo "== $1 x $2" >> $OUT/ab.txt
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --render-views 0 --width $1 --height $2 --opt flat_project=1 2>>$OUT/err.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])" >> $OUT/ab.txt || exit 1
done
cat $OUT/ab.txt
