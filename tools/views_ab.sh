#!/bin/bash
# does the seg pool fitting into the 256 MB Infinity Cache change the per-view cost?  (1080p: 2.07 MB per view)
set -o pipefail
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/views_ab; mkdir -p $OUT; rm -f $OUT/ab.txt; cd $ROOT
for v in 200 120 100 60 30 200; do
  echo "== views $v" >> $OUT/ab.txt
  timeout -k 10 200 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --render-views 0 --views $v --opt wave_cull=0 2>>$OUT/err.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['ms_per_step']/$v*200, d['value'])" >> $OUT/ab.txt || exit 1
done
cat $OUT/ab.txt
