#!/bin/bash
# occupancy sensitivity of the vote kernel: fewer classes -> smaller LDS rows -> more waves per CU, same arithmetic
set -o pipefail
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/occ_ab; mkdir -p $OUT; rm -f $OUT/ab.txt; cd $ROOT
for cl in 150 123 91 59 27 150; do
  echo "== classes $cl" >> $OUT/ab.txt
  timeout -k 10 200 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --render-views 0 --classes $cl 2>>$OUT/err.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'])" >> $OUT/ab.txt || exit 1
done
cat $OUT/ab.txt
