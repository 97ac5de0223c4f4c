#!/bin/bash
set -o pipefail
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/ablate; mkdir -p $OUT; rm -f $OUT/ab.txt; cd $ROOT
for m in 0 32 2; do for cull in 0 1; do
  if [ $m = 0 ]; then unset GSX_LIBRARY; else export GSX_LIBRARY=$ROOT/tools/ablate/libgsx_$m.so; fi
  echo "== ablate mask $m wave_cull $cull" >> $OUT/ab.txt
  timeout -k 10 200 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --render-views 0 --opt wave_cull=$cull 2>>$OUT/err.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])" >> $OUT/ab.txt || exit 1
done; done
cat $OUT/ab.txt
