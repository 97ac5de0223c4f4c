#!/bin/bash
# A/B of the Gaussians-per-thread variants of the vote kernel (tests first, then bench lines)
set -o pipefail
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/gpt_ab; mkdir -p $OUT; rm -f $OUT/ab.txt; cd $ROOT
timeout -k 10 600 python -m pytest tests/test_vote_gpu.py -x -q -m gpu -k "not cli" > $OUT/tests.log 2>&1 || { tail -30 $OUT/tests.log; exit 1; }
tail -2 $OUT/tests.log
for o in "--opt view_pairs=0" "--opt view_pairs=1" "--opt view_pairs=1 --opt vote_unroll=4" "--opt view_pairs=0" "--opt view_pairs=1" "--seg-cell 1 --opt view_pairs=1"; do
  echo "== $o" >> $OUT/ab.txt
  timeout -k 10 200 python bench.py --steps 10 --warmup 3 --cpu-sample 0 --render-views 0 $o 2>>$OUT/err.log | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'], d['value'], d['config'].get('wave_views_culled_fraction'))" >> $OUT/ab.txt || exit 1
done
cat $OUT/ab.txt
