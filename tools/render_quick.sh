#!/bin/bash
# GPU box: rasterizer leg only, for each option string given as an argument -> stdout
for o in "$@"; do
  echo "== $o"
  python bench.py --steps 1 --warmup 0 --views 8 --gaussians 100000 --cpu-sample 0 --side-steps 0 --render-views 8 $o 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())['render']
print(d['views_per_s'], 'views/s; pairs', d['tile_splat_pairs_per_view'], 'consumed', d['pairs_consumed_per_view'], 'kernel sum', d['kernel_ms_sum_per_view'])
print(' '.join(f'{k}={v}' for k,v in d['kernel_ms_per_view'].items()))
"
done
