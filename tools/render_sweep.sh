#!/bin/bash
# GPU box: rasterizer leg of bench.py under different binning options -> gpurun_out/render_sweep.txt
OUT=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out/render_sweep.txt
: > $OUT
for o in "--opt render_phases=1 --opt exact_cull=0" "--opt render_phases=1" "--opt render_phases=2" "--opt render_phases=2 --opt render_phase_ratio=8" "--opt render_phases=3" "--opt render_phases=3 --opt render_phase_ratio=3" "--opt render_phases=3 --opt render_phase_ratio=6" "--opt render_phases=4" "--opt render_phases=4 --opt render_phase_ratio=3" "--opt render_phases=5 --opt render_phase_ratio=3" "--opt render_phases=3 --opt exact_cull=0" "--opt render_phases=3 --opt blend_pk2=2" "--opt render_phases=3 --opt blend_pk2=0"; do
  echo "== $o" >> $OUT
  python bench.py --steps 1 --warmup 0 --views 8 --gaussians 100000 --cpu-sample 0 --side-steps 0 --render-views 8 $o 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())['render']
print(d['views_per_s'], 'views/s; pairs', d['tile_splat_pairs_per_view'], 'consumed', d['pairs_consumed_per_view'], 'kernel sum', d['kernel_ms_sum_per_view'])
print(' '.join(f'{k}={v}' for k,v in d['kernel_ms_per_view'].items()))
" >> $OUT
done
cat $OUT
