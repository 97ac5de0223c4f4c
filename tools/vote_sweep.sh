#!/bin/bash
# GPU box: vote kernel alone (packed maps resident) under tuning options, pixel-accurate maps -> stdout
for o in "" "--opt vote_unroll=4" "--opt xcd_swizzle=16" "--opt xcd_swizzle=64" "--opt xcd_swizzle=128" "--opt lds_batch=1" "--opt fast_div=1 --opt flat_project=0" "--opt wave_cull=0" "--opt seg_coarse=0" "--seg-cell 4"; do
  python bench.py --steps 2 --warmup 1 --cpu-sample 0 --render-views 0 --side-steps 5 $o 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('%-45s kernel-resident %.4f ms   resident %.3f ms   value-step %.2f ms' % ('$o', d['side']['kernel_resident_ms_per_step'], d['side']['resident_ms_per_step'], d['ms_per_step']))"
done
