#!/bin/bash
# GPU box: the packer pool's placement policy under bench.py's own run (the driver's flags), interleaved
#   l3 = round 2 (workers dealt over the node's L3 domains), node = bound to the NUMA node only, 0 = not bound at all
for rep in 1 2 3; do for pol in l3 node 0; do
  GSX_HOST_AFFINITY=$pol python bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 --render-views 0 2>/dev/null | grep "^{" | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']
print('$pol', 'ms_per_step %.3f' % d['ms_per_step'], 'median %.2f' % c['step_ms_median_min_max'][0], 'steps', c['step_ms'], 'throttled', c['cpu_quota_throttling_in_timed_region']['throttled_periods'])"
done; done
