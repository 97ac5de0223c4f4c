#!/bin/bash
# GPU box, end of a round: the GPU suite, the driver's bench command, the multi-rank rehearsals -> gpurun_out/final_<tag>/
#   tools/gpu_final.sh <tag> [tests|bench|rehearse ...]   (default: all three)
set -o pipefail
TAG=${1:-r03}; shift
WHAT=${*:-tests bench rehearse}
OUT=gpurun_out/final_$TAG
mkdir -p $OUT
for w in $WHAT; do case $w in
tests)
  timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=15 > $OUT/pytest_gpu.log 2>&1 || { tail -40 $OUT/pytest_gpu.log; exit 1; }
  tail -22 $OUT/pytest_gpu.log ;;
bench)
  timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 2> $OUT/bench.err | grep "^{" > $OUT/bench_$TAG.json || { tail -20 $OUT/bench.err; exit 1; }
  python -c "import json; d=json.load(open('$OUT/bench_$TAG.json')); print('value %.3e  ms %.3f  render %s  roofline %s %s' % (d['value'], d['ms_per_step'], d['render']['views_per_s'], d['roofline']['kernel'], d['roofline']['frac']))" ;;
rehearse)
  # the N > 1 code path: one rank through real RCCL (every collective runs, minus the peers), three ranks over gloo sharing the GPU
  GSX_DIST_FORCE_COLLECTIVES=1 timeout -k 10 300 python bench.py --gpus 1 --steps 10 --warmup 3 --cpu-sample 0 --render-views 0 2> $OUT/rccl1.err | grep "^{" > $OUT/rehearsal_c3_1rank_rccl.json || { tail -20 $OUT/rccl1.err; exit 1; }
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29641 bench.py --gpus 3 --steps 3 --warmup 1 --backend gloo --cpu-sample 0 --render-views 0 2> $OUT/c3_gloo3.err | grep "^{" > $OUT/rehearsal_c3_3ranks_gloo_one_gpu.json || { tail -20 $OUT/c3_gloo3.err; exit 1; }
  python - <<PY
import json, glob
for f in sorted(glob.glob('$OUT/rehearsal*.json')):
    d = json.load(open(f)); c = d['config']
    print(f, 'ms_per_step', d['ms_per_step'], 'labels ok', c.get('exchanged_labels_equal_single_gpu_vote'), 'phases', c.get('phases_ms'))
PY
  ;;
esac; done
