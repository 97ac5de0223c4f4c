#!/bin/bash
# GPU box, end of round 2: the GPU suite, the driver's bench command, the multi-rank rehearsals -> gpurun_out/final_r02/
set -o pipefail
OUT=gpurun_out/final_r02
mkdir -p $OUT
timeout -k 10 600 python -m pytest tests -x -q -m gpu > $OUT/pytest_gpu.log 2>&1 || { tail -30 $OUT/pytest_gpu.log; exit 1; }
tail -3 $OUT/pytest_gpu.log
timeout -k 10 300 python bench.py --gpus 1 --steps 20 --warmup 5 2> $OUT/bench.err | grep "^{" > $OUT/bench_r02.json || { tail -20 $OUT/bench.err; exit 1; }
python -c "import json; d=json.load(open('$OUT/bench_r02.json')); print('value %.3e  ms %.3f  render %s' % (d['value'], d['ms_per_step'], d['render']['views_per_s']))"
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29641 bench.py --gpus 3 --steps 3 --warmup 1 --backend gloo --cpu-sample 0 --render-views 0 2> $OUT/c3_gloo3.err | grep "^{" > $OUT/rehearsal_c3_3ranks_gloo_one_gpu.json || { tail -20 $OUT/c3_gloo3.err; exit 1; }
timeout -k 10 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 3 --master-addr 127.0.0.1 --master-port 29643 bench.py --gpus 3 --steps 2 --warmup 1 --backend gloo --gaussians 10000000 --views 300 --width 3840 --height 2160 --cpu-sample 0 --render-views 0 2> $OUT/c5_gloo3.err | grep "^{" > $OUT/rehearsal_10M_300views_4K_3ranks_gloo_one_gpu.json || { tail -20 $OUT/c5_gloo3.err; exit 1; }
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/final_r02/rehearsal*.json')):
    d=json.load(open(f)); print(f, d['ms_per_step'], d['config']['workload'], d['config'].get('exchanged_labels_equal_single_gpu_vote'))
PY
