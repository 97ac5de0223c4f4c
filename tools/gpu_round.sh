#!/bin/bash
# One GPU-box visit: full gpu tests, option sweep, FETCH_SIZE calibration, rocprofv3 passes.
# Usage (through gpurun): tools/gpu_round.sh <tag>
set -o pipefail
TAG=${1:-x}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/round_$TAG
mkdir -p $OUT
cd $ROOT
python -m pytest tests -x -q -m gpu > $OUT/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -3 $OUT/pytest.log
for o in "" "--seg-cell 1" "--opt wave_cull=0" "--opt seg_coarse=0" "--opt seg_coarse=0 --opt wave_cull=0" "--opt xcd_swizzle=1" "--opt flat_project=0" "--opt xcd_swizzle=0" "--opt seg_tiled=0" "--opt spatial_sort=0" ""; do
  timeout -k 10 300 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --render-views 0 $o 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$o', d['ms_per_step'], d['roofline']['kernel_ms'], d['value'])" | tee -a $OUT/sweep.log
done
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29631 bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo --gaussians 1000000 --views 40 > $OUT/bench_2rank_gloo_v3.json 2> $OUT/bench_2rank_gloo_v3.err; echo "2-rank sparse rehearsal rc=$?"
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29635 bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo --exchange a2a --gaussians 1000000 --views 40 > $OUT/bench_2rank_gloo.json 2> $OUT/bench_2rank_gloo.err; echo "2-rank a2a rehearsal rc=$?"
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29633 bench.py --gpus 2 --steps 3 --warmup 1 --backend gloo --exchange allreduce --gaussians 1000000 --views 40 > $OUT/bench_2rank_gloo_v1.json 2> $OUT/bench_2rank_gloo_v1.err; echo "2-rank allreduce rehearsal rc=$?"
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --render-views 0 --force-exchange-path > $OUT/bench_exchange_path.json 2>/dev/null; echo "exchange-path rc=$?"
timeout -k 10 300 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --render-views 0 --force-exchange-path --exchange a2a > $OUT/bench_exchange_path_v2.json 2>/dev/null; echo "exchange-path v2 rc=$?"
timeout -k 10 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/calib_fetch -- $ROOT/tools/calib_gather > $OUT/calib_fetch.log 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_MISS_sum --output-format csv -d $OUT/calib_tcc -- $ROOT/tools/calib_gather > $OUT/calib_tcc.log 2>&1
BENCH="python3 $ROOT/bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-profile --render-views 0"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- $BENCH > $OUT/stats.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $BENCH > $OUT/pmc_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $BENCH > $OUT/pmc_write.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS --output-format csv -d $OUT/pmc_sq -- $BENCH > $OUT/pmc_sq.log 2>&1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --output-format csv -d $OUT/pmc_tcc -- $BENCH > $OUT/pmc_tcc.log 2>&1
RB="python3 $ROOT/bench.py --steps 1 --warmup 1 --cpu-sample 0 --no-profile --render-views 4"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/render_stats -- $RB > $OUT/render_stats.log 2>&1
find $OUT -name "*kernel_trace.csv" -size +5M -delete
ls $OUT
