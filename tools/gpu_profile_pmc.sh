#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace stats + PMC passes (one counter group per pass, never combined
# with sys/hip/hsa tracing) of bench.py: labeler headline run + device-map side leg + rasterizer leg.
# Usage: tools/gpu_profile_pmc.sh <tag> [bench args]; outputs under gpurun_out/prof_<tag>/ ; then tools/collect_profiles_pmc.py (on the box)
# and tools/summarize_profiles.py (build container).  One counter group per pass; FETCH_SIZE and WRITE_SIZE never share one.
set -o pipefail
TAG=${1:-r03}; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 3 --warmup 1 --cpu-sample 0 --side-steps 2 --no-profile --render-views 4 $*"
echo "python3 bench.py $ARGS" > $OUT/command.txt
pass() {  # name, rocprofv3 options...
  local name=$1; shift
  echo "pass $name" >&2
  timeout -k 10 280 rocprofv3 "$@" --output-format csv -d $OUT/$name -- python3 $ROOT/bench.py $ARGS > $OUT/$name.log 2>&1 || { echo "pass $name failed" >&2; tail -5 $OUT/$name.log >&2; return 1; }
}
pass stats --kernel-trace --stats || exit 1
grep "^{" $OUT/stats.log | tail -1 > $OUT/bench_line.json
sha256sum $ROOT/3d_gaussian_splatting_project_amd/csrc/vote.hip | cut -c1-16 > $OUT/vote_hip_sha16.txt
pass pmc_fetch --kernel-trace --pmc FETCH_SIZE || exit 1
pass pmc_write --kernel-trace --pmc WRITE_SIZE || exit 1
pass pmc_sq --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS || exit 1
pass pmc_lds --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR || exit 1
pass pmc_grbm --kernel-trace --pmc GRBM_GUI_ACTIVE || exit 1
pass pmc_tcc --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum || exit 1
find $OUT -name "*.csv" | sed "s|$OUT/||" | sort | head -40
