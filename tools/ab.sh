#!/bin/bash
# GPU box: A/B of bench.py's labeler leg over a list of argument sets, same box, same build, one line per run.
#   tools/ab.sh <tag> "<bench args of run 1>" "<bench args of run 2>" ...
# e.g. tools/ab.sh filter "--opt filter_project=1" "--opt filter_project=0" "--opt filter_project=1" "--opt filter_project=0"
#      tools/ab.sh shapes "--gaussians 500000 --views 16 --width 1280 --height 720" "--gaussians 10000000 --views 125 --width 3840 --height 2160"
#      GSX_LIBRARY=tools/ablate/libgsx_2.so tools/ab.sh ablate ""        (a timing-only build of tools/ablate.sh)
#      AB_STEPS=20 AB_WARMUP=5 tools/ab.sh affinity "GSX_HOST_AFFINITY=l3" "GSX_HOST_AFFINITY=node" "GSX_HOST_AFFINITY=0"   (per-run environment)
# Replaces round 1-2's one-off scripts (gpt_ab, occ_ab, mem_ab, views_ab, configs_ab, ablate_run, render_ab, sort_ab, ...).
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/ab_$TAG; mkdir -p $OUT; rm -f $OUT/ab.txt; cd $ROOT
i=0
for o in "$@"; do
  i=$((i+1))
  # leading NAME=VALUE words of an argument set are environment for that run only (GSX_HOST_AFFINITY=node, GSX_LIBRARY=..)
  envs=(); args=()
  for w in $o; do if [[ ${#args[@]} -eq 0 && $w =~ ^[A-Z_][A-Z0-9_]*=.*$ ]]; then envs+=("$w"); else args+=("$w"); fi; done
  env "${envs[@]}" timeout -k 10 400 python bench.py --steps ${AB_STEPS:-10} --warmup ${AB_WARMUP:-3} --cpu-sample 0 --render-views 0 "${args[@]}" > $OUT/run_$i.json 2>>$OUT/err.log || { tail -5 $OUT/err.log; exit 1; }
  python - "$o" $OUT/run_$i.json <<'PY' | tee -a $OUT/ab.txt
import json, sys
d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
k = {n: v["ms_per_launch"] for n, v in (d.get("kernels_ms") or {}).items() if n.startswith("vote_")}
s = d.get("side") or {}
print(f"[{sys.argv[1]}] ms_per_step {d['ms_per_step']} value {d['value']:.4g} kernel_resident_ms {s.get('kernel_resident_ms_per_step')} "
      f"resident_ms {s.get('resident_ms_per_step')} culled {d['config'].get('wave_views_culled_fraction')} {k}")
PY
done
