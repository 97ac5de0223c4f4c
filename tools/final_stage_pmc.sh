#!/bin/bash
# GPU box: counters of the early vote's last stage next to the one-piece labels kernel (tools/final_stage_loop.py), one counter
# group per rocprofv3 pass, the program directly after `--`.  Output: gpurun_out/pmc_final/<pass>/ + summary.txt
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_final
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() {
  local name=$1; shift
  echo "pass $name" >&2
  timeout -k 10 150 rocprofv3 "$@" --output-format csv -d $OUT/$name -- python3 $ROOT/tools/final_stage_loop.py 4 > $OUT/$name.log 2>&1 || { echo "pass $name failed" >&2; tail -5 $OUT/$name.log >&2; return 1; }
}
pass stats --kernel-trace --stats || exit 1
pass sq1 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_INSTS_SALU SQ_INSTS_LDS || exit 1
pass sq2 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR || exit 1
# one TCC counter per pass: FETCH_SIZE takes 3 of the 4 TCC slots, WRITE_SIZE 2 (MI355X_MICROARCH.md) - together they abort rocprofv3
pass mem_fetch --kernel-trace --pmc FETCH_SIZE || exit 1
pass mem_write --kernel-trace --pmc WRITE_SIZE || exit 1
pass grbm --kernel-trace --pmc GRBM_GUI_ACTIVE || exit 1
pass occ --kernel-trace --pmc SQ_LEVEL_WAVES SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU || echo "occ pass failed (counter names)" >&2
python3 - <<PY > $OUT/summary.txt
import csv, glob, collections, os
out = "$OUT"
for d in sorted(glob.glob(out + "/*/")):
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "vote_fused" in k:
                acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, cs in acc.items():
            print(os.path.basename(os.path.dirname(d)), k, {c: round(sum(v) / len(v)) for c, v in cs.items()}, "launches", len(next(iter(cs.values()))))
    for f in glob.glob(d + "**/*kernel_stats.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "vote_fused" in r["Name"]:
                print("stats", r["Name"][:60], r["Calls"], r["AverageNs"])
PY
cat $OUT/summary.txt
