#!/bin/bash
# GPU box: kernel times and HBM-side bytes (FETCH_SIZE / WRITE_SIZE, one counter per pass) of the rasterizer's kernels under
# gsx_render_views, four frames in flight -> gpurun_out/r03/pre_pmc.txt
set -o pipefail
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/pre_pmc; rm -rf $OUT; mkdir -p $OUT $ROOT/gpurun_out/r03
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $ROOT/tools/render_views_loop.py 2 $PRE_OPTS > $OUT/stats.log 2>&1 || { tail -5 $OUT/stats.log; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python3 $ROOT/tools/render_views_loop.py 1 $PRE_OPTS > $OUT/fetch.log 2>&1 || { tail -5 $OUT/fetch.log; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python3 $ROOT/tools/render_views_loop.py 1 $PRE_OPTS > $OUT/write.log 2>&1 || { tail -5 $OUT/write.log; exit 1; }
python3 - <<PY | tee $ROOT/gpurun_out/r03/pre_pmc.txt
import csv, glob, collections
out = "$OUT"
for f in glob.glob(out + "/stats/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if any(k in r["Name"] for k in ("pre_", "blend", "radix", "bin_kernel", "ranges", "bucket", "scan")):
            print("stats", r["Name"][:70], "calls", r["Calls"], "avg_us", round(float(r["AverageNs"]) / 1e3, 1))
for name in ("fetch", "write"):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for f in glob.glob(out + f"/{name}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if any(x in k for x in ("pre_", "blend")):
                acc[k[:60]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for k, d in acc.items():
        kb = sum(d.values()) / len(d)
        print(name, k, "launches", len(d), "MB per launch", round(kb * 1024 * (2 if name == "fetch" else 1) / 1e6, 1), "(FETCH_SIZE x 2: gfx950 tallies 128-B requests at 64 B)" if name == "fetch" else "")
PY
rm -rf $OUT
