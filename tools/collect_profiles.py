#!/usr/bin/env python3
"""Turn one tools/gpu_round.sh visit (gpurun_out/round_<tag>/) into the committed summaries under profiles/<round>/:
kernel stats CSVs, per-launch PMC means of the vote kernel, the sweep, the bench lines, and profiles/traffic.json
(HBM-side bytes per launch = FETCH_SIZE * 1024 * 2 + WRITE_SIZE * 1024, gfx950 correction per MI355X_MICROARCH.md,
calibrated in fetch_size_calibration.json).  Usage: tools/collect_profiles.py gpurun_out/round_r01e profiles/r01 final"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
os.makedirs(dst, exist_ok=True)


def one(pattern):
    f = glob.glob(os.path.join(src, pattern))
    return f[0] if f else None


for name, sub in (("vote_kernel_stats", "stats"), ("render_kernel_stats", "render_stats")):
    f = one(f"{sub}/*/*kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(dst, f"{name}_{tag}.csv"))

counters = {}
for sub in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_tcc"):
    f = one(f"{sub}/*/*counter_collection.csv")
    if not f:
        continue
    acc = collections.defaultdict(list)
    kernel = None
    for r in csv.DictReader(open(f)):
        if "vote_fused_labels" in r["Kernel_Name"]:
            kernel = r["Kernel_Name"]
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        counters[k] = {"launches": len(v), "mean_per_launch": sum(v) / len(v)}
if counters:
    fetch = counters.get("FETCH_SIZE", {}).get("mean_per_launch")
    write = counters.get("WRITE_SIZE", {}).get("mean_per_launch")
    out = {"kernel": kernel,
           "command": "tools/gpu_round.sh: rocprofv3 --kernel-trace --pmc <one group per pass> --output-format csv -- "
                      "python3 bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-profile --render-views 0",
           "counters": counters}
    if fetch is not None and write is not None:
        traffic = int(fetch * 1024 * 2 + write * 1024)
        out["hbm_bytes_per_launch"] = traffic
        out["formula"] = "FETCH_SIZE[KB]*1024*2 + WRITE_SIZE[KB]*1024"
        json.dump({"_comment": f"HBM-side bytes per launch from rocprofv3 PMC passes ({dst}/vote_pmc_{tag}.json): "
                               "FETCH_SIZE*1024*2 (gfx950 correction, calibrated for this access pattern in "
                               "profiles/r01/fetch_size_calibration.json) + WRITE_SIZE*1024",
                   "vote_fused_labels": traffic}, open(os.path.join(os.path.dirname(dst.rstrip("/")), "traffic.json"), "w"), indent=1)
    json.dump(out, open(os.path.join(dst, f"vote_pmc_{tag}.json"), "w"), indent=1)

for name in ("sweep.log", "bench.json", "bench_exchange_path.json", "bench_exchange_path_v2.json"):
    f = os.path.join(src, name)
    if os.path.exists(f):
        base, ext = os.path.splitext(name)
        target = {"sweep.log": f"ablation_{tag}.txt", "bench.json": f"bench_{tag}.json"}.get(name, name)
        lines = [l for l in open(f) if not l.startswith("[Gloo]")]
        open(os.path.join(dst, target), "w").writelines(lines)
print("collected into", dst, sorted(os.listdir(dst)))
