#!/usr/bin/env python3
"""Runs ON THE GPU BOX right after tools/gpu_profile_pmc.sh: reduces the raw rocprofv3 CSVs (too large to travel) of
gpurun_out/prof_<tag>/ to gpurun_out/prof_<tag>/summary/: the kernel stats table, and per kernel of interest the
per-launch means of every collected counter (summed over the XCD / SE / instance rows rocprofv3 emits per dispatch).
Usage: tools/collect_profiles_pmc.py gpurun_out/prof_r03a"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

src = sys.argv[1]
dst = os.path.join(src, "summary")
os.makedirs(dst, exist_ok=True)
KERNELS = {"vote_fused_labels": "vote_fused_labels_kernel", "seg_pack_fused": "seg_pack_fused_kernel", "blend2": "blend2_kernel", "blend4": "blend4_kernel",
           "radix_scatter": "radix_scatter_kernel", "radix_hist": "radix_hist_kernel", "pre": "pre_kernel", "bin_count": "bin_kernel<false",
           "bin_emit": "bin_kernel<true", "ranges": "ranges_kernel", "bucket": "bucket_kernel", "unpermute": "unpermute_labels_kernel",
           "seg_expand": "seg_expand_kernel", "labels_narrow": "labels_narrow_kernel",
           "vote_early_planes": "vote_fused_planes_kernel", "vote_fused_final": "vote_fused_final_kernel",
           "vote_early_record": "vote_record_kernel", "vote_fused_replay": "vote_fused_replay_kernel", "pre_multi": "pre_multi_kernel"}


def one(pattern):
    f = glob.glob(os.path.join(src, pattern))
    return f[0] if f else None


f = one("stats/*/*kernel_stats.csv")
if f:
    shutil.copy(f, os.path.join(dst, "kernel_stats.csv"))
shutil.copy(os.path.join(src, "command.txt"), os.path.join(dst, "command.txt"))
for extra in ("bench_line.json", "vote_hip_sha16.txt"):   # the profiled run's own JSON line and the stamp of the kernels' source
    if os.path.exists(os.path.join(src, extra)):
        shutil.copy(os.path.join(src, extra), os.path.join(dst, extra))

out = {k: {"kernel": None, "counters": {}} for k in KERNELS}
for sub in sorted(glob.glob(os.path.join(src, "pmc_*"))):
    f = one(os.path.basename(sub) + "/*/*counter_collection.csv")
    if not f or not os.path.isfile(f):
        continue
    # (kernel key, dispatch id, counter) -> sum over the rows of that dispatch
    acc = collections.defaultdict(float)
    names = {}
    with open(f) as fh:
        for r in csv.DictReader(fh):
            kn = r["Kernel_Name"]
            for key, pat in KERNELS.items():
                if pat in kn:
                    acc[(key, r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"])
                    names[key] = kn
                    break
    per = collections.defaultdict(list)
    for (key, disp, cname), v in acc.items():
        per[(key, cname)].append(v)
    for (key, cname), vals in per.items():
        out[key]["kernel"] = names[key]
        out[key]["counters"][cname] = {"launches": len(vals), "mean_per_launch": sum(vals) / len(vals), "min": min(vals), "max": max(vals)}

# kernel durations from the plain kernel trace of the stats pass (ns)
f = one("stats/*/*kernel_trace.csv")
if f:
    dur = collections.defaultdict(list)
    with open(f) as fh:
        for r in csv.DictReader(fh):
            for key, pat in KERNELS.items():
                if pat in r["Kernel_Name"]:
                    dur[key].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
                    break
    for key, v in dur.items():
        out[key]["duration_us"] = {"launches": len(v), "mean": sum(v) / len(v) / 1e3, "min": min(v) / 1e3, "max": max(v) / 1e3}
out = {k: v for k, v in out.items() if v["counters"] or "duration_us" in v}
json.dump(out, open(os.path.join(dst, "counters_by_kernel.json"), "w"), indent=1)
for sub in glob.glob(os.path.join(src, "*")):
    if os.path.basename(sub) != "summary":
        shutil.rmtree(sub) if os.path.isdir(sub) else os.remove(sub)
print("summary:", sorted(os.listdir(dst)), {k: sorted(v["counters"]) for k, v in out.items()})
