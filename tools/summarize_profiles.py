#!/usr/bin/env python3
"""Build container: gpurun_out/prof_<tag>/summary (made on the GPU box by gpu_profile_pmc.sh + collect_profiles_pmc.py)
-> profiles/<round>/{kernel_stats_<tag>.csv, counters_by_kernel_<tag>.json, derived_<tag>.json} and profiles/counters.json
(the cached PMC figures bench.py quotes next to its live HIP-event timings).
Derivations (MI355X_MICROARCH.md, HBM / rocprofv3 sections):
  hbm bytes   = FETCH_SIZE[KB]*1024*2 + WRITE_SIZE[KB]*1024   (gfx950 tallies 128-B read requests at 64 B; calibrated for this
                repo's 1-byte gathers in profiles/r01/fetch_size_calibration.json; separate --pmc passes)
  clock       = GRBM_GUI_ACTIVE / 8 / kernel time             (sum over 8 XCDs; reads high on dispatches < 0.3 ms)
  VALU busy   = SQ_ACTIVE_INST_VALU * 4 / (1024 SIMDs * clock * time)   (quad-cycles -> SIMD cycles)
Usage: tools/summarize_profiles.py gpurun_out/prof_r03a r03a [r03]
profiles/counters.json is stamped with the sha256 of csrc/vote.hip and the early-vote split of the profiled run: bench.py quotes
its traffic figures only for a build and a split they were measured on."""
import json
import os
import shutil
import sys

src, tag = sys.argv[1], sys.argv[2]
rnd = sys.argv[3] if len(sys.argv) > 3 else "r03"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dst = os.path.join(ROOT, "profiles", rnd)
os.makedirs(dst, exist_ok=True)
summ = os.path.join(src, "summary")
shutil.copy(os.path.join(summ, "kernel_stats.csv"), os.path.join(dst, f"kernel_stats_{tag}.csv"))
raw = json.load(open(os.path.join(summ, "counters_by_kernel.json")))
cmd = open(os.path.join(summ, "command.txt")).read().strip()
json.dump({"command": "rocprofv3 --kernel-trace --pmc <one counter group per pass> --output-format csv -- " + cmd, "kernels": raw},
          open(os.path.join(dst, f"counters_by_kernel_{tag}.json"), "w"), indent=1)
derived = {}
for key, v in raw.items():
    c = {k: x["mean_per_launch"] for k, x in v["counters"].items()}
    if "duration_us" not in v or not c:
        continue
    t = v["duration_us"]["mean"] * 1e-6
    d = {"kernel": v["kernel"], "launches": v["duration_us"]["launches"], "mean_duration_us": round(v["duration_us"]["mean"], 2)}
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        d["hbm_bytes_per_launch"] = int(c["FETCH_SIZE"] * 1024 * 2 + c["WRITE_SIZE"] * 1024)
        d["hbm_GBps"] = round(d["hbm_bytes_per_launch"] / t / 1e9, 1)
        d["hbm_frac_of_8TBps"] = round(d["hbm_bytes_per_launch"] / t / 8e12, 4)
    if "GRBM_GUI_ACTIVE" in c:
        clock = c["GRBM_GUI_ACTIVE"] / 8 / t
        d["clock_GHz"] = round(clock / 1e9, 3)
        if "SQ_ACTIVE_INST_VALU" in c:
            busy = c["SQ_ACTIVE_INST_VALU"] * 4
            d["valu_busy_simd_cycles"] = int(busy)
            d["simd_cycles_available"] = int(1024 * clock * t)
            d["valu_busy_frac"] = round(busy / (1024 * clock * t), 4)
    if c.get("SQ_LDS_IDX_ACTIVE"):
        d["lds_bank_conflict_frac_of_lds_cycles"] = round(c.get("SQ_LDS_BANK_CONFLICT", 0.0) / c["SQ_LDS_IDX_ACTIVE"], 4)
    if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
        for name in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
            if name in c:
                d[name.lower() + "_frac_of_wave_cycles"] = round(c[name] / c["SQ_WAVE_CYCLES"], 4)
    if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
        d["l2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
    if "SQ_INSTS_VALU" in c:
        d["valu_wave_instructions"] = int(c["SQ_INSTS_VALU"])
    derived[key] = d
json.dump(derived, open(os.path.join(dst, f"derived_{tag}.json"), "w"), indent=1)
stamp = {}
try:
    stamp["vote_hip_sha16"] = open(os.path.join(summ, "vote_hip_sha16.txt")).read().strip()
    line = json.load(open(os.path.join(summ, "bench_line.json")))
    stamp["early_views"] = line["config"].get("early_vote_views")
    stamp["workload"] = line["config"].get("workload")
    json.dump(line, open(os.path.join(dst, f"bench_line_of_profile_{tag}.json"), "w"))
except Exception as e:   # an older summary without the stamp files
    stamp["error"] = str(e)
cached = {"_stamp": stamp, "_comment": f"PMC figures of profiles/{rnd}/counters_by_kernel_{tag}.json (rocprofv3, separate --pmc passes; derivations in "
                      "tools/summarize_profiles.py).  bench.py quotes them as CACHED profile figures next to its live timings."}
for name in ("vote_fused_labels", "vote_early_planes", "vote_fused_final", "vote_early_record", "vote_fused_replay"):
    v = derived.get(name)
    if not v:
        continue
    cached[name] = {
        "hbm_bytes_per_launch": v.get("hbm_bytes_per_launch"),
        "source": f"profiles/{rnd}/derived_{tag}.json",
        "valu_f64": {"bound": "valu_f64", "achieved": v.get("valu_busy_simd_cycles"), "peak": v.get("simd_cycles_available"),
                     "unit": "SIMD-cycles per launch (SQ_ACTIVE_INST_VALU x 4 vs 1024 SIMDs x measured clock x kernel time)",
                     "frac": v.get("valu_busy_frac"), "clock_GHz": v.get("clock_GHz"), "kernel_us_in_profile": v.get("mean_duration_us"),
                     "source": f"cached profile figure: profiles/{rnd}/derived_{tag}.json"}}
json.dump(cached, open(os.path.join(ROOT, "profiles", "counters.json"), "w"), indent=1)
print(json.dumps(derived, indent=1))
