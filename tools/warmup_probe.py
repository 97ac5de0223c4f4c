#!/usr/bin/env python3
"""GPU box: hand-over and tail of every run from the first one of a process (how long does a process take to reach its steady
state?), early vote on (argv[1] = 1, default) or off (0); configs[2], seg-cell 4 maps."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("3d_gaussian_splatting_project_amd")
scene = pkg.scene
import torch
torch.cuda.set_device(0)
pkg.bind_to_gpu_numa_node(0)
n, V, W, H = 3_000_000, 200, 1920, 1080
pos = scene.make_positions(n, scene.BASE_SEED + 3)
cams = [pkg.Camera.from_dict(c) for c in scene.make_cameras(V, W, H, convention="w2c")]
base = [scene.make_segmap(H, W, 150, 3000 + v, cell=4) for v in range(8)]
segs = [base[v % 8].copy() for v in range(V)]
out = np.empty(n, np.int32)
for f in ("enabled", "defrag", "khugepaged/pages_to_scan", "khugepaged/scan_sleep_millisecs", "khugepaged/defrag"):
    try:
        print("thp", f, open("/sys/kernel/mm/transparent_hugepage/" + f).read().strip(), flush=True)
    except OSError as e:
        print("thp", f, e)
try:
    print("numa_balancing", open("/proc/sys/kernel/numa_balancing").read().strip())
except OSError as e:
    print("numa_balancing", e)


def huge_kb():
    for line in open("/proc/self/smaps_rollup"):
        if line.startswith("AnonHugePages"):
            return int(line.split()[1])
    return -1


def pages_on_nodes():
    """pages of the first map per NUMA node (move_pages query through the library's own helper is not exported: numa_maps)"""
    addr = segs[0].ctypes.data
    try:
        for line in open("/proc/self/numa_maps"):
            a = int(line.split()[0], 16)
            if a <= addr < a + (64 << 20) and "anon" in line:
                return " ".join(t for t in line.split() if t.startswith("N") or t.startswith("kernelpagesize"))
    except OSError:
        pass
    return "?"


with pkg.Context(0) as ctx:
    ctx.set_option("early_vote", int(sys.argv[1]) if len(sys.argv) > 1 else 1)
    ctx.upload_positions(pos)
    ctx.profile(True)
    for r in range(40):
        t0 = time.perf_counter()
        ctx.vote_begin(150, 0, V)
        for v in range(V):
            ctx.vote_view(cams[v], segs[v])
        t1 = time.perf_counter()
        ctx.vote_finalize(out=out)
        t2 = time.perf_counter()
        ks = {k: ctx.profile_get(k) for k in ("vote_early_planes", "vote_fused_final", "vote_fused_labels") if k in ctx.profile_names()}
        ctx.profile(True)   # reset
        print(f"run {r:2d}: hand-over {(t1 - t0) * 1e3:6.3f}  tail {(t2 - t1) * 1e3:6.3f}  span {(t2 - t0) * 1e3:6.3f} ms   " +
              "  ".join(f"{k} {ms / max(c, 1):.3f}" for k, (c, ms) in ks.items() if c) + f"   AnonHuge {huge_kb() >> 10} MB  map0 {pages_on_nodes() if r % 5 == 0 else ''}", flush=True)
