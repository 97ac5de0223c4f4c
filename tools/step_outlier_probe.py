#!/usr/bin/env python3
"""GPU box: which steps of a run of labeler steps are slow, and on which side?  configs[2] with cell-4 maps; per step the hand-over
(host: first vote_view -> last one returned) and the tail (-> labels on the host), around fences, pauses and profile resets; the
cgroup's throttling counters next to it.  argv[1]: host threads (0 = the library's default)."""
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
how = sys.argv[2] if len(sys.argv) > 2 else "numpy"
if how == "torch":      # memory of a CPU tensor (what `.cpu().numpy()` of a segmentation network's output is): plain 4 KB pages
    keep = [torch.from_numpy(base[v % 8]).clone() for v in range(V)]
    segs = [t.numpy() for t in keep]
else:                   # numpy's own allocation (np.load of a _segmap.npy): madvise(MADV_HUGEPAGE) above 4 MB
    segs = [base[v % 8].copy() for v in range(V)]


def huge_mb():
    for line in open("/proc/self/smaps_rollup"):
        if line.startswith("AnonHugePages"):
            return int(line.split()[1]) >> 10
    return -1


print("maps allocated by", how, "- AnonHugePages", huge_mb(), "MB of", sum(s_.nbytes for s_ in segs) >> 20, "MB of maps", flush=True)
out = np.empty(n, np.int32)


def thr():
    try:
        kv = dict(line.split() for line in open("/sys/fs/cgroup/cpu.stat"))
        return int(kv["nr_periods"]), int(kv["nr_throttled"]), int(kv["throttled_usec"])
    except (OSError, KeyError, ValueError):
        return (0, 0, 0)


with pkg.Context(0) as ctx:
    if len(sys.argv) > 1 and int(sys.argv[1]) > 0:
        ctx.set_option("host_threads", int(sys.argv[1]))
    ctx.upload_positions(pos)
    print("host threads", ctx.host_threads(), "cpu.max", open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "?", flush=True)

    def steps(k, label):
        t_a = thr()
        rows = []
        for _ in range(k):
            t0 = time.perf_counter()
            ctx.vote_begin(150, 0, V)
            for v in range(V):
                ctx.vote_view(cams[v], segs[v])
            t1 = time.perf_counter()
            ctx.vote_finalize(out=out)
            t2 = time.perf_counter()
            rows.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3))
        t_b = thr()
        print(f"{label:34s} " + " ".join(f"{a:5.2f}+{b:4.2f}" for a, b in rows) +
              f"   | periods {t_b[0] - t_a[0]} throttled {t_b[1] - t_a[1]} ({(t_b[2] - t_a[2]) / 1e3:.1f} ms)", flush=True)

    def fence():
        ctx.synchronize()
        torch.cuda.synchronize()

    steps(6, "warm-up")
    if len(sys.argv) > 3 and sys.argv[3] == "short":
        steps(14, "back to back")
        steps(14, "back to back")
        print("AnonHugePages now", huge_mb(), "MB", flush=True)
        sys.exit(0)
    for rep in range(2):
        steps(14, "back to back")
        fence(); steps(4, "after a fence")
        fence(); ctx.profile(True); steps(4, "after fence + profile(True)")
        ctx.profile(False)
        fence(); ctx.vote_culled(reset=True); steps(4, "after fence + vote_culled(reset)")
        fence(); thr(); steps(4, "after fence + cpu.stat read")
        fence(); time.sleep(0.05); steps(4, "after fence + 50 ms pause")
        time.sleep(0.05); steps(4, "after a 50 ms pause, no fence")
        import gc
        fence(); gc.collect(); steps(4, "after fence + gc.collect()")
