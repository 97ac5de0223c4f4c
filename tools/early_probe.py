#!/usr/bin/env python3
"""GPU box: the early vote (option early_vote) against the one-piece vote on configs[2] (3 M Gaussians
x 200 views @1080p): hand-over, tail and span of a run for a sweep of the split point (option early_vote_at, permille of
the announced views), with the HIP-event times of the kernels involved.  Labels of every variant are compared with the
one-piece vote's.  argv: runs per variant, seg cell (1 = pixel-accurate boundaries, generated on the GPU like bench.py)."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# the timing-only option "ablate" exists in the experiments build only (`make -C 3d_gaussian_splatting_project_amd/csrc experiments`)
_exp = os.path.join(ROOT, "3d_gaussian_splatting_project_amd", "libgsx_experiments.so")
if os.path.exists(_exp):
    os.environ.setdefault("GSX_LIBRARY", _exp)
else:
    raise SystemExit("build libgsx_experiments.so first: make -C 3d_gaussian_splatting_project_amd/csrc experiments")
pkg = importlib.import_module("3d_gaussian_splatting_project_amd")
scene = pkg.scene
import torch
RUNS = int(sys.argv[1]) if len(sys.argv) > 1 else 30
CELL = int(sys.argv[2]) if len(sys.argv) > 2 else 4
torch.cuda.set_device(0)
pkg.bind_to_gpu_numa_node(0)
n, V, W, H = 3_000_000, 200, 1920, 1080
pos = scene.make_positions(n, scene.BASE_SEED + 3)
cams = [pkg.Camera.from_dict(c) for c in scene.make_cameras(V, W, H, convention="w2c")]
base = [scene.make_segmap(H, W, 150, 3000 + v, cell=CELL) for v in range(8)]
segs = [base[v % 8].copy() for v in range(V)]
ctx = pkg.Context(0)
ctx.upload_positions(pos)
out = np.empty(n, np.int32)


def measure(tag):
    rows = []
    ctx.profile(True)
    for r in range(RUNS + 3):
        if r == 3:
            ctx.profile(False)
            ctx.profile(True)   # forget the warm-up launches
        t0 = time.perf_counter()
        ctx.vote_begin(150, 0, V)
        for v in range(V):
            ctx.vote_view(cams[v], segs[v])
        t1 = time.perf_counter()
        ctx.vote_finalize(out=out)
        t2 = time.perf_counter()
        if r >= 3:
            rows.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t2 - t0) * 1e3))
    a = np.array(rows)
    med = np.median(a, axis=0)
    ks = []
    for k in ("vote_fused_labels", "vote_early_planes", "vote_fused_final", "vote_early_record", "vote_fused_replay", "seg_expand"):
        if k in ctx.profile_names():
            cnt, ms = ctx.profile_get(k)
            if cnt:
                ks.append(f"{k} {ms / cnt:.3f} ms x{cnt}")
    ctx.profile(False)
    print(f"{tag}: hand-over {med[0]:.3f}  tail {med[1]:.3f}  span {med[2]:.3f} ms  (min span {a[:,2].min():.3f}, max {a[:,2].max():.3f})   {'; '.join(ks)}", flush=True)
    return out.copy()


ctx.set_option("early_vote", 0)
ref = measure("one piece      ")
for at in (0, 700, 0):   # 0: split point from the hand-over rate
    ctx.set_option("early_vote", 1)
    ctx.set_option("early_vote_at", at)
    got = measure(f"early at {at:4d}")
    print(f"    early views {ctx.vote_early_views()}", flush=True)
    assert np.array_equal(got, ref), at
ctx.set_option("early_replay", 1)      # record + replay instead of planes + fold
for at in (700, 800, 850, 880, 920):
    ctx.set_option("early_vote_at", at)
    got = measure(f"replay at {at:4d}")
    assert np.array_equal(got, ref), at
ctx.set_option("ablate", 32)
measure("replay at  920, no views behind the early ones")
ctx.set_option("ablate", 0)
ctx.set_option("early_replay", 0)
ctx.set_option("early_vote_at", 700)
for ab in (32, 0):     # timing only, see vote.hip early_vote_finish
    ctx.set_option("ablate", ab)
    measure(f"at 700 ablate {ab:4d}")
# the last stage alone, back to back (the run's early planes are still there): what the kernel costs when nothing else runs
ctx.profile(True)
for r in range(10):
    ctx.vote_finalize(out=out)
cnt, ms = ctx.profile_get("vote_fused_final")
print(f"last stage back to back: vote_fused_final {ms / cnt:.3f} ms x{cnt}", flush=True)
ctx.profile(False)
assert np.array_equal(out, ref)
ctx.set_option("early_vote", 0)
measure("one piece again")
print("labels of every variant equal the one-piece vote's")
