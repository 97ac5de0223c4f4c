#!/usr/bin/env python3
"""GPU box: where the span of one labelling run goes - the hand-over of the views, then the TAIL (last DMA, descriptors,
vote kernel, labels to the host), with the A/Bs of DESIGN.md section 3: labels as bytes or int32 over the link, host maps
in compact or pool form, and the two timing-only ablations (option "ablate": 1 = pack but do not copy, 2 = copy but do not
pack; results invalid) that show which of the two the hand-over waits for.  Median of RUNS runs of configs[2] (3 M Gaussians x 200 views @1080p, seg-cell 4 maps:
the generator is the numpy one, the tail does not depend on the maps' content beyond the kernel's +9 %)."""
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
CELL = int(sys.argv[2]) if len(sys.argv) > 2 else 4      # 1 = pixel-accurate Voronoi boundaries (slow to generate on the host)
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
rows = []
for r in range(RUNS + 3):
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
print(f"hand-over {med[0]:.3f} ms  tail {med[1]:.3f} ms  span {med[2]:.3f} ms   (median of {RUNS}; min span {a[:,2].min():.3f}, tail min {a[:,1].min():.3f})")
for name, val in (("labels_u8", 0), ("labels_u8", 1), ("host_compact", 0), ("ablate", 1), ("ablate", 2), ("ablate", 0), ("host_compact", 1),
                  ("ablate", 1), ("ablate", 0), ("host_prefetch", 512), ("host_prefetch", -8192), ("host_prefetch", 8192), ("host_prefetch_burst", 0),
                  ("host_prefetch_burst", 1), ("host_threads", 15), ("host_threads", 8), ("host_threads", 24), ("host_threads", 16)):
    ctx.set_option(name, val)
    rows = []
    for r in range(RUNS + 3):
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
    print(f"{name}={val}: hand-over {med[0]:.3f} ms  tail {med[1]:.3f} ms  span {med[2]:.3f} ms   link bytes {ctx.vote_link_bytes()}  max span {a[:,2].max():.3f}")
