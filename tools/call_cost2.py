#!/usr/bin/env python3
"""GPU box: fixed cost of one gsx_vote_view call (Python + C entry + fork-join + DMA bookkeeping), from maps too small for
their bytes to matter: 64x8 (one band: no fork-join), 64x16 (two bands), 64x128 (16 bands = one per thread), 64x1080."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("3d_gaussian_splatting_project_amd")
scene = pkg.scene
import torch
torch.cuda.set_device(0)
pkg.bind_to_gpu_numa_node(0)
ctx = pkg.Context(0)
ctx.upload_positions(np.zeros((1000, 3), np.float32))
V = 200
for (w, h) in ((64, 8), (64, 16), (64, 128), (64, 1080), (1920, 1080)):
    cams = [pkg.Camera.from_dict(c) for c in scene.make_cameras(V, w, h, convention="w2c")]
    segs = [np.random.default_rng(v).integers(-1, 150, size=(h, w), dtype=np.int32) for v in range(8)]
    segs = [segs[v % 8].copy() for v in range(V)]
    out = np.empty(1000, np.int32)
    ts = []
    for r in range(23):
        ctx.vote_begin(150, 0, V)
        t0 = time.perf_counter()
        for v in range(V):
            ctx.vote_view(cams[v], segs[v])
        t1 = time.perf_counter()
        ctx.vote_finalize(out=out)
        if r >= 3:
            ts.append((t1 - t0) / V * 1e6)
    print(f"{w}x{h}: {np.median(ts):.2f} us per vote_view call (min {min(ts):.2f})")
