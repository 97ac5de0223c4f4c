#!/usr/bin/env python3
"""GPU box: where the time of one labelling run (host int32 maps -> labels on the host) goes."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
pkg = importlib.import_module("3d_gaussian_splatting_project_amd")
sys.argv = [sys.argv[0]]
import bench
scene = pkg.scene
n, V, W, H = 3_000_000, 200, 1920, 1080
pos = scene.make_positions(n, scene.BASE_SEED + 3)
cams = [pkg.Camera.from_dict(c) for c in scene.make_cameras(V, W, H, convention="w2c")]
segs = bench.make_segmaps(scene, torch, 0, H, W, 150, [3000 + v for v in range(V)], 1)
ctx = pkg.Context(0)
ctx.upload_positions(pos)
out = np.empty(n, np.int32)
for rep in range(6):
    t0 = time.perf_counter()
    ctx.vote_begin(150, 0, V)
    t1 = time.perf_counter()
    for v in range(V):
        ctx.vote_view(cams[v], segs[v])
    t2 = time.perf_counter()
    ctx.synchronize()
    t3 = time.perf_counter()
    ctx.vote_finalize(out=out)
    t4 = time.perf_counter()
    ctx.vote_rewind(); ctx.vote_finalize(to_host=False)
    t5 = time.perf_counter()
    ctx.vote_rewind(); ctx.vote_finalize(out=out)
    t6 = time.perf_counter()
    print(f"begin {1e3*(t1-t0):.3f}  submit loop {1e3*(t2-t1):.3f}  drain DMA {1e3*(t3-t2):.3f}  finalize(to host) {1e3*(t4-t3):.3f}  | kernel only {1e3*(t5-t4):.3f}  kernel+D2H {1e3*(t6-t5):.3f} ms")
