#!/usr/bin/env python3
"""GPU box: views/s of the rasterizer with S contexts (one HIP stream each) rendering disjoint views from S host threads."""
import importlib, os, sys, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("3d_gaussian_splatting_project_amd")
scene = pkg.scene
n, W, H = 3_000_000, 1920, 1080
seed = scene.BASE_SEED + 3
xyz = scene.make_positions(n, seed)
a = scene.make_splat_attributes(n, seed, sh_degree=3)
cams = scene.make_cameras(8, W, H, convention="c2w")
for S in (1, 2, 3):
    ctxs = []
    for s in range(S):
        c = pkg.Context(0)
        c.upload_splats(xyz, a["scale"], a["rot"], a["opacity"], a["f_dc"])
        c.upload_sh(a["f_rest"], 3)
        c.render_view(cams[0], W, H, to_host=False)
        ctxs.append(c)
    reps = 6
    def work(c, s):
        for r in range(reps):
            for k, cam in enumerate(cams):
                if k % S == s:
                    c.render_view(cam, W, H, to_host=False)
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(c, s)) for s, c in enumerate(ctxs)]
    for t in th: t.start()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print(f"{S} stream(s): {reps*len(cams)/dt:.0f} views/s")
    for c in ctxs: c.close()
