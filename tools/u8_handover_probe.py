#!/usr/bin/env python3
"""GPU box: the hand-over of 200 1080p maps (first vote_view -> last DMA done, no vote) as int32 maps and as uint8 class images
(GSX_SEG_U8_LABELS, a quarter of the bytes), and the whole run (hand-over + vote + labels on the host) for both; the benchmark's
pixel-accurate maps with their -1 pixels set to class 0 so that both forms hold the same labels."""
import importlib, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
pkg = importlib.import_module("3d_gaussian_splatting_project_amd")
scene, Camera = pkg.scene, pkg.Camera
n, V, W, H = 3_000_000, 200, 1920, 1080
pkg.bind_to_gpu_numa_node(0)
pos = scene.make_positions(n, scene.BASE_SEED + 3)
cams = [Camera.from_dict(c) for c in scene.make_cameras(V, W, H, convention="w2c")]
segs = bench.make_segmaps(scene, torch, 0, H, W, 150, [3000 + v for v in range(V)], 1)
maps = {"int32": [np.where(s < 0, 0, s).astype(np.int32) for s in segs]}
maps["uint8"] = [m.astype(np.uint8) for m in maps["int32"]]
del segs
labels = {}
with pkg.Context(0) as ctx:
    ctx.upload_positions(pos)
    out = np.empty(n, np.int32)
    for rep in range(2):
        for kind in ("int32", "uint8"):
            mm = maps[kind]
            def run(finalize):
                ctx.vote_begin(150, 0, V)
                t0 = time.perf_counter()
                for v in range(V):
                    ctx.vote_view(cams[v], mm[v])
                t_submit = time.perf_counter() - t0
                if finalize:
                    ctx.vote_finalize(out=out)
                else:
                    ctx.synchronize()
                t = time.perf_counter() - t0
                if not finalize:
                    ctx.vote_finalize(out=out)
                return t_submit, t
            for _ in range(3):
                run(True)
            a = [run(False) for _ in range(5)]
            b = [run(True) for _ in range(8)]
            labels[kind] = out.copy()
            print(f"{kind}: hand-over {np.median([x[1] for x in a]) * 1e3:.3f} ms (submit {np.median([x[0] for x in a]) * 1e3:.3f}), whole run "
                  f"{np.median([x[1] for x in b]) * 1e3:.3f} ms (last view submitted at {np.median([x[0] for x in b]) * 1e3:.3f}), early views {ctx.vote_early_views()}, "
                  f"link bytes {ctx.vote_link_bytes()}", flush=True)
print("labels equal:", bool(np.array_equal(labels["int32"], labels["uint8"])))
