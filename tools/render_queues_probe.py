#!/usr/bin/env python3
"""GPU box: frames in flight (option render_frames) against the number of hardware queues the HIP runtime may use (env
GPU_MAX_HW_QUEUES, default 4; set by the caller before this process starts).  3 M splats / 1080p / SH 3."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("3d_gaussian_splatting_project_amd")
scene = pkg.scene
n, W, H = 3_000_000, 1920, 1080
seed = scene.BASE_SEED + 3
xyz = scene.make_positions(n, seed)
a = scene.make_splat_attributes(n, seed, sh_degree=3)
cams = scene.make_cameras(24, W, H, convention="c2w")
with pkg.Context(0) as c:
    c.upload_splats(xyz, a["scale"], a["rot"], a["opacity"], a["f_dc"])
    c.upload_sh(a["f_rest"], 3)
    for frames in (4, 5, 6, 4, 6):
        c.set_option("render_frames", frames)
        c.render_views(cams, W, H, to_host=False)
        t0 = time.perf_counter()
        for rep in range(5):
            c.render_views(cams, W, H, to_host=False)
        dt = (time.perf_counter() - t0) / (5 * len(cams))
        print(f"GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES', 'default')} frames {frames}: {dt * 1e3:.3f} ms/view = {1 / dt:.0f} views/s", flush=True)
