#!/usr/bin/env python3
"""GPU box: the floor of a frame - a scene so small that every kernel is empty (1000 splats, 64x64 pixels): ms per view of
gsx_render_views by frames in flight = launches, host threads, end-of-frame synchronisation and nothing else."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("3d_gaussian_splatting_project_amd")
scene = pkg.scene
n, W, H = 1000, 64, 64
xyz = scene.make_positions(n, 5)
a = scene.make_splat_attributes(n, 5, sh_degree=3)
cams = scene.make_cameras(48, W, H, convention="c2w")
with pkg.Context(0) as c:
    c.upload_splats(xyz, a["scale"], a["rot"], a["opacity"], a["f_dc"])
    c.upload_sh(a["f_rest"], 3)
    for F in (1, 2, 4, 1, 4):
        c.set_option("render_frames", F)
        c.render_views(cams, W, H, to_host=False)
        t0 = time.perf_counter()
        for rep in range(5):
            c.render_views(cams, W, H, to_host=False)
        dt = (time.perf_counter() - t0) / (5 * len(cams))
        print(f"render_frames={F}: {dt * 1e3:.3f} ms/view (empty kernels: the launch + synchronisation floor)", flush=True)
