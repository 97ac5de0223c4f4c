#!/usr/bin/env python3
"""GPU box: views/s of gsx_render_views by the number of frames in flight (option render_frames), bench scene
(3 M splats / 1080p / SH 3), frames left on the device; every setting's last frame is compared with the one-frame-at-a-time one.
argv: [c1] configs[1] sizes; [early] an early-vote labelling run on the SAME context first (as in bench.py's process: it creates
the context's second stream - are the frames' streams then more than the hardware queues?); [multi0] option render_multi_pre = 0."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("3d_gaussian_splatting_project_amd")
scene = pkg.scene
n, W, H = (500_000, 1280, 720) if "c1" in sys.argv[1:] else (3_000_000, 1920, 1080)
seed = scene.BASE_SEED + 3
xyz = scene.make_positions(n, seed)
a = scene.make_splat_attributes(n, seed, sh_degree=3)
cams = scene.make_cameras(24, W, H, convention="c2w")
with pkg.Context(0) as c:
    if "early" in sys.argv[1:]:
        m, V = 400_000, 40
        pos, lc, segs = scene.make_scene(m, V, 320, 180, config_id=71, convention="w2c")
        c.set_option("early_vote", 2)
        c.upload_positions(pos)
        c.vote_begin(150, 0, V)
        for cam, seg in zip(lc, segs):
            c.vote_view(cam, seg)
        c.vote_finalize()
        print("labelling run with an early stage done:", c.vote_early_views(), "early views", flush=True)
    if os.environ.get("GSX_PROBE_SHARE") == "0":      # the first extra frame on a stream of its own (round 2) instead of the context's second stream
        c.set_option("render_share_stream", 0)
    if "multi0" in sys.argv[1:]:
        c.set_option("render_multi_pre", 0)
    c.upload_splats(xyz, a["scale"], a["rot"], a["opacity"], a["f_dc"])
    c.upload_sh(a["f_rest"], 3)
    ref = None
    for F in (1, 2, 3, 4, 5, 4):
        c.set_option("render_frames", F)
        c.render_views(cams, W, H, to_host=False)
        t0 = time.perf_counter()
        for rep in range(3):
            c.render_views(cams, W, H, to_host=False)
        dt = (time.perf_counter() - t0) / (3 * len(cams))
        frames = c.render_views(cams[:5], W, H)
        if ref is None:
            ref = frames
        same = all(np.array_equal(x, y) for x, y in zip(ref, frames))
        print(f"render_frames={F}: {dt * 1e3:.3f} ms/view = {1 / dt:.0f} views/s   frames identical to one at a time: {same}", flush=True)
