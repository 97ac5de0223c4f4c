import importlib, os, sys, time
import numpy as np
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
pkg = importlib.import_module("3d_gaussian_splatting_project_amd")
scene = pkg.scene
import torch
torch.cuda.set_device(0)
if sys.argv[1] == "bind":
    pkg.bind_to_gpu_numa_node(0)
n, W, H = 3_000_000, 1920, 1080
seed = scene.BASE_SEED + 3
xyz = scene.make_positions(n, seed)
a = scene.make_splat_attributes(n, seed, sh_degree=3)
cams24 = scene.make_cameras(24, W, H, convention="c2w")
cams4 = scene.make_cameras(8, W, H, convention="c2w")[:4] * 6
with pkg.Context(0) as c:
    c.upload_splats(xyz, a["scale"], a["rot"], a["opacity"], a["f_dc"])
    c.upload_sh(a["f_rest"], 3)
    for name, cams in (("24 cameras", cams24), ("4 cameras x 6", cams4)):
        for F in (1, 4):
            c.set_option("render_frames", F)
            c.render_views(cams, W, H, to_host=False)
            ts = []
            for rep in range(4):
                t0 = time.perf_counter()
                c.render_views(cams, W, H, to_host=False)
                ts.append((time.perf_counter() - t0) / len(cams))
            print(f"{sys.argv[1]:6s} {name:14s} render_frames={F}: best {1 / min(ts):.0f} views/s, first call {1 / ts[0]:.0f}, mean {len(ts) / sum(ts):.0f}", flush=True)
