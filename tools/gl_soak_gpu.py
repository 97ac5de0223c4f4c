#!/usr/bin/env python3
"""GPU box: the HIP rasterizer against frames of the reference's own shaders (Mesa llvmpipe) on an UNCOMMITTED fixture made in the
build container by `tests/golden/make_golden_gl.py --big K gl_soak_fixture.npz` (it travels with the gpurun snapshot; the reference
itself cannot).  Per frame: max |HIP - GL| outside the pixels that differ by more than 1e-4, and how many of those there are (fragments
on the discard threshold, which GL's 1/256-pixel vertex snapping decides: DESIGN.md section 2).
usage: tools/gl_soak_gpu.py gl_soak_fixture.npz [option=value ...]   (gsx_set_option, e.g. render_bin32=0 render_compact=0)"""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("3d_gaussian_splatting_project_amd")
z = np.load(sys.argv[1])
worst, flips, covered, frames, worst_flip = 0.0, 0, 0, 0, 0.0
with pkg.Context(0) as ctx:
    for kv in sys.argv[2:]:
        ctx.set_option(kv.split("=")[0], int(kv.split("=")[1]))
    last = None
    for i in (int(k) for k in z["calls"]):
        j = int(z[f"c{i}_scene"])
        fx, fy, W, H = (float(v) for v in z[f"c{i}_cam"])
        W, H = int(W), int(H)
        if j != last:
            ctx.upload_splats(z[f"s{j}_xyz"], z[f"s{j}_scale"], z[f"s{j}_rot"], z[f"s{j}_opacity"], z[f"s{j}_f_dc"])
            last = j
        cam = {"fx": fx, "fy": fy, "width": W, "height": H, "rotation": z[f"c{i}_R"].tolist(), "position": z[f"c{i}_p"].tolist()}
        img = ctx.render_view(cam, W, H)
        g = z[f"c{i}_frame"]
        d = np.abs(img.astype(np.float64) - g).max(axis=2)
        over = d > 1e-4
        frames += 1
        flips += int(over.sum())
        covered += int((g[..., 3] > 0).sum())
        worst = max(worst, float(d[~over].max()))
        worst_flip = max(worst_flip, float(d.max()))
        assert d.max() <= np.exp(-4.0) + 1e-4, (i, float(d.max()))
print(f"options {sys.argv[2:]}: {frames} frames, {covered} covered pixels: max |HIP - GL| = {worst:.3e} outside {flips} pixels (1 in {covered // max(flips, 1)}) that differ by "
      f"more than 1e-4 (largest {worst_flip:.3e} <= e^-4 = one fragment at the discard threshold); GL = {z['gl']}")
