#!/usr/bin/env python3
"""Generate the viewer-side golden vectors (tests/golden/render_js.npz) by running the reference's
own worker code under node (tools/make_golden_js.js) on a seeded synthetic 3DGS PLY.

Pins, byte for byte: processPlyBuffer (gs.js:464-585: importance order, 32-byte rows, u8
quantisation), generateTexture (gs.js:286-357: 4*Sigma as truncated fp16), runSort (gs.js:417-462:
16-bit counting sort incl. the dropped max-depth splat) and the matrices of gs.js:66-123.
Build-container only.  Usage: python tests/golden/make_golden_render.py
"""
import base64
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
OUT = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)


def write_3dgs_ply(path, attrs, labels=None):
    """Minimal binary-little-endian 3DGS PLY writer (tools only; the product has its own)."""
    n = len(attrs["xyz"])
    names = ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"]
    n_rest = attrs["f_rest"].shape[1]
    names += [f"f_rest_{i}" for i in range(n_rest)] + ["opacity", "scale_0", "scale_1", "scale_2",
                                                        "rot_0", "rot_1", "rot_2", "rot_3"]
    cols = [attrs["xyz"], np.zeros((n, 3), np.float32), attrs["f_dc"], attrs["f_rest"], attrs["opacity"][:, None],
            attrs["scale"], attrs["rot"]]
    rows = np.concatenate([c.astype("<f4") for c in cols], axis=1)
    dt = [(k, "<f4") for k in names]
    hdr = "ply\nformat binary_little_endian 1.0\nelement vertex %d\n" % n
    hdr += "".join(f"property float {k}\n" for k in names)
    if labels is not None:
        hdr += "property int label\n"
        dt.append(("label", "<i4"))
    hdr += "end_header\n"
    rec = np.empty(n, dtype=dt)
    for i, k in enumerate(names):
        rec[k] = rows[:, i]
    if labels is not None:
        rec["label"] = labels
    with open(path, "wb") as f:
        f.write(hdr.encode("ascii"))
        f.write(rec.tobytes())


def main():
    import importlib
    scene = importlib.import_module("3d_gaussian_splatting_project_amd.scene")
    n = 1000
    seed = 0xC0FFEE40
    rng = np.random.default_rng(seed)
    xyz = scene.make_positions(n, seed)
    attrs = scene.make_splat_attributes(n, seed, sh_degree=3)
    attrs["xyz"] = xyz
    # make the quantisation edge cases appear: un-normalised quats, a w=1 quat, extreme opacities / colours
    attrs["rot"][:50] *= rng.uniform(0.1, 10.0, size=(50, 1)).astype(np.float32)
    attrs["rot"][50] = [1, 0, 0, 0]
    attrs["opacity"][51:55] = [-20.0, 20.0, 0.0, -0.0]
    attrs["f_dc"][55:59] = [[-5, 5, 0], [1.7724, -1.7724, 0.5], [10, -10, 2], [0.001, -0.001, 1e-9]]
    attrs["scale"][59] = [-12, -12, -12]            # tiny -> covariance underflows into fp16 subnormals
    attrs["scale"][60] = [2.5, 2.5, 2.5]            # huge -> fp16 overflow to inf (gs.js:269-272)
    attrs["scale"][61] = [-20, -20, -19]           # 4*sigma < 2^-46: floatToHalf's shift count wraps mod 32 (gs.js:262)
    labels = rng.integers(-1, 150, size=n).astype(np.int32)
    cams = scene.make_cameras(3, 640, 360, convention="c2w")
    real = json.load(open("/root/reference/Web_Viewer_Gaussians_Selection/cameras.json"))[0]
    cams.append(real)
    import oracle
    buf0, _ = oracle.pack_splats(xyz, attrs["scale"], attrs["rot"], attrs["opacity"], attrs["f_dc"])
    for c in cams:
        c["render_width"], c["render_height"] = (640, 360) if c["width"] == 640 else (1557, 1037)
        # clicks: a grid over the frame + clicks exactly ON projected centres (dist == 0, depth ties possible)
        W_, H_ = c["render_width"], c["render_height"]
        clicks = [[float(gx), float(gy)] for gx in np.linspace(5, W_ - 5, 12) for gy in np.linspace(5, H_ - 5, 8)]
        m = oracle.multiply4(oracle.proj_matrix(c["fx"], c["fy"], W_, H_), oracle.view_matrix(c))
        pos = buf0[:, :12].copy().view(np.float32).astype(np.float64)
        r = pos @ m.reshape(4, 4)[:3] + m.reshape(4, 4)[3]
        ok = np.nonzero(r[:, 3] > 0)[0][:40]
        for i in ok:
            clicks.append([float((r[i, 0] / r[i, 3] + 1) * 0.5 * W_), float((r[i, 1] / r[i, 3] + 1) * 0.5 * H_)])
            clicks.append([float((r[i, 0] / r[i, 3] + 1) * 0.5 * W_) + 9.999, float((r[i, 1] / r[i, 3] + 1) * 0.5 * H_)])
        c["clicks"] = clicks
    with tempfile.TemporaryDirectory() as d:
        ply, cj, oj = (os.path.join(d, f) for f in ("in.ply", "cams.json", "out.json"))
        write_3dgs_ply(ply, attrs, labels)
        json.dump(cams, open(cj, "w"))
        subprocess.check_call(["node", os.path.join(ROOT, "tools", "make_golden_js.js"), ply, cj, oj])
        out = json.load(open(oj))
    assert out["vertexCount"] == n
    buf = np.frombuffer(base64.b64decode(out["buffer"]), np.uint8).reshape(n, 32)
    tex = np.frombuffer(base64.b64decode(out["texdata"]), np.uint32)
    store = dict(xyz=xyz, scale=attrs["scale"], rot=attrs["rot"], opacity=attrs["opacity"], f_dc=attrs["f_dc"],
                 labels=labels, buffer=buf, texdata=tex[:8 * n].copy(), texwidth=out["texwidth"], texheight=out["texheight"],
                 cam_fx=np.array([c["fx"] for c in cams]), cam_fy=np.array([c["fy"] for c in cams]),
                 cam_R=np.array([c["rotation"] for c in cams]), cam_p=np.array([c["position"] for c in cams]),
                 cam_wh=np.array([[c["render_width"], c["render_height"]] for c in cams], np.int32),
                 view=np.array([c["view"] for c in out["cameras"]]), proj=np.array([c["proj"] for c in out["cameras"]]),
                 viewproj=np.array([c["viewProj"] for c in out["cameras"]]),
                 hit_xy=np.array([c["clicks"] for c in cams[:3]]), hit_labels=np.array([c["hits"] for c in out["cameras"][:3]], np.int32),
                 hit_xy_real=np.array(cams[3]["clicks"]), hit_labels_real=np.array(out["cameras"][3]["hits"], np.int32),
                 depth_index=np.stack([np.frombuffer(base64.b64decode(c["depthIndex"]), np.uint32) for c in out["cameras"]]))
    np.savez_compressed(os.path.join(OUT, "render_js.npz"), **store)
    di = store["depth_index"]
    print("render_js.npz", os.path.getsize(os.path.join(OUT, "render_js.npz")), "bytes; unique per camera:",
          [len(np.unique(r)) for r in di], "trailing zeros:", [int((r[-3:] == 0).sum()) for r in di])


if __name__ == "__main__":
    main()
