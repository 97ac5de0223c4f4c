#!/usr/bin/env python3
"""Generate the labeler (vote path) golden vectors under tests/golden/.

Runs ONLY in the build container, where /root/reference exists.  It imports the
reference's own deep_learning_segmentation.py (missing third-party modules are
replaced by inert empty modules, the model loader and segment_image are
replaced by a seeded synthetic segmentation) and records what the reference's
own arithmetic returns:

  G1  project_gaussian   (deep_learning_segmentation.py:43-82)
  G2  assign_labels      (deep_learning_segmentation.py:241-308)

Only inputs and outputs are stored; no reference source travels.
Usage:  python tools/make_golden.py            (rewrites tests/golden/vote_*.npz)
"""
import json
import os
import sys
import tempfile
import types

import numpy as np

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")


def import_reference():
    for name, attrs in (("plyfile", ("PlyData", "PlyElement")), ("ultralytics", ("YOLO",)), ("cv2", ())):
        if name not in sys.modules:
            try:
                __import__(name)
            except ImportError:
                m = types.ModuleType(name)
                for a in attrs:
                    setattr(m, a, object)
                sys.modules[name] = m
    sys.path.insert(0, REF)
    import deep_learning_segmentation as dls
    return dls


# ----------------------------------------------------------------------------------------------
# scene helpers (plain numpy, seeded)
# ----------------------------------------------------------------------------------------------
def look_at_camera(rng, img_name, width, height, radius=6.0, fx=None, fy=None):
    """Camera whose labeler-convention matrix R (pos_cam = R @ (x - p)) looks at the origin."""
    d = rng.normal(size=3)
    d /= np.linalg.norm(d)
    p = d * radius
    zc = -d                                    # camera z axis (towards the origin)
    up = np.array([0.0, 1.0, 0.0]) + 0.1 * rng.normal(size=3)
    xc = np.cross(up, zc)
    xc /= np.linalg.norm(xc)
    yc = np.cross(zc, xc)
    R = np.stack([xc, yc, zc])                 # rows = camera axes
    return {
        "id": 0, "img_name": img_name, "width": int(width), "height": int(height),
        "position": [float(v) for v in p], "rotation": [[float(v) for v in row] for row in R],
        "fy": float(fy if fy is not None else 0.9 * width),
        "fx": float(fx if fx is not None else 0.9 * width),
    }


def blocky_segmap(rng, h, w, n_classes, cell=12):
    """Piecewise-constant int32 map with values in [-1, n_classes-1]."""
    gh, gw = (h + cell - 1) // cell, (w + cell - 1) // cell
    grid = rng.integers(-1, n_classes, size=(gh, gw), dtype=np.int32)
    return np.kron(grid, np.ones((cell, cell), dtype=np.int32))[:h, :w].copy()


def cams_to_arrays(cams):
    return dict(
        cam_fx=np.array([c["fx"] for c in cams], dtype=np.float64),
        cam_fy=np.array([c["fy"] for c in cams], dtype=np.float64),
        cam_wh=np.array([[c["width"], c["height"]] for c in cams], dtype=np.int32),
        cam_R=np.array([c["rotation"] for c in cams], dtype=np.float64),
        cam_p=np.array([c["position"] for c in cams], dtype=np.float64),
    )


def make_gaussians(positions_f32):
    g = np.zeros(len(positions_f32),
                 dtype=[('position', np.float32, 3), ('scale', np.float32, 3), ('rotation', np.float32, 4)])
    g['position'] = positions_f32
    return g


# ----------------------------------------------------------------------------------------------
# G1: project_gaussian
# ----------------------------------------------------------------------------------------------
def golden_project(dls):
    rng = np.random.default_rng(0xC0FFEE01)
    real = json.load(open(os.path.join(REF, "Web_Viewer_Gaussians_Selection", "cameras.json")))
    cams = [real[0], real[57], real[310]]
    cams += [look_at_camera(rng, f"syn{i}", w, h) for i, (w, h) in
             enumerate([(1280, 720), (1920, 1080), (640, 480), (3840, 2160), (333, 777)])]
    n = 50000                                  # 50 k positions x 8 cameras
    pos = rng.normal(scale=2.5, size=(n, 3))
    # a third of the points are placed in front of the real cameras (x = R^T pc + p  =>  R @ (x-p) = pc approx)
    for k, c in enumerate(cams[:3]):
        R = np.array(c["rotation"]); p = np.array(c["position"])
        m = 5000
        pc = np.stack([rng.uniform(-3, 3, m), rng.uniform(-2, 2, m), rng.uniform(0.05, 9, m)], 1)
        pos[k * m:(k + 1) * m] = (np.linalg.inv(R) @ pc.T).T + p
    pos = pos.astype(np.float32)
    # edge cases appended: exactly on the camera plane, behind, huge, tiny depth, nan/inf
    edge = np.array([
        cams[3]["position"],                                  # pos_cam == 0  -> z <= 0 -> None
        [0, 0, 0], [1e30, -1e30, 1e30], [1e-30, 1e-30, 1e-30],
        [np.nan, 0, 0], [np.inf, 0, 0], [0, -np.inf, 1], [3.4e38, 3.4e38, 3.4e38],
    ], dtype=np.float32)
    # points engineered to land within 1e-9..1 px of every border of synthetic camera 3 (1280x720)
    c = cams[3]
    R = np.array(c["rotation"]); p = np.array(c["position"])
    border = []
    for px, py in [(-0.5, 10), (-1e-7, 10), (0.0, 10), (1279.9999, 10), (1280.0, 10), (1280.0001, 10),
                   (10, -0.3), (10, 0.0), (10, 719.99999), (10, 720.0), (640.5, 360.5), (0.999999, 0.999999)]:
        z = 4.0
        pc = np.array([(px - c["width"] / 2) * z / c["fx"], (py - c["height"] / 2) * z / c["fy"], z])
        border.append(np.linalg.inv(R) @ pc + p)
    pos = np.concatenate([pos, edge, np.array(border, dtype=np.float32)])
    g = make_gaussians(pos)
    V = len(cams)
    xs = np.full((V, len(pos)), -1, np.int32); ys = np.full((V, len(pos)), -1, np.int32)
    for v, cam in enumerate(cams):
        for i in range(len(pos)):
            with np.errstate(all="ignore"):
                r = dls.project_gaussian(g[i]['position'], cam)
            if r is not None:
                xs[v, i], ys[v, i] = r
    print("G1 visible fraction per camera:", (xs >= 0).mean(1).round(3))
    assert xs.max() < 32768 and ys.max() < 32768            # int16 keeps the fixture under a megabyte
    np.savez_compressed(os.path.join(OUT, "vote_project.npz"), positions=pos, x=xs.astype(np.int16), y=ys.astype(np.int16),
                        **cams_to_arrays(cams))


# ----------------------------------------------------------------------------------------------
# G2: assign_labels
# ----------------------------------------------------------------------------------------------
class _Dummy:
    def to(self, *_a, **_k):
        return self


def run_assign(dls, positions, cams, img_sizes, present, segmaps):
    """Drive the reference's assign_labels with synthetic 'segment_image' output."""
    from PIL import Image
    by_name = {}
    with tempfile.TemporaryDirectory() as d:
        for cam, (iw, ih), ok, seg in zip(cams, img_sizes, present, segmaps):
            by_name[cam["img_name"]] = seg
            if ok:
                Image.new("L", (int(iw), int(ih))).save(os.path.join(d, cam["img_name"] + ".png"))
        dls.initialize_model = lambda model_type, device: (None, _Dummy())
        dls.segment_image = lambda image_path, output_dir, processor, model, device, model_type: \
            by_name[os.path.splitext(os.path.basename(image_path))[0]]
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()), np.errstate(all="ignore"):
            return dls.assign_labels(make_gaussians(positions), cams, d, d, 'segformer')


def pack_case(prefix, store, positions, cams, img_sizes, present, segmaps, labels):
    store[prefix + "positions"] = positions
    for k, v in cams_to_arrays(cams).items():
        store[prefix + k] = v
    store[prefix + "img_wh"] = np.array(img_sizes, dtype=np.int32)
    store[prefix + "present"] = np.array(present, dtype=np.bool_)
    store[prefix + "seg_shapes"] = np.array([s.shape for s in segmaps], dtype=np.int32)
    store[prefix + "seg_flat"] = np.concatenate([s.astype(np.int16).ravel() for s in segmaps])
    store[prefix + "labels"] = labels.astype(np.int32)


def golden_assign(dls):
    store = {}
    names = []
    # ---- random scenes, V in {1, 3, 8}, 150 classes + (-1) pixels ---------------------------
    for V in (1, 3, 8):
        rng = np.random.default_rng(0xC0FFEE10 + V)
        n = 20000                              # N = 20 k
        w, h = 320, 180
        cams = [look_at_camera(rng, f"v{V}_{i:03d}", w, h) for i in range(V)]
        pos = rng.normal(scale=1.6, size=(n, 3)).astype(np.float32)
        segs = [blocky_segmap(rng, h, w, 150) for _ in range(V)]
        sizes = [(w, h)] * V
        present = [True] * V
        labels = run_assign(dls, pos, cams, sizes, present, segs)
        name = f"rand{V}_"
        pack_case(name, store, pos, cams, sizes, present, segs, labels)
        names.append(name)
        print(name, "labelled(!=-1):", (labels != -1).mean().round(3), "unique:", len(np.unique(labels)))

    # ---- few coarse classes: many exact ties, pins first-inserted-wins (dls.py:303) -----------
    rng = np.random.default_rng(0xC0FFEE20)
    V, n, w, h = 6, 20000, 200, 120
    cams = [look_at_camera(rng, f"tie_{i:03d}", w, h, radius=5.0) for i in range(V)]
    pos = rng.normal(scale=1.0, size=(n, 3)).astype(np.float32)
    segs = [blocky_segmap(rng, h, w, 3, cell=7) for _ in range(V)]          # labels in {-1,0,1,2}
    sizes = [(w, h)] * V
    present = [True] * V
    labels = run_assign(dls, pos, cams, sizes, present, segs)
    pack_case("ties_", store, pos, cams, sizes, present, segs, labels)
    names.append("ties_")

    # ---- missing PNG (camera skipped), seg map smaller than image, cam size != image size ----
    rng = np.random.default_rng(0xC0FFEE30)
    V, n = 5, 20000
    cams = [look_at_camera(rng, f"mix_{i:03d}", 320, 180) for i in range(V)]
    cams[2]["width"], cams[2]["height"] = 300, 200          # camera intrinsics size != image size
    cams[4]["fx"] = 250                                      # an int focal, as JSON may hold
    pos = rng.normal(scale=1.6, size=(n, 3)).astype(np.float32)
    sizes = [(320, 180), (320, 180), (320, 180), (640, 360), (333, 187)]
    seg_hw = [(180, 320), (90, 160), (180, 320), (360, 640), (100, 100)]  # (h, w); view 1: half res; view 4: odd ratio
    segs = [blocky_segmap(rng, hh, ww, 150, cell=9) for hh, ww in seg_hw]
    present = [True, True, True, False, True]                # view 3's PNG is missing -> skipped
    labels = run_assign(dls, pos, cams, sizes, present, segs)
    pack_case("mix_", store, pos, cams, sizes, present, segs, labels)
    names.append("mix_")

    store["cases"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "vote_assign.npz"), **store)


if __name__ == "__main__":
    os.makedirs(OUT, exist_ok=True)
    dls = import_reference()
    golden_project(dls)
    golden_assign(dls)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))
