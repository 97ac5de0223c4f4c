#!/usr/bin/env python3
"""Time the reference's own assign_labels loop (deep_learning_segmentation.py:255-308, imported unmodified,
model loader bypassed as in tools/make_golden.py) on a subsample of the benchmark scene.  Build container
only; writes profiles/r01/cpu_reference_python.json.  BASELINE.md section 3, item 1."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import importlib

import make_golden as mg

scene = importlib.import_module("3d_gaussian_splatting_project_amd.scene")
import oracle

if __name__ == "__main__":
    dls = mg.import_reference()
    n, V, W, H = 100_000, 4, 1920, 1080
    pos = scene.make_positions(3_000_000, scene.BASE_SEED + 3)[:n]
    cams = scene.make_cameras(200, W, H, convention="w2c")[:V]
    for c in cams:
        c["img_name"] = c["img_name"]
    segs = [scene.make_segmap(H, W, 150, 3000 + v) for v in range(V)]
    t0 = time.perf_counter()
    labels = mg.run_assign(dls, pos, cams, [(W, H)] * V, [True] * V, segs)
    dt = time.perf_counter() - t0
    ok = bool(np.array_equal(labels, oracle.assign_labels(pos, cams, segs, [(W, H)] * V, threads=1)))
    out = {"what": "reference assign_labels (Python double loop), imported unmodified, 1 thread, build container",
           "gaussians": n, "views": V, "seconds": round(dt, 2), "gaussian_views_per_s": round(n * V / dt, 1),
           "labels_equal_oracle": ok, "extrapolated_seconds_for_3M_x_200": round(3_000_000 * 200 / (n * V / dt), 0)}
    print(json.dumps(out))
    json.dump(out, open(os.path.join(ROOT, "profiles", "r01", "cpu_reference_python.json"), "w"), indent=1)
