#!/usr/bin/env python3
"""Time the k-means labeler on the GPU box: 3 M rows, k = 10, the CLI's 10 iterations (3D_clustering/k_means.py:209).
Prints one JSON line; the oracle (numpy) is timed on a 300 k-row sample for scale."""
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
labeler = importlib.import_module("3d_gaussian_splatting_project_amd.labeler")

n, k, iters = 3_000_000, 10, 10
rng = np.random.default_rng(1)
centres = rng.normal(size=(14, 6)) * np.array([5, 5, 5, 1, 1, 1])
data = (centres[rng.integers(0, 14, size=n)] + rng.normal(size=(n, 6)) * np.array([0.9, 0.9, 0.9, 0.3, 0.3, 0.3])).astype(np.float32)
pts, col = np.ascontiguousarray(data[:, :3]), np.ascontiguousarray(data[:, 3:])
init = rng.choice(n, k, replace=False)
with labeler.Context(0) as c:
    c.kmeans(pts[:100_000], col[:100_000], k, init % 100_000, max_iter=2)      # warm-up
    c.profile(True)
    t0 = time.perf_counter()
    cent, labels, it, conv = c.kmeans(pts, col, k, init, max_iter=iters)
    dt = time.perf_counter() - t0
    prof = {name: c.profile_get(name) for name in ("kmeans_assign", "kmeans_sum", "radix_scatter", "radix_hist", "radix_rowscan")}
from oracle import kmeans_oracle
m = 300_000
t0 = time.perf_counter()
want = kmeans_oracle.k_means_with_color(pts[:m], k, col[:m], init % m, max_iter=iters)
dto = time.perf_counter() - t0
got = None
with labeler.Context(0) as c:
    got = c.kmeans(pts[:m], col[:m], k, init % m, max_iter=iters)
print(json.dumps({"rows": n, "k": k, "iterations": it, "seconds_total": round(dt, 4),
                  "rows_iterations_per_s": round(n * it / dt, 1),
                  "kernel_ms": {a: (b[0], round(b[1], 3)) for a, b in prof.items()},
                  "oracle_numpy_rows_iterations_per_s": round(m * want[2] / dto, 1),
                  "sample_equals_oracle": bool(np.array_equal(got[1], want[1]) and np.array_equal(got[0], want[0]))}))
