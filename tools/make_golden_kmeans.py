#!/usr/bin/env python3
"""Generate the k-means labeler golden vectors (tests/golden/kmeans.npz).

Runs ONLY in the build container, where /root/reference exists.  It imports the reference's own
3D_clustering/k_means.py (plyfile, which is not installed, is replaced by an inert empty module; scipy is the real
one) and records what `k_means_with_color` (k_means.py:107-151) returns on seeded inputs.  The reference draws its
initial centroids with an unseeded `np.random.choice` (k_means.py:111); the generator wraps that one call to RECORD
the indices it drew, so that the same start can be injected into the oracle and the GPU path.
Only inputs and outputs are stored; no reference source travels.
Usage:  python tools/make_golden_kmeans.py
"""
import contextlib
import importlib.util
import io
import os
import sys
import types

import numpy as np

REF = "/root/reference/3D_clustering/k_means.py"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "kmeans.npz")


def import_reference():
    if "plyfile" not in sys.modules:
        try:
            __import__("plyfile")
        except ImportError:
            m = types.ModuleType("plyfile")
            m.PlyData = m.PlyElement = object
            sys.modules["plyfile"] = m
    spec = importlib.util.spec_from_file_location("ref_k_means", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    km = import_reference()
    rng = np.random.default_rng(20241218)
    cases = {}
    specs = [  # name, N, k, max_iter, blobs
        ("blobs_k4", 3000, 4, 100, 4),       # converges (exercises the tolerance break, k_means.py:134-136)
        ("blobs_k10", 6000, 10, 10, 7),      # the CLI's max_iter=10 (k_means.py:209), more clusters than blobs
        ("uniform_k13", 4000, 13, 6, 0),     # k > leafsize of the KD-tree (10): the tree really splits
        ("tiny_k3", 40, 3, 100, 2),
        ("dupes_k5", 2500, 5, 12, 3),        # many exactly repeated rows
    ]
    for name, n, k, max_iter, blobs in specs:
        if blobs:
            centres = rng.normal(size=(blobs, 6)) * np.array([4, 4, 4, 1, 1, 1])
            data = centres[rng.integers(0, blobs, size=n)] + rng.normal(size=(n, 6)) * np.array([0.6, 0.6, 0.6, 0.2, 0.2, 0.2])
        else:
            data = rng.uniform(-1, 1, size=(n, 6)) * np.array([5, 5, 5, 1.5, 1.5, 1.5])
        data = data.astype(np.float32)
        if name.startswith("dupes"):
            data[n // 2:] = data[rng.integers(0, n // 8, size=n - n // 2)]
        points, colors = np.ascontiguousarray(data[:, :3]), np.ascontiguousarray(data[:, 3:])
        drawn = {}
        real_choice = np.random.choice

        def recording_choice(a, size=None, replace=True, p=None):
            idx = real_choice(a, size, replace=replace, p=p)
            drawn["idx"] = np.array(idx, dtype=np.int64)
            return idx

        np.random.seed(1000 + len(cases))
        np.random.choice = recording_choice
        try:
            with contextlib.redirect_stdout(io.StringIO()) as log:
                centroids, labels, _ = km.k_means_with_color(points, k, colors.copy(), max_iter=max_iter)
        finally:
            np.random.choice = real_choice
        text = log.getvalue()
        cases[name] = dict(points=points, colors=colors, k=k, max_iter=max_iter, init=drawn["idx"],
                           labels=np.asarray(labels, np.int64), centroids=np.asarray(centroids),
                           converged=("Converged" in text))
        print(name, "n", n, "k", k, "converged" if cases[name]["converged"] else "max_iter", "centroid dtype", centroids.dtype,
              "cluster sizes", np.bincount(labels, minlength=k).tolist())
    flat = {}
    for name, c in cases.items():
        for key, val in c.items():
            flat[f"{name}/{key}"] = np.asarray(val)
    np.savez_compressed(OUT, **flat)
    print("wrote", OUT, os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
