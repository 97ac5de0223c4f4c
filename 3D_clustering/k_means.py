#!/usr/bin/env python3
"""Drop-in for the reference's 3D_clustering/k_means.py on libgsx.so (MI355X): k-means over (xyz | f_dc) rows, the
cluster index written as `property int label` into an ASCII PLY - the same field the majority-vote labeler fills.

Same functions and CLI as the reference (`--file_path --save_path --k`, 10 iterations as at k_means.py:209); the
assignment and the centroid update run as HIP kernels (csrc/kmeans.hip) and reproduce the reference's float64 distance
and float32 mean arithmetic bit for bit.  The reference draws its initial centroids with an unseeded
np.random.choice (k_means.py:111); here the draw happens in `k_means_with_color` too (numpy's global generator, so
`np.random.seed` / `--seed` make a run repeatable) unless the rows are given (`init=` / `--init i,j,...`).
There is no CPU path: without libgsx.so and a gfx950 GPU this raises."""
import argparse
import importlib
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_HERE)
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)
_ply = importlib.import_module("3d_gaussian_splatting_project_amd.ply_io")
_labeler = importlib.import_module("3d_gaussian_splatting_project_amd.labeler")
PlyData = _ply.PlyData

COLORS = [[252, 199, 55], [242, 107, 15], [231, 56, 121], [126, 24, 145], [247, 44, 91], [255, 116, 139], [167, 212, 119],
          [228, 241, 172]]


def get_vertex_info(plydata):
    """(N, 3) positions and (N, 3) diffuse colours f_dc_0..2 (k_means.py:10-32)."""
    vertices = plydata["vertex"]
    points = np.column_stack((vertices["x"], vertices["y"], vertices["z"]))
    colors = np.column_stack((vertices["f_dc_0"], vertices["f_dc_1"], vertices["f_dc_2"]))
    return points, colors


def k_means_with_color(points, k, colors, max_iter=100, tol=1e-4, init=None, ctx=None):
    """k_means.py:107-151.  Returns (centroids, labels, colors) like the reference: centroids float32 (k, 6), labels
    int (N,), colors = the palette colour of every point's cluster (in [0, 1])."""
    points = np.asarray(points, np.float32)
    colors = np.array(colors, np.float32)
    n = len(points)
    if init is None:
        init = np.random.choice(n, k, replace=False)                 # k_means.py:111
    own = ctx is None
    if own:
        ctx = _labeler.Context(0)
    try:
        centroids, labels, iters, converged = ctx.kmeans(points, colors, k, init, max_iter=max_iter, tol=tol)
    finally:
        if own:
            ctx.close()
    if converged:
        print(f"Converged after {iters} iterations.")
    labels = labels.astype(np.int64)
    for cluster in range(k):                                        # k_means.py:147-149
        colors[labels == cluster] = np.array(COLORS[cluster % len(COLORS)]) / 255.0
    return centroids, labels, colors


def add_label_proberty(ply_data, output_ply, label):
    """Every vertex property of `ply_data` plus `property int label`, ASCII (k_means.py:169-194)."""
    ply_data.write(output_ply, labels=np.asarray(label, np.int32), text=True)
    print(f"New PLY file with label added saved to {output_ply}")


def main(argv=None):
    parser = argparse.ArgumentParser(description="K-means clustering on a point cloud.")
    parser.add_argument("--file_path", type=str, required=True, help="Path to the input PLY file.")
    parser.add_argument("--save_path", type=str, required=True, help="Path to save the modified PLY file.")
    parser.add_argument("--k", type=int, default=10, help="Number of clusters for k-means.")
    parser.add_argument("--seed", type=int, default=None, help="seed numpy's generator before the initial draw")
    parser.add_argument("--init", type=str, default=None, help="comma-separated row indices of the initial centroids")
    args = parser.parse_args(argv)
    if args.seed is not None:
        np.random.seed(args.seed)
    plydata = PlyData.read(args.file_path)
    points, colors = get_vertex_info(plydata)
    init = None if args.init is None else np.array([int(v) for v in args.init.split(",")], np.int64)
    _, labels, _ = k_means_with_color(points, args.k, colors, max_iter=10, init=init)   # k_means.py:209
    print(labels)
    add_label_proberty(plydata, args.save_path, labels)


if __name__ == "__main__":
    main()
