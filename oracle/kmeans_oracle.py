"""ORACLE (test infrastructure, not the product): CPU restatement of the reference's k-means labeler,
3D_clustering/k_means.py:107-151 `k_means_with_color`, in plain numpy.

What the reference computes, line by line:
  :109-110  data = (x, y, z, f_dc_0, f_dc_1, f_dc_2) as float32 rows
  :111      initial centroids = k distinct random rows (unseeded np.random.choice -> injected here as `init`)
  :116-122  every point goes to its nearest centroid: scipy's KD-tree upcasts to float64 and compares SQUARED
            distances, summed as ((d0^2 + d1^2) + d2^2) + d3^2, then + d4^2, + d5^2 (its 4-accumulator loop for 6
            dimensions, no FMA); the smallest wins.  Restated as an arg-min (first minimum).  Exact ties between two
            centroids are broken by the tree's traversal order in the reference: parity on such inputs is unpinned.
  :125-128  new centroid = float32 mean of the member rows: numpy reduces axis 0 of an (M, 6) float32 array by adding
            the rows IN ORDER into a float32 accumulator, then divides by float32(M); an empty cluster keeps its
            centroid
  :132-136  stop when the float32 Frobenius norm of the centroid change is < tol (numpy computes it through BLAS
            sdot; the oracle calls numpy's own norm, so it takes the same decision) - before adopting the new centroids
  :138      otherwise adopt them and iterate, at most max_iter times
  :142-147  final labels from the centroids in hand
Pinned by tests/golden/kmeans.npz (tools/make_golden_kmeans.py runs the reference itself): labels and centroids
bit-exact on all cases.
"""
import numpy as np


def sq_distances(data32, centroids32):
    """(N, k) float64 squared distances with the reference's summation order."""
    p = np.asarray(data32, np.float32).astype(np.float64)
    c = np.asarray(centroids32, np.float32).astype(np.float64)
    d = p[:, None, :] - c[None, :, :]
    d = d * d
    s = ((d[:, :, 0] + d[:, :, 1]) + d[:, :, 2]) + d[:, :, 3]
    s = s + d[:, :, 4]
    return s + d[:, :, 5]


def assign(data32, centroids32, chunk=200_000):
    out = np.empty(len(data32), np.int64)
    for a in range(0, len(data32), chunk):
        out[a:a + chunk] = np.argmin(sq_distances(data32[a:a + chunk], centroids32), axis=1)
    return out


def update(data32, labels, centroids32):
    """k_means.py:125-128 - sequential float32 row sums in index order, / float32(count)."""
    k = len(centroids32)
    new = np.array(centroids32, np.float32, copy=True)
    for c in range(k):
        members = data32[labels == c]
        if len(members):
            new[c] = members.mean(axis=0)          # numpy: rows added in order, float32 accumulator
    return new


def k_means_with_color(points, k, colors, init, max_iter=100, tol=1e-4):
    """Returns (centroids float32 (k, 6), labels int64 (N,), iterations run, converged)."""
    data = np.concatenate((np.asarray(points, np.float32), np.asarray(colors, np.float32)), axis=1)
    centroids = data[np.asarray(init, np.int64)]
    iters, converged = 0, False
    for _ in range(max_iter):
        labels = assign(data, centroids)
        new = update(data, labels, centroids)
        iters += 1
        if np.linalg.norm(new - centroids) < tol:
            converged = True
            break
        centroids = new
    return centroids, assign(data, centroids), iters, converged
