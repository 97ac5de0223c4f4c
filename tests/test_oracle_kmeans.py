"""The k-means oracle (oracle/kmeans_oracle.py) against the vectors the reference itself produced
(tests/golden/kmeans.npz, tools/make_golden_kmeans.py): labels and float32 centroids bit-exact."""
import os

import numpy as np
import pytest

from oracle import kmeans_oracle

GOLD = os.path.join(os.path.dirname(__file__), "golden", "kmeans.npz")


def golden_cases():
    z = np.load(GOLD)
    names = sorted({k.split("/")[0] for k in z.files})
    return [(n, {k.split("/")[1]: z[k] for k in z.files if k.startswith(n + "/")}) for n in names]


@pytest.mark.parametrize("case", golden_cases(), ids=lambda c: c[0])
def test_oracle_equals_reference(case):
    name, g = case
    cent, labels, iters, converged = kmeans_oracle.k_means_with_color(g["points"], int(g["k"]), g["colors"], g["init"],
                                                                       max_iter=int(g["max_iter"]))
    assert np.array_equal(labels, g["labels"]), name
    assert cent.dtype == np.float32 and np.array_equal(cent, g["centroids"]), name
    assert converged == bool(g["converged"])


def test_mean_is_sequential_float32_and_distance_order():
    """The two numerical facts the restatement (and the GPU kernels) rely on."""
    rng = np.random.default_rng(0)
    a = (rng.normal(size=(20_001, 6)) * np.array([1e-3, 1, 1e3, 1, 1, 1])).astype(np.float32)
    acc = np.zeros(6, np.float32)
    for row in a:
        acc = (acc + row).astype(np.float32)
    assert np.array_equal(a.mean(axis=0), acc / np.float32(len(a)))
    from scipy.spatial import KDTree
    cent = rng.normal(size=(13, 6)).astype(np.float32)
    pts = rng.normal(size=(3000, 6)).astype(np.float32)
    tree = KDTree(cent)
    assert np.array_equal(np.array([tree.query(p)[1] for p in pts]), kmeans_oracle.assign(pts, cent))
