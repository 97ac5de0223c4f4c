"""Test helper: the oracle's vote counted one view at a time (see the class)."""
import numpy as np

import oracle


class StreamingOracle:
    """The reference's vote (dls.py:288-306) for a sample of Gaussians, one view at a time: the oracle's per-view votes
    (oracle.view_bins) are counted as they come and the first-inserted label among the maxima - the one whose first vote
    came from the earliest view - wins.  Lets a test walk hundreds of 4K maps without holding them all."""

    def __init__(self, pos, n_classes):
        self.pos = np.ascontiguousarray(pos, np.float32)
        self.cnt = np.zeros((len(pos), n_classes + 1), np.int32)
        self.first = np.full((len(pos), n_classes + 1), np.iinfo(np.int32).max, np.int32)
        self.v = 0

    def view(self, cam, seg, size):
        b = oracle.view_bins(self.pos, cam, seg, size)
        i = np.nonzero(b >= 0)[0]
        self.cnt[i, b[i]] += 1
        self.first[i, b[i]] = np.minimum(self.first[i, b[i]], self.v)
        self.v += 1

    def labels(self):
        M = self.cnt.max(axis=1)
        cand = (self.cnt == M[:, None]) & (self.cnt > 0)
        win = np.where(cand, self.first, np.iinfo(np.int32).max).argmin(axis=1)
        return np.where(M > 0, win - 1, -1).astype(np.int32)
