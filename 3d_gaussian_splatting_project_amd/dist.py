"""Multi-GPU majority vote: views are sharded over ranks (one process per GPU), every rank holds
all Gaussians, and the per-Gaussian integer vote histogram is all-reduced over RCCL/xGMI.

Protocol (exact, including the reference's first-inserted-label tie rule, dls.py:303):
  1. each rank votes its contiguous block of views  -> planes cnt[bins][n], fv[bins][n]
  2. all-reduce SUM over cnt, viewed as int32 words (8/16-bit counters cannot carry: a bin's total
     is <= total_views, which fits the counter)                       -- the histogram exchange
  3. each rank: among the bins with the global maximum count, key = (fv_local << 8 | bin), the
     bin it saw FIRST; 0 if it saw none of them
  4. all-reduce MAX over the n int32 keys (a larger fv code = an earlier global view)
  5. label = (key & 255) - 1, or -1 when key == 0 (never visible)
Ranks must own contiguous, rank-ordered view ranges [first_view, first_view + k) so that the
global view index orders first votes exactly as the single-process loop (dls.py:255) does.
"""
import numpy as np
import torch
import torch.distributed as dist


class _DeviceWords:
    """Zero-copy int32 view of device memory owned by libgsx (via __cuda_array_interface__)."""

    def __init__(self, ptr, n_words):
        self.__cuda_array_interface__ = {"shape": (int(n_words),), "typestr": "<i4", "data": (int(ptr), False),
                                         "version": 2}


def device_words_tensor(ptr, n_words, device):
    return torch.as_tensor(_DeviceWords(ptr, n_words), device=torch.device("cuda", device))


class GpuVoteShard:
    """One rank's vote state on its MI355X (the product path)."""

    def __init__(self, ctx):
        self.ctx = ctx

    def counts_tensor(self):
        self.ctx.vote_flush()
        ptr, n = self.ctx.counts_device()
        self.ctx.synchronize()
        return device_words_tensor(ptr, n, self.ctx.device)

    def compute_keys(self):
        torch.cuda.synchronize(self.ctx.device)      # the reduced counts must have landed
        self.ctx.vote_tiebreak_keys()
        ptr, n = self.ctx.keys_device()
        self.ctx.synchronize()
        return device_words_tensor(ptr, n, self.ctx.device)

    def labels(self, to_host=True):
        torch.cuda.synchronize(self.ctx.device)
        return self.ctx.vote_labels_from_keys(to_host)


class HostVoteShard:
    """Adapter for a numpy-backed shard (tests, gloo)."""

    def __init__(self, shard):
        self.shard = shard

    def counts_tensor(self):
        return torch.from_numpy(self.shard.counts_words())

    def compute_keys(self):
        self.shard.compute_keys()
        return torch.from_numpy(self.shard.keys)

    def labels(self, to_host=True):
        return self.shard.labels_from_keys()


def exchange_labels(shard, group=None, to_host=True):
    """Steps 2-5 above.  `shard` is a GpuVoteShard (RCCL) or HostVoteShard (gloo)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    counts = shard.counts_tensor()
    if world > 1:
        dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
    keys = shard.compute_keys()
    if world > 1:
        dist.all_reduce(keys, op=dist.ReduceOp.MAX, group=group)
    return shard.labels(to_host)


def view_range(n_views_total, rank, world):
    """Contiguous, rank-ordered split of the processed camera list."""
    base, rem = divmod(n_views_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
