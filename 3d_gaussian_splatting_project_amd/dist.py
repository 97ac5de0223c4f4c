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


# ---- protocol v2: all-to-all (reduce-scatter by hand) -> local arg-max -> all-gather of labels -------------
#   1. each rank votes its views into u8 planes laid out [slab][bins][sn] (slab j = rank j's share of the
#      Gaussians), per-rank counters (<= 255 views per rank) and LOCAL first-view codes
#   2. all_to_all over the count plane and over the first-view plane: rank j gets [src rank][bins][sn]
#   3. rank j sums the counters of its slab and breaks ties by (lowest rank, earliest local view)
#   4. all_gather of the sn int32 labels per rank; back to the caller's order
# Moves 2*(world-1)/world * bins*n bytes per rank instead of the all-reduce's ~2*(world-1)/world * 2*bins*n
# (16-bit counters), and the wide histogram crosses xGMI once.
class GpuSlabShard:
    """Rank-local state of protocol v2 on the GPU.  The ctx must have been configured with
    configure_a2a(ctx, world) BEFORE upload/vote_begin."""

    def __init__(self, ctx):
        self.ctx = ctx

    def planes(self):
        self.ctx.vote_flush()
        cp, words = self.ctx.counts_device()
        fp, _ = self.ctx.first_device()
        self.ctx.synchronize()
        dev = self.ctx.device
        return device_words_tensor(cp, words, dev), device_words_tensor(fp, words, dev)

    def reduce(self, recv_cnt, recv_fv):
        torch.cuda.synchronize(self.ctx.device)
        self.ctx.vote_slab_reduce(recv_cnt.data_ptr(), recv_fv.data_ptr())
        kp, _ = self.ctx.keys_device()
        return device_words_tensor(kp, self.ctx.slab_size(), self.ctx.device)

    def finish(self, all_labels, to_host=True):
        torch.cuda.synchronize(self.ctx.device)
        return self.ctx.vote_labels_from_sorted(all_labels.data_ptr(), to_host)


class HostSlabShard:
    """Adapter for a numpy-backed slab shard (tests, gloo)."""

    def __init__(self, shard):
        self.shard = shard

    def planes(self):
        return torch.from_numpy(self.shard.cnt.reshape(-1).view(np.int32)), torch.from_numpy(self.shard.fv.reshape(-1).view(np.int32))

    def reduce(self, recv_cnt, recv_fv):
        return torch.from_numpy(self.shard.reduce(recv_cnt.numpy().view(np.uint8), recv_fv.numpy().view(np.uint8)))

    def finish(self, all_labels, to_host=True):
        return self.shard.finish(all_labels.numpy())


def configure_a2a(ctx, world):
    """Plane layout of protocol v2; call before upload_positions / vote_begin."""
    ctx.set_option("exchange_slabs", world)
    ctx.set_option("exchange_local", 1)


def _all_to_all(out, inp, group):
    if dist.get_backend(group) == "gloo":
        # rehearsal / CPU tests only: gloo has no all_to_all (and no GPU all_gather): stage through the
        # host, gather everything, keep my column
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        host = inp.cpu()
        parts = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(parts, host, group=group)
        chunk = inp.numel() // world
        for r in range(world):
            out[r * chunk:(r + 1) * chunk] = parts[r][rank * chunk:(rank + 1) * chunk].to(out.device)
    else:
        dist.all_to_all_single(out, inp, group=group)


def _all_gather_into(full, part, group):
    if dist.get_backend(group) == "gloo":
        world = dist.get_world_size(group)
        host = part.cpu().contiguous()
        parts = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(parts, host, group=group)
        full.copy_(torch.cat(parts).to(full.device))
    else:
        dist.all_gather_into_tensor(full, part.contiguous(), group=group)


def exchange_labels_a2a(shard, group=None, to_host=True):
    """Steps 2-4 of protocol v2.  `shard` is a GpuSlabShard (RCCL) or HostSlabShard (gloo)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    cnt, fv = shard.planes()
    if world > 1:
        rc, rf = torch.empty_like(cnt), torch.empty_like(fv)
        _all_to_all(rc, cnt, group)
        _all_to_all(rf, fv, group)
    else:
        rc, rf = cnt, fv
    slab = shard.reduce(rc, rf)
    if world > 1:
        full = torch.empty(slab.numel() * world, dtype=slab.dtype, device=slab.device)
        _all_gather_into(full, slab, group)
    else:
        full = slab
    return shard.finish(full, to_host)


# ---- protocol v3: counts-only all-to-all + sparse tie pass ------------------------------------------------------
#   1. fast u8-histogram kernel -> count plane only, u8 [slab][bins][sn]
#   2. all_to_all(counts); slab owner: unique max -> label, else TIED + bit mask of the max-count bins
#   3. all_gather(masks); every rank walks its views forward for the tied Gaussians only, stops at the first
#      candidate vote -> u16 code (255 - local view) << 8 | bin
#   4. all_to_all(codes); slab owner: lowest rank with a code = globally earliest view -> label
#   5. all_gather(labels)
# Half of v2's bytes on the fabric, and the rank-local kernel is the 16-waves/CU one.
class GpuSparseShard:
    def __init__(self, ctx):
        self.ctx = ctx

    def _t(self, ptr_words):
        ptr, words = ptr_words
        return device_words_tensor(ptr, words, self.ctx.device)

    def counts(self):
        self.ctx.vote_flush_counts()
        self.ctx.synchronize()
        return self._t(self.ctx.counts_device())

    def totals(self, recv_cnt):
        torch.cuda.synchronize(self.ctx.device)
        self.ctx.vote_slab_totals(recv_cnt.data_ptr())
        return self._t(self.ctx.cand_device())

    def tie_codes(self, cand_all):
        torch.cuda.synchronize(self.ctx.device)
        self.ctx.vote_tie_codes(cand_all.data_ptr())
        return self._t(self.ctx.codes_device())

    def resolve(self, recv_codes):
        torch.cuda.synchronize(self.ctx.device)
        self.ctx.vote_tie_resolve(recv_codes.data_ptr())
        kp, _ = self.ctx.keys_device()
        return device_words_tensor(kp, self.ctx.slab_size(), self.ctx.device)

    def finish(self, all_labels, to_host=True):
        torch.cuda.synchronize(self.ctx.device)
        return self.ctx.vote_labels_from_sorted(all_labels.data_ptr(), to_host)


class HostSparseShard:
    """Adapter for a numpy-backed v3 shard (tests, gloo)."""

    def __init__(self, shard):
        self.shard = shard

    def counts(self):
        return torch.from_numpy(self.shard.cnt.reshape(-1).view(np.int32))

    def totals(self, recv_cnt):
        return torch.from_numpy(self.shard.totals(recv_cnt.numpy().view(np.uint8)).reshape(-1).view(np.int32))

    def tie_codes(self, cand_all):
        return torch.from_numpy(self.shard.tie_codes(cand_all.numpy().view(np.uint32)).reshape(-1).view(np.int32))

    def resolve(self, recv_codes):
        return torch.from_numpy(self.shard.resolve(recv_codes.numpy().view(np.uint16)))

    def finish(self, all_labels, to_host=True):
        return self.shard.finish(all_labels.numpy())


def exchange_labels_sparse(shard, group=None, to_host=True):
    """Protocol v3.  `shard` is a GpuSparseShard (RCCL) or HostSparseShard (gloo)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    cnt = shard.counts()
    if world > 1:
        rc = torch.empty_like(cnt)
        _all_to_all(rc, cnt, group)
    else:
        rc = cnt
    cand = shard.totals(rc)
    if world > 1:
        cand_all = torch.empty(cand.numel() * world, dtype=cand.dtype, device=cand.device)
        _all_gather_into(cand_all, cand, group)
    else:
        cand_all = cand
    codes = shard.tie_codes(cand_all)
    if world > 1:
        rcodes = torch.empty_like(codes)
        _all_to_all(rcodes, codes, group)
    else:
        rcodes = codes
    slab = shard.resolve(rcodes)
    if world > 1:
        full = torch.empty(slab.numel() * world, dtype=slab.dtype, device=slab.device)
        _all_gather_into(full, slab, group)
    else:
        full = slab
    return shard.finish(full, to_host)


def view_range(n_views_total, rank, world):
    """Contiguous, rank-ordered split of the processed camera list."""
    base, rem = divmod(n_views_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
