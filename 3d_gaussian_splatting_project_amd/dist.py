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
import contextlib

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


class _DeviceBytes:
    def __init__(self, ptr, n_bytes):
        self.__cuda_array_interface__ = {"shape": (int(n_bytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def device_bytes_tensor(ptr, n_bytes, device):
    return torch.as_tensor(_DeviceBytes(ptr, n_bytes), device=torch.device("cuda", device))


class _GpuShard:
    """Common part of the GPU shards: every collective is issued with the ctx's own HIP stream as torch's current
    stream, so libgsx kernels, RCCL collectives and the copies in between are ordered by the stream (RCCL's internal
    stream joins it through events) and the host never waits between the stages of a protocol."""

    def __init__(self, ctx):
        self.ctx = ctx
        self._ext = None
        world = dist.get_world_size() if dist.is_initialized() else 1
        if _collectives_needed(world) and hasattr(ctx, "set_option"):
            # a rank of a multi-rank run never holds ALL views when it votes its own block: the early vote's result would always be
            # discarded (1.2 GB of planes and 1.6 ms of GPU work competing with the all-gather on rank 0, whose block starts at view 0)
            ctx.set_option("early_vote", 0)

    def stream(self):
        if self._ext is None:
            self._ext = torch.cuda.ExternalStream(self.ctx.stream, device=torch.device("cuda", self.ctx.device))
        return torch.cuda.stream(self._ext)


def _stream_of(shard):
    return shard.stream() if hasattr(shard, "stream") else contextlib.nullcontext()


class GpuVoteShard(_GpuShard):
    """One rank's vote state on its MI355X (the product path)."""

    def counts_tensor(self):
        self.ctx.vote_flush()
        ptr, n = self.ctx.counts_device()
        return device_words_tensor(ptr, n, self.ctx.device)

    def compute_keys(self):
        self.ctx.vote_tiebreak_keys()
        ptr, n = self.ctx.keys_device()
        return device_words_tensor(ptr, n, self.ctx.device)

    def labels(self, to_host=True, out=None):
        return self.ctx.vote_labels_from_keys(to_host, out=out)


class HostVoteShard:
    """Adapter for a numpy-backed shard (tests, gloo)."""

    def __init__(self, shard):
        self.shard = shard

    def counts_tensor(self):
        return torch.from_numpy(self.shard.counts_words())

    def compute_keys(self):
        self.shard.compute_keys()
        return torch.from_numpy(self.shard.keys)

    def labels(self, to_host=True, out=None):
        return _into(out, self.shard.labels_from_keys())


def _into(out, labels):
    if out is None:
        return labels
    out[...] = labels
    return out


def exchange_labels(shard, group=None, to_host=True, out=None):
    """Steps 2-5 above.  `shard` is a GpuVoteShard (RCCL) or HostVoteShard (gloo)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    with _stream_of(shard):
        counts = shard.counts_tensor()
        if _collectives_needed(world):
            dist.all_reduce(counts, op=dist.ReduceOp.SUM, group=group)
        keys = shard.compute_keys()
        if _collectives_needed(world):
            dist.all_reduce(keys, op=dist.ReduceOp.MAX, group=group)
        return shard.labels(to_host, out=out)


# ---- protocol v2: all-to-all (reduce-scatter by hand) -> local arg-max -> all-gather of labels -------------
#   1. each rank votes its views into u8 planes laid out [slab][bins][sn] (slab j = rank j's share of the
#      Gaussians), per-rank counters (<= 255 views per rank) and LOCAL first-view codes
#   2. all_to_all over the count plane and over the first-view plane: rank j gets [src rank][bins][sn]
#   3. rank j sums the counters of its slab and breaks ties by (lowest rank, earliest local view)
#   4. all_gather of the sn int32 labels per rank; back to the caller's order
# Moves 2*(world-1)/world * bins*n bytes per rank instead of the all-reduce's ~2*(world-1)/world * 2*bins*n
# (16-bit counters), and the wide histogram crosses xGMI once.
class GpuSlabShard(_GpuShard):
    """Rank-local state of protocol v2 on the GPU.  The ctx must have been configured with
    configure_a2a(ctx, world) BEFORE upload/vote_begin."""

    def planes(self):
        self.ctx.vote_flush()
        cp, words = self.ctx.counts_device()
        fp, _ = self.ctx.first_device()
        dev = self.ctx.device
        return device_words_tensor(cp, words, dev), device_words_tensor(fp, words, dev)

    def reduce(self, recv_cnt, recv_fv):
        self.ctx.vote_slab_reduce(recv_cnt.data_ptr(), recv_fv.data_ptr())
        kp, _ = self.ctx.keys_device()
        return device_words_tensor(kp, self.ctx.slab_size(), self.ctx.device)

    def finish(self, all_labels, to_host=True, out=None):
        return self.ctx.vote_labels_from_sorted(all_labels.data_ptr(), to_host, out=out)


class HostSlabShard:
    """Adapter for a numpy-backed slab shard (tests, gloo)."""

    def __init__(self, shard):
        self.shard = shard

    def planes(self):
        return torch.from_numpy(self.shard.cnt.reshape(-1).view(np.int32)), torch.from_numpy(self.shard.fv.reshape(-1).view(np.int32))

    def reduce(self, recv_cnt, recv_fv):
        return torch.from_numpy(self.shard.reduce(recv_cnt.numpy().view(np.uint8), recv_fv.numpy().view(np.uint8)))

    def finish(self, all_labels, to_host=True, out=None):
        return _into(out, self.shard.finish(all_labels.numpy()))


def configure_a2a(ctx, world):
    """Plane layout of protocol v2; call before upload_positions / vote_begin."""
    ctx.set_option("exchange_slabs", world)
    ctx.set_option("exchange_local", 1)


def _collectives_needed(world):
    """world > 1; or, for rehearsing the real RCCL calls on a one-GPU box, a 1-rank group with GSX_DIST_FORCE_COLLECTIVES=1
    (every collective then runs for real through RCCL, on this rank alone)."""
    import os
    return world > 1 or (dist.is_initialized() and os.environ.get("GSX_DIST_FORCE_COLLECTIVES") == "1")


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


def _all_gather_into(full, part, group, async_op=False):
    """-> a Work to wait() on when async_op and the backend can do it, else None (the gather has been completed)."""
    if dist.get_backend(group) == "gloo":
        world = dist.get_world_size(group)
        host = part.cpu().contiguous()
        parts = [torch.empty_like(host) for _ in range(world)]
        dist.all_gather(parts, host, group=group)
        full.copy_(torch.cat(parts).to(full.device))
        return None
    return dist.all_gather_into_tensor(full, part.contiguous(), group=group, async_op=async_op) if async_op else \
        dist.all_gather_into_tensor(full, part.contiguous(), group=group)


def exchange_labels_a2a(shard, group=None, to_host=True, out=None):
    """Steps 2-4 of protocol v2.  `shard` is a GpuSlabShard (RCCL) or HostSlabShard (gloo)."""
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    with _stream_of(shard):
        cnt, fv = shard.planes()
        if _collectives_needed(world):
            rc, rf = torch.empty_like(cnt), torch.empty_like(fv)
            _all_to_all(rc, cnt, group)
            _all_to_all(rf, fv, group)
        else:
            rc, rf = cnt, fv
        slab = shard.reduce(rc, rf)
        if _collectives_needed(world):
            full = torch.empty(slab.numel() * world, dtype=slab.dtype, device=slab.device)
            _all_gather_into(full, slab, group)
        else:
            full = slab
        return shard.finish(full, to_host, out=out)


# ---- protocol v3: counts-only all-to-all + sparse tie pass ------------------------------------------------------
#   1. fast u8-histogram kernel -> count plane only, u8 [slab][bins][sn]
#   2. all_to_all(counts); slab owner: unique max -> label, else TIED + bit mask of the max-count bins
#   3. all_gather(masks); every rank walks its views forward for the tied Gaussians only, stops at the first
#      candidate vote -> u16 code (255 - local view) << 8 | bin
#   4. all_to_all(codes); slab owner: lowest rank with a code = globally earliest view -> label
#   5. all_gather(labels)
# Half of v2's bytes on the fabric, and the rank-local kernel is the 16-waves/CU one.
class GpuSparseShard(_GpuShard):
    def _t(self, ptr_words):
        ptr, words = ptr_words
        return device_words_tensor(ptr, words, self.ctx.device)

    def counts(self):
        self.ctx.vote_flush_counts()
        return self._t(self.ctx.counts_device())

    def totals(self, recv_cnt):
        self.ctx.vote_slab_totals(recv_cnt.data_ptr())
        return self._t(self.ctx.cand_device())

    def tie_codes(self, cand_all):
        self.ctx.vote_tie_codes(cand_all.data_ptr())
        return self._t(self.ctx.codes_device())

    def resolve(self, recv_codes):
        self.ctx.vote_tie_resolve(recv_codes.data_ptr())
        kp, _ = self.ctx.keys_device()
        return device_words_tensor(kp, self.ctx.slab_size(), self.ctx.device)

    def finish(self, all_labels, to_host=True, out=None):
        return self.ctx.vote_labels_from_sorted(all_labels.data_ptr(), to_host, out=out)


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

    def finish(self, all_labels, to_host=True, out=None):
        return _into(out, self.shard.finish(all_labels.numpy()))


def exchange_labels_sparse(shard, group=None, to_host=True, out=None):
    """Protocol v3.  `shard` is a GpuSparseShard (RCCL) or HostSparseShard (gloo)."""
    with _stream_of(shard):
        return _exchange_labels_sparse(shard, group, to_host, out)


def _exchange_labels_sparse(shard, group, to_host, out):
    world = dist.get_world_size(group) if dist.is_initialized() else 1
    cnt = shard.counts()
    if _collectives_needed(world):
        rc = torch.empty_like(cnt)
        _all_to_all(rc, cnt, group)
    else:
        rc = cnt
    cand = shard.totals(rc)
    if _collectives_needed(world):
        cand_all = torch.empty(cand.numel() * world, dtype=cand.dtype, device=cand.device)
        _all_gather_into(cand_all, cand, group)
    else:
        cand_all = cand
    codes = shard.tie_codes(cand_all)
    if _collectives_needed(world):
        rcodes = torch.empty_like(codes)
        _all_to_all(rcodes, codes, group)
    else:
        rcodes = codes
    slab = shard.resolve(rcodes)
    if _collectives_needed(world):
        full = torch.empty(slab.numel() * world, dtype=slab.dtype, device=slab.device)
        _all_gather_into(full, slab, group)
    else:
        full = slab
    return shard.finish(full, to_host, out=out)


# ---- protocol v4: all-gather of the packed maps -> every rank votes its slab of the Gaussians -> all-gather of labels
#   0. every rank has staged its contiguous, rank-ordered block of views (packed u8 maps in its pool)
#   1. all_gather of a small header per rank: view count, pool bytes, the view blobs (camera + map geometry)
#   2. all_gather of the pools, padded to the largest (`chunk` bytes per rank): 0.44 GB in total for 200 1080p views,
#      against a 0.45 GB histogram PER RANK in v1-v3
#   3. import: every rank now holds all views; it votes Gaussians [rank*S, (rank+1)*S) of the Morton order with the
#      single-GPU fused kernel (all views in order -> the reference's tie rule needs no exchange)
#   4. all_gather of the S int32 labels per rank; back to the caller's order
_HDR = 16  # bytes: int64 views, int64 pool bytes


class GpuGatherShard(_GpuShard):
    def __init__(self, ctx):
        super().__init__(ctx)
        self._pool_all = None
        self._labels_all = None

    def header(self, cap_views):
        """uint8 tensor [_HDR + 256 * cap_views] on the device: this rank's view count, pool bytes and view blobs."""
        _, used, blobs = self.ctx.vote_export(0)
        h = np.zeros(_HDR + 256 * cap_views, np.uint8)
        h[:_HDR].view(np.int64)[:] = (len(blobs), used)
        h[_HDR:_HDR + blobs.size] = blobs.reshape(-1)
        return torch.from_numpy(h).to(torch.device("cuda", self.ctx.device), non_blocking=False)

    def pool(self, chunk):
        ptr, _, _ = self.ctx.vote_export(chunk, blobs=False)
        return device_bytes_tensor(ptr, chunk, self.ctx.device)

    def pool_all(self, nbytes):
        if self._pool_all is None or self._pool_all.numel() < nbytes:
            self._pool_all = torch.empty(nbytes, dtype=torch.uint8, device=torch.device("cuda", self.ctx.device))
        return self._pool_all[:nbytes]

    def staged(self):
        """(views staged so far, pool bytes in use) - no stream work"""
        return self.ctx.vote_num_views(), self.ctx.vote_pool_bytes()

    def flush(self):
        """every staged map's DMA is queued on the ctx stream (packed host maps go up in groups)"""
        self.ctx.vote_export(0, blobs=False)

    def device(self):
        return torch.device("cuda", self.ctx.device)

    def import_all(self, part_views, part_offsets, blobs, pool_all):
        self.ctx.vote_import(part_views, part_offsets, blobs, pool_all.data_ptr(), pool_all.numel())

    def import_uniform(self, part_views, part_offsets, cameras, map_size, image_size, pool_all):
        self.ctx.vote_import_uniform(part_views, part_offsets, cameras, map_size, image_size, pool_all.data_ptr(), pool_all.numel())

    def undo_import(self):
        self.ctx.vote_import_undo()

    def event(self):
        """a timing event recorded on the ctx stream (torch's current stream must be the ctx stream: `with shard.stream()`)"""
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def slab_labels(self, rank, world):
        sn = self.ctx.vote_slab_labels(rank, world)
        kp, _ = self.ctx.keys_device()
        return device_words_tensor(kp, sn, self.ctx.device)

    def labels_all(self, n):
        if self._labels_all is None or self._labels_all.numel() < n:
            self._labels_all = torch.empty(n, dtype=torch.int32, device=torch.device("cuda", self.ctx.device))
        return self._labels_all[:n]

    def finish(self, all_labels, to_host=True, out=None):
        return self.ctx.vote_labels_from_sorted(all_labels.data_ptr(), to_host, out=out)


class HostGatherShard:
    """Adapter for a numpy-backed v4 shard (tests, gloo)."""

    def __init__(self, shard):
        self.shard = shard

    def header(self, cap_views):
        blobs, used = self.shard.export()
        h = np.zeros(_HDR + 256 * cap_views, np.uint8)
        h[:_HDR].view(np.int64)[:] = (len(blobs), used)
        h[_HDR:_HDR + blobs.size] = blobs.reshape(-1)
        return torch.from_numpy(h)

    def pool(self, chunk):
        return torch.from_numpy(self.shard.pool(chunk))

    def pool_all(self, nbytes):
        return torch.empty(nbytes, dtype=torch.uint8)

    def staged(self):
        return self.shard.staged()

    def flush(self):
        pass

    def device(self):
        return torch.device("cpu")

    def import_all(self, part_views, part_offsets, blobs, pool_all):
        self.shard.import_all(part_views, part_offsets, blobs, pool_all.numpy())

    def import_uniform(self, part_views, part_offsets, cameras, map_size, image_size, pool_all):
        self.shard.import_uniform(part_views, part_offsets, cameras, map_size, image_size, pool_all.numpy())

    def undo_import(self):
        self.shard.views = None

    def slab_labels(self, rank, world):
        return torch.from_numpy(self.shard.slab_labels(rank, world))

    def labels_all(self, n):
        return torch.empty(n, dtype=torch.int32)

    def finish(self, all_labels, to_host=True, out=None):
        return _into(out, self.shard.finish(all_labels.numpy()))


def exchange_labels_gather(shard, group=None, to_host=True, out=None, cap_views=None):
    """Protocol v4.  `shard` is a GpuGatherShard (RCCL) or HostGatherShard (gloo).  cap_views: an upper bound on the
    views of any one rank that all ranks agree on (default: 1024); sizes the header exchange."""
    on = dist.is_initialized()
    world = dist.get_world_size(group) if on else 1
    rank = dist.get_rank(group) if on else 0
    cap = int(cap_views) if cap_views else 1024
    # 1. headers.  Issued on torch's own stream, NOT the ctx stream: it does not have to wait for the maps' DMA.
    mine = shard.header(cap)
    if _collectives_needed(world):
        heads = torch.empty(world * mine.numel(), dtype=torch.uint8, device=mine.device)
        _all_gather_into(heads, mine, group)
    else:
        heads = mine
    heads = heads.cpu().numpy().reshape(world, -1)
    counts = heads[:, :_HDR].copy().view(np.int64).reshape(world, 2)
    part_views = counts[:, 0].astype(np.int32)
    if int(part_views.max()) > cap:
        raise ValueError(f"exchange_labels_gather: a rank staged {int(part_views.max())} views, more than cap_views={cap}")
    chunk = max(256, int(counts[:, 1].max()))
    blobs = np.concatenate([heads[r, _HDR:_HDR + 256 * part_views[r]] for r in range(world)])
    with _stream_of(shard):
        # 2. the maps
        pool = shard.pool(chunk)
        if _collectives_needed(world):
            pool_all = shard.pool_all(world * chunk)
            _all_gather_into(pool_all, pool, group)
        else:
            pool_all = pool
        # 3. my slab of the Gaussians over all views
        shard.import_all(part_views, np.arange(world, dtype=np.int64) * chunk, blobs, pool_all)
        slab = shard.slab_labels(rank, world)
        # 4. the labels
        if _collectives_needed(world):
            full = shard.labels_all(slab.numel() * world)
            _all_gather_into(full, slab, group)
        else:
            full = slab
        return shard.finish(full, to_host, out=out)


def chunk_bounds(n_max, chunks):
    """Cut points 0 = b_0 < b_1 < .. < b_C = n_max of the ranks' view blocks (a function of (n_max, chunks) alone: every rank
    derives the same).  Balanced chunks and a SHORT last one (an eighth of the block, at least one view): the last chunk's
    all-gather is the one no hand-over hides, so it should carry little."""
    n_max, chunks = int(n_max), max(1, int(chunks))
    if n_max <= 0:
        return [0]
    if chunks == 1 or n_max < 2 * chunks:
        m = -(-n_max // chunks)
        return list(range(0, n_max, m)) + [n_max]
    tail = max(1, n_max // 8)
    body = n_max - tail
    return [body * k // (chunks - 1) for k in range(chunks - 1)] + [body, n_max]


class GatherPipeline:
    """Protocol v4 with the all-gather of the packed maps OVERLAPPED with the hand-over of the later views.

        pipe = GatherPipeline(shard, total_views, cameras=all_cameras, map_size=(w, h))     # after gsx_vote_begin
        for view in my block:  ctx.vote_view(...);  pipe.after_view()
        labels = pipe.finish(out=...)

    Every rank must own the block view_range(total_views, rank, world) of views.  The ranks' blocks are cut into C chunks
    (chunk_bounds: balanced, the last one short); as soon as a rank has staged the views of chunk j it joins all_gather
    number j (async, on the ctx stream: the DMAs of the later views keep flowing, RCCL's stream picks the chunk up behind
    the copies that fill it).  The sequence of collectives is a function of (total_views, world) alone, identical on every
    rank whatever happens locally.

    Two ways to learn the other ranks' views:
      * cameras + map_size given (round 3; a capture from ONE camera model, which is what a cameras.json describes): every rank
        derives every view's descriptor itself (gsx_vote_import_uniform) - no header exchange, no host wait anywhere between the
        first vote_view and the labels.  Collectives: C chunk all_gathers, one 4-byte flag all_gather nobody waits for until
        the labels are there, the labels all_gather.  A rank whose own pool is not what the schedule assumes (fewer views than
        its share, a map of another geometry) raises its flag; every rank sees the flags next to the labels and all fall back
        to the plain gather together.
      * otherwise (round 2): one agreement all_gather of the ranks' map strides (skipped with assume_uniform), the chunk
        all_gathers, a header all_gather carrying the view blobs (the host waits for it), the labels all_gather; mixed
        geometries fall back to the plain gather of exchange_labels_gather.
    timing=True: finish() leaves self.phases_ms (host stamps + events on the ctx stream)."""

    def __init__(self, shard, total_views, group=None, chunks=4, assume_uniform=False, cameras=None, map_size=None, image_size=None,
                 timing=False):
        self.shard, self.group, self.total = shard, group, int(total_views)
        self.local_views = cameras is not None and map_size is not None
        self.assume_uniform = bool(assume_uniform) or self.local_views
        self.cameras, self.map_size = cameras, map_size
        self.image_size = image_size if image_size is not None else map_size
        if self.local_views and len(cameras) != self.total:
            raise ValueError(f"GatherPipeline: {len(cameras)} cameras for {self.total} views")
        on = dist.is_initialized()
        self.world = dist.get_world_size(group) if on else 1
        self.rank = dist.get_rank(group) if on else 0
        self.active = _collectives_needed(self.world)
        self.n = [hi - lo for lo, hi in (view_range(self.total, r, self.world) for r in range(self.world))]
        n_max = max(self.n) if self.n else 0
        self.bounds = chunk_bounds(n_max, chunks)               # views [b_j, b_j+1) of every rank's block form chunk j
        self.C = len(self.bounds) - 1                           # chunk all_gathers every rank will issue
        self.stride = None       # bytes per staged map, agreed by all ranks; 0: no pipelining (fallback)
        self.next_chunk = 0
        self.works = []
        self.pool_all = None
        self.pool_src = None
        self.timing = bool(timing)
        self.phases_ms = None
        self._t_first = self._t_last = None

    # -- agreement on the map stride: the first collective of every rank -------------------------------------------------
    def _agree(self):
        nv, used = self.shard.staged()
        mine = used // nv if nv and used % nv == 0 else (-1 if nv == 0 else 0)   # -1: no view yet (no opinion); 0: irregular
        if self.assume_uniform and min(self.n) >= 1:
            if not nv:  # the peers are about to join chunk collectives this rank cannot size: fail loudly, not silently out of step
                raise RuntimeError(f"GatherPipeline: rank {self.rank} staged no view, but view_range gives it {self.n[self.rank]}")
            self.stride = mine if mine > 0 and mine % 256 == 0 else 0             # every rank derives the same number locally
        else:
            dev = self.shard.device()
            t = torch.tensor([mine], dtype=torch.int64, device=dev)
            alls = torch.empty(self.world, dtype=torch.int64, device=dev)
            _all_gather_into(alls, t, self.group)
            vals = [int(v) for v in alls.cpu().tolist()]
            have = [v for v in vals if v >= 0]
            self.stride = have[0] if have and all(v == have[0] and v > 0 and v % 256 == 0 for v in have) else 0
        if self.stride:
            with _stream_of(self.shard):
                need = self.bounds[-1] * self.stride
                self.pool_src = self.shard.pool(need)                     # reserves: chunk j reads (b_j+1 - b_j) * stride bytes
                self.pool_all = self.shard.pool_all(self.world * need)

    def _chunk_at(self, j):
        """byte offset of chunk j's region in the gathered buffer, bytes per rank in it"""
        return self.world * self.bounds[j] * self.stride, (self.bounds[j + 1] - self.bounds[j]) * self.stride

    def _issue(self, j):
        if self.stride:
            at, cb = self._chunk_at(j)
            self.shard.flush()
            with _stream_of(self.shard):
                w = _all_gather_into(self.pool_all[at:at + self.world * cb], self.pool_src[self.bounds[j] * self.stride:self.bounds[j] * self.stride + cb],
                                     self.group, async_op=True)
            if w is not None:
                self.works.append(w)
        self.next_chunk = j + 1

    def after_view(self):
        if self.timing:
            import time
            self._t_last = time.perf_counter()
            if self._t_first is None:
                self._t_first = self._t_last
        if not self.active:
            return
        nv, _ = self.shard.staged()
        if self.stride is None:
            self._agree()
        me = self.n[self.rank]
        while self.next_chunk < self.C and nv >= min(self.bounds[self.next_chunk + 1], me):
            self._issue(self.next_chunk)

    def _parts(self, relative):
        """(views, byte offset) of every (rank, chunk) part in global view order = rank-major, chunk-minor.  relative: the offset
        the round-2 blobs are rebased by (they hold offsets inside the exporting rank's own pool, b_j * stride + ..)."""
        pv, po = [], []
        for r in range(self.world):
            for j in range(self.C):
                k = min(max(self.n[r] - self.bounds[j], 0), self.bounds[j + 1] - self.bounds[j])
                if k:
                    at, cb = self._chunk_at(j)
                    pv.append(k)
                    po.append(at + r * cb - (self.bounds[j] * self.stride if relative else 0))
        return np.asarray(pv, np.int32), np.asarray(po, np.int64)

    def _mark(self, events, name):
        if self.timing and hasattr(self.shard, "event"):
            events.append((name, self.shard.event()))

    def finish(self, to_host=True, out=None):
        import time
        shard, world, rank = self.shard, self.world, self.rank
        if not self.active:
            return exchange_labels_gather(shard, self.group, to_host, out, cap_views=max(1, self.total))
        t_fin = time.perf_counter()
        if self.stride is None:
            self._agree()                                   # a rank without views gets here first
        while self.next_chunk < self.C:
            self._issue(self.next_chunk)
        nv, used = shard.staged()
        regular = bool(self.stride) and nv == self.n[rank] and used == nv * self.stride
        events = []
        self._mark(events, "start")
        if self.local_views:
            return self._finish_local(regular, events, t_fin, to_host, out)
        # header: counts, bytes, blobs - and whether my pool is what the chunk schedule assumed
        cap = max(1, max(self.n))
        mine = shard.header(cap)
        flag = torch.tensor([1 if regular else 0], dtype=torch.uint8, device=mine.device)
        mine = torch.cat([mine, flag])
        heads = torch.empty(world * mine.numel(), dtype=torch.uint8, device=mine.device)
        _all_gather_into(heads, mine, self.group)
        heads = heads.cpu().numpy().reshape(world, -1)
        counts = heads[:, :_HDR].copy().view(np.int64).reshape(world, 2)
        part_views = counts[:, 0].astype(np.int64)
        ok = bool(heads[:, -1].all()) and [int(v) for v in part_views] == self.n
        with _stream_of(shard):                             # Work.wait() makes the CURRENT stream wait: it must be the ctx stream,
            for w in self.works:                            # on which the import's uploads and the slab vote are queued
                w.wait()
        self.works = []
        if not ok:                                          # every rank sees the same flags: all take the plain path together
            return exchange_labels_gather(shard, self.group, to_host, out, cap_views=max(1, self.total))
        blobs = np.concatenate([heads[r, _HDR:_HDR + 256 * int(part_views[r])] for r in range(world)])
        pv, po = self._parts(relative=True)
        with _stream_of(shard):
            self._mark(events, "gathers")
            shard.import_all(pv, po, blobs, self.pool_all[:world * self.bounds[-1] * self.stride])
            slab = shard.slab_labels(rank, world)
            self._mark(events, "vote")
            full = shard.labels_all(slab.numel() * world)
            _all_gather_into(full, slab, self.group)
            self._mark(events, "labels")
            res = shard.finish(full, to_host, out=out)
        self._phases(events, t_fin)
        return res

    def _finish_local(self, regular, events, t_fin, to_host, out):
        shard, world, rank = self.shard, self.world, self.rank
        dev = shard.device()
        # the flags: on torch's own stream, nobody waits for them before the labels are on the host
        flag = torch.tensor([1 if regular else 0], dtype=torch.int32, device=dev)
        flags = torch.empty(world, dtype=torch.int32, device=dev)
        fw = _all_gather_into(flags, flag, self.group, async_op=True)
        with _stream_of(shard):                             # Work.wait() makes the CURRENT stream wait: the ctx stream
            for w in self.works:
                w.wait()
            self.works = []
            self._mark(events, "gathers")
            res = None
            if self.stride:
                pv, po = self._parts(relative=False)
                shard.import_uniform(pv, po, self.cameras, self.map_size, self.image_size,
                                     self.pool_all[:world * self.bounds[-1] * self.stride])
                slab = shard.slab_labels(rank, world)
                self._mark(events, "vote")
                full = shard.labels_all(slab.numel() * world)
                _all_gather_into(full, slab, self.group)
                self._mark(events, "labels")
                res = shard.finish(full, to_host, out=out)   # the host waits here (labels D2H), for the first time in the run
        if fw is not None:
            fw.wait()
        ok = bool(self.stride) and bool(flags.cpu().numpy().all())
        if not ok:                                          # every rank sees the same flags: all take the plain path together
            if self.stride:
                shard.undo_import()                         # ... from their own views again
            return exchange_labels_gather(shard, self.group, to_host, out, cap_views=max(1, self.total))
        self._phases(events, t_fin)
        return res

    def _phases(self, events, t_fin):
        """ms: hand_over = first vote_view -> last one returned (host); then, on the ctx stream: chunk all_gathers still
        running when finish() was called; import + slab vote; labels all_gather; labels to the host (D2H + widening)."""
        if not self.timing:
            return
        import time
        t_end = time.perf_counter()
        ph = {"hand_over": None if self._t_first is None else (t_fin - self._t_first) * 1e3,
              "finish_host_total": (t_end - t_fin) * 1e3}
        ev = dict(events)
        if all(k in ev for k in ("start", "gathers", "vote", "labels")):
            for e in ev.values():
                e.synchronize()
            ph["gathers_exposed"] = ev["start"].elapsed_time(ev["gathers"])
            ph["import_and_slab_vote"] = ev["gathers"].elapsed_time(ev["vote"])
            ph["labels_all_gather"] = ev["vote"].elapsed_time(ev["labels"])
            ph["labels_to_host"] = max(0.0, ph["finish_host_total"] - ev["start"].elapsed_time(ev["labels"]))
        self.phases_ms = {k: (None if v is None else round(float(v), 4)) for k, v in ph.items()}


def view_range(n_views_total, rank, world):
    """Contiguous, rank-ordered split of the processed camera list."""
    base, rem = divmod(n_views_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
