#!/usr/bin/env python3
"""bench.py — Gaussian·views/s labelled by the MI355X majority-vote labeler, on SURVEY.md §8(d)'s definition:

    value = N·V / wall time from the first `vote_view` submit to `labels_out` ready on the host

One "step" = one complete labelling run of the hot path (deep_learning_segmentation.py:255-308): gsx_vote_begin,
one gsx_vote_view per view handing over a HOST int32 segmentation map (pageable numpy memory, as the reference's
`segment_image` returns it), the fused vote kernel, arg-max, int32 labels copied into a host array.  Positions are
resident (PLY parsing is outside the metric).  Workload = BASELINE.json configs[2]: 3 M Gaussians, 200 views @1080p,
150 classes (+ label -1), Voronoi maps with pixel-accurate boundaries.

With --gpus N the SAME 200 views are sharded over the ranks (BASELINE configs[3]; strong scaling): every rank ingests
its block of views, the packed maps are all-gathered (chunk by chunk, overlapped with the hand-over: dist.GatherPipeline),
every rank votes its slab of the Gaussians, the labels are all-gathered (protocol v4).  --weak-views V gives every rank V
views instead.

Side numbers at N=1 (never `value`): the same run with the int32 maps already resident in HBM (`resident_value`), and
the vote kernel alone over maps already packed in HBM (`kernel_resident_value`, round 1's headline).

  python bench.py --gpus 1 --steps 5 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  See DESIGN.md §5-§6 for the byte models behind "roofline".
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec peak


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--gaussians", type=int, default=3_000_000)
    ap.add_argument("--views", type=int, default=200, help="views in TOTAL, sharded over the GPUs (BASELINE configs[2]/[3])")
    ap.add_argument("--weak-views", type=int, default=0, help="weak scaling instead: this many views PER GPU")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--classes", type=int, default=150)
    ap.add_argument("--seg-cell", type=int, default=1,
                    help="synthetic maps: Voronoi regions evaluated on a grid of this many pixels (1 = pixel-accurate boundaries, "
                         "SURVEY 8d's generator; 4 = constant 4x4 blocks, the best case for the coarse map level)")
    ap.add_argument("--cpu-threads", type=int, default=16, help="cap on CPU-baseline OpenMP threads (box CPU share)")
    ap.add_argument("--cpu-sample", type=int, default=3_000_000, help="Gaussians in the all-core CPU-baseline sample (0 = skip)")
    ap.add_argument("--render-views", type=int, default=4, help="rasterizer leg on rank 0 at N=1: views to render (0 = skip)")
    ap.add_argument("--render-splats", type=int, default=3_000_000)
    ap.add_argument("--exchange", default="pipelined", choices=["pipelined", "gather", "sparse", "a2a", "allreduce"],
                    help="multi-GPU protocol: all-gather of the packed maps + Gaussian slabs (v4) with the all-gather overlapped "
                         "with the hand-over (GatherPipeline, default) or in one piece (gather), counts-only all-to-all + sparse "
                         "tie pass (v3), all-to-all of both planes (v2) or all-reduce of the histogram (v1)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend; gloo + several ranks on one GPU is a functional rehearsal only")
    ap.add_argument("--opt", action="append", default=[], help="library tuning option name=value (gsx_set_option)")
    ap.add_argument("--side-steps", type=int, default=3, help="steps of each side measurement at N=1 (0 = skip)")
    ap.add_argument("--no-profile", action="store_true", help="skip the HIP-event kernel timing")
    ap.add_argument("--no-numa-bind", action="store_true",
                    help="do not bind this process to the CPUs of its GPU's NUMA node (one process per GPU: maps, pinned staging and "
                         "packer threads next to the GPU's PCIe root)")
    return ap.parse_args()


def _cpu_throttle():
    """(periods, throttled periods, throttled microseconds) of this process's cgroup, or None: the GPU box gives a rank a CPU
    quota; a step that ran into it was frozen for the rest of a 100 ms scheduler period."""
    try:
        kv = dict(line.split() for line in open("/sys/fs/cgroup/cpu.stat"))
        return int(kv["nr_periods"]), int(kv["nr_throttled"]), int(kv["throttled_usec"])
    except (OSError, KeyError, ValueError):
        return None


def make_segmaps(scene, torch, device, H, W, classes, seeds, cell):
    """Voronoi class maps (400 sites, classes U{-1..C-1}) as HOST int32 arrays.  cell > 1: the package's numpy/scipy
    generator.  cell == 1 (pixel-accurate): the same construction evaluated on the GPU (2 M pixels x 400 sites per
    map is seconds of KD-tree queries per map on the host); the resulting host arrays are what BOTH the GPU run and
    the CPU baseline consume."""
    if cell > 1:
        return [scene.make_segmap(H, W, classes, s, cell=cell) for s in seeds]
    return [scene.make_segmap_gpu(torch, device, H, W, classes, s) for s in seeds]


def render_leg(pkg, ctx, args, W, H):
    """Forward rasterizer on the same kind of scene (SH degree 3, viewer camera convention): views/s through
    gsx_render_views (four frames in flight on HIP streams of their own inside one context) and one frame at a time, both timed
    WITHOUT the per-kernel events (they serialise the launches); then per-kernel HIP-event times and the blend kernel's
    algorithmic bytes / time."""
    scene = pkg.scene
    n = args.render_splats
    seed = scene.BASE_SEED + 3
    xyz = scene.make_positions(n, seed)
    a = scene.make_splat_attributes(n, seed, sh_degree=3)
    cams = scene.make_cameras(max(args.render_views, 8), W, H, convention="c2w")[:args.render_views]
    ctx.upload_splats(xyz, a["scale"], a["rot"], a["opacity"], a["f_dc"])
    ctx.upload_sh(a["f_rest"], 3)
    ctx.render_view(cams[0], W, H, to_host=False)          # warm-up (sizes the pair buffers)
    ctx.render_views((cams * 8)[:8], W, H, to_host=False)  # ... of the further streams too (up to 6 frames in flight)
    reps = 3
    t0 = time.perf_counter()
    for _ in range(reps):
        for cam in cams:
            ctx.render_view(cam, W, H, to_host=False)
    dt_one = (time.perf_counter() - t0) / reps
    reps_many = 2 * reps                                   # per call: the streams' host threads start and end once
    ctx.render_views(cams * reps_many, W, H, to_host=False)   # warm-up of the same shape
    t0 = time.perf_counter()
    for _ in range(2):
        ctx.render_views(cams * reps_many, W, H, to_host=False)
    dt = (time.perf_counter() - t0) / (2 * reps_many)
    ctx.profile(True)
    pairs = consumed = 0
    for cam in cams:
        ctx.render_view(cam, W, H, to_host=False)
        pairs += ctx.render_num_pairs()
        consumed += ctx.render_num_pairs_consumed()
    k_ms = {}
    for k in ctx.profile_names():
        if k.startswith("render_") or k.startswith("radix_") or k == "scan":
            cnt_k, ms_k = ctx.profile_get(k)
            if cnt_k:
                k_ms[k] = round(ms_k / len(cams), 4)
    ctx.profile(False)
    P = pairs / len(cams)
    blend_ms = k_ms.get("render_blend", 0.0)
    # blend: 4 B sorted index + 40 B record per (tile, splat) pair it actually reads + 16 B/pixel out
    Pc = consumed / len(cams)
    alg = Pc * 44.0 + W * H * 16.0
    achieved = alg / (blend_ms * 1e-3) / 1e9 if blend_ms > 0 else None
    return {"views": len(cams), "splats": n, "sh_degree": 3, "width": W, "height": H,
            "views_per_s": round(len(cams) / dt, 2), "gaussian_views_per_s": round(n * len(cams) / dt, 1),
            "views_per_s_note": "gsx_render_views: four frames in flight on four HIP streams of one context (option render_frames)",
            "views_per_s_one_frame_at_a_time": round(len(cams) / dt_one, 2),
            "tile_splat_pairs_per_view": int(P), "pairs_consumed_per_view": int(Pc),
            "pairs_note": "pairs binned, sorted and ranged per view: (32x32-pixel bin, splat) with the mask of the bin's tiles in the value (option render_bin32), "
                          "splats without a rectangle left out by the level-1 sort (render_compact); consumed = records a tile's wave blended",
            "kernel_ms_per_view": k_ms,
            "kernel_ms_sum_per_view": round(sum(k_ms.values()), 4),
            "blend_roofline": {"bound": "hbm", "achieved": None if achieved is None else round(achieved, 2), "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": None if achieved is None else round(achieved / HBM_PEAK_GBS, 5),
                               "algorithmic_bytes": int(alg),
                               "note": "blend is fp32-VALU bound at 16x16 tiles (counters: profiles/r03/derived_r03d.json: traffic = 1.5 x these bytes); see DESIGN.md"}}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    args = parse()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch N>1 with torch.distributed.run")

    import torch
    import torch.distributed as dist
    pkg = importlib.import_module("3d_gaussian_splatting_project_amd")
    scene = pkg.scene
    Camera = pkg.Camera

    device = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(device)
    bound = None if args.no_numa_bind else pkg.bind_to_gpu_numa_node(device)
    # GSX_DIST_FORCE_COLLECTIVES=1 with one GPU: rehearse the N > 1 code path (process group, exchange protocol, fences, the
    # max-over-ranks reduction) through real RCCL with a one-rank group; the JSON line says so
    rehearsal = world == 1 and os.environ.get("GSX_DIST_FORCE_COLLECTIVES") == "1"
    if world > 1 or rehearsal:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29655")
        if rehearsal:
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo")

    n, W, H = args.gaussians, args.width, args.height
    weak = args.weak_views > 0
    total_views = args.weak_views * world if weak else args.views
    first, last = pkg.dist.view_range(total_views, rank, world)
    V = last - first                                   # this rank's views

    # ---- inputs (outside the timed region): positions -> HBM, cameras, HOST segmentation maps -------------------------
    t0 = time.time()
    pos = scene.make_positions(n, scene.BASE_SEED + 3)
    cams_all = scene.make_cameras(total_views, W, H, convention="w2c")
    cam_structs = [Camera.from_dict(c) for c in cams_all[first:last]]
    cams_all_arr = pkg.camera_array(cams_all)      # every rank holds the whole camera list (cameras.json)
    host_segs = make_segmaps(scene, torch, device, H, W, args.classes, [3000 + first + v for v in range(V)], args.seg_cell)
    ctx = pkg.Context(device)
    for kv in args.opt:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    mode = args.exchange if (world > 1 or rehearsal) else None
    multi = mode is not None
    if mode in ("a2a", "sparse"):
        if max(b - a for a, b in (pkg.dist.view_range(total_views, r, world) for r in range(world))) > 255:
            raise SystemExit("--exchange a2a/sparse keep 8-bit per-rank counters: at most 255 views per rank")
        pkg.dist.configure_a2a(ctx, world)
    ctx.upload_positions(pos)
    setup_s = time.time() - t0
    labels_buf = np.empty(n, np.int32)

    shard = None
    if multi:
        shard = {"gather": pkg.dist.GpuGatherShard, "pipelined": pkg.dist.GpuGatherShard, "sparse": pkg.dist.GpuSparseShard,
                 "a2a": pkg.dist.GpuSlabShard, "allreduce": pkg.dist.GpuVoteShard}[mode](ctx)
    exchange = {"gather": pkg.dist.exchange_labels_gather, "sparse": pkg.dist.exchange_labels_sparse,
                "a2a": pkg.dist.exchange_labels_a2a, "allreduce": pkg.dist.exchange_labels}.get(mode)

    def step(timing=False):
        """The metric's span: first vote_view submit -> labels on the host."""
        ctx.vote_begin(args.classes, first, total_views)
        if mode == "pipelined":
            # one camera model: every rank derives all view descriptors from the shared camera list (no header exchange)
            pipe = pkg.dist.GatherPipeline(shard, total_views, cameras=cams_all_arr, map_size=(W, H), timing=timing)
            for v in range(V):
                ctx.vote_view(cam_structs[v], host_segs[v])
                pipe.after_view()
            pipe.finish(out=labels_buf)
            return pipe.phases_ms
        for v in range(V):
            ctx.vote_view(cam_structs[v], host_segs[v])
        if not multi:
            ctx.vote_finalize(out=labels_buf)
        else:
            exchange(shard, out=labels_buf)

    def fence():
        ctx.synchronize()
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
            torch.cuda.synchronize()

    # Everything that changes the process's state happens BEFORE the warm-up, so the warm-up steps run exactly as the timed ones do
    # (r03's records show the FIRST timed step at 11-17 ms among 8 ms ones without any cgroup throttling when the per-kernel events
    # were switched on and the set-up's garbage collected between the warm-up and the timed region).  The collector itself stays on.
    import gc
    gc.collect()
    if not args.no_profile:
        ctx.profile(True)
    for _ in range(args.warmup):
        step()
    fence()
    ctx.vote_culled(reset=True)
    if not args.no_profile:
        ctx.profile(True)   # drops the warm-up's figures (the events are recycled, nothing is created or freed)
    thr0 = _cpu_throttle()
    step_ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts = time.perf_counter()
        step()
        step_ms.append((time.perf_counter() - ts) * 1e3)   # host-side view of each step (the LAST one ends at the fence below)
    fence()
    elapsed = time.perf_counter() - t0
    thr1 = _cpu_throttle()
    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = n * total_views / (elapsed / args.steps)
    labels_main = labels_buf.copy()
    # ---- N > 1: where a step's time goes.  Extra steps OUTSIDE the timed region (the events' host waits would perturb it):
    # per phase the median over the steps, then the MAX over the ranks; per-rank hand-over rate of the int32 maps. ------------
    phases = None
    if mode == "pipelined":
        rows, walls = [], []
        for _ in range(max(3, min(args.steps, 7))):
            fence()
            ts = time.perf_counter()
            ph = step(timing=True)
            walls.append((time.perf_counter() - ts) * 1e3)
            if ph:
                rows.append(ph)
        keys = ["hand_over", "gathers_exposed", "import_and_slab_vote", "labels_all_gather", "labels_to_host"]
        if rows and all(all(r.get(k) is not None for k in keys) for r in rows):
            med = [float(np.median([r[k] for r in rows])) for k in keys] + [float(np.median(walls))]
            t = torch.tensor(med, dtype=torch.float64, device="cuda" if args.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            mx = [float(x) for x in t.tolist()]
            raw_bytes = float(sum(s_.nbytes for s_ in host_segs))
            mine = torch.tensor([raw_bytes / (med[0] * 1e-3) / 1e9 if med[0] > 0 else 0.0], dtype=torch.float64, device=t.device)
            alls = [torch.zeros_like(mine) for _ in range(world)]
            dist.all_gather(alls, mine)
            phases = {k: round(v, 4) for k, v in zip(keys, mx[:-1])}
            phases["sum"] = round(sum(mx[:-1]), 4)
            phases["step_ms_of_these_steps"] = round(mx[-1], 4)
            phases["note"] = ("max over ranks of each rank's median over extra steps run after the timed region: hand_over = first "
                              "vote_view -> last one returned (host; the chunk all_gathers of the earlier views run underneath); then on "
                              "the ctx stream: chunk all_gathers still running at that point, import + vote of this rank's slab, "
                              "all_gather of the slab labels; labels_to_host = D2H (one byte per label) + widening + Python")
            phases["host_ingest_GBps_of_int32_maps_per_rank"] = [round(float(a.item()), 2) for a in alls]

    labels_check = None
    if mode in ("gather", "pipelined"):
        # after the import every rank holds every view: a plain single-GPU vote of ALL Gaussians on this rank must give
        # the labels the exchange assembled from the ranks' slabs (rank-local, no collective)
        labels_check = bool(np.array_equal(ctx.vote_finalize(), labels_main))

    # ---- dominant kernel: HIP-event time on the ctx stream over the timed region, algorithmic bytes / time -----------
    kname = {None: "vote_fused_labels", "gather": "vote_fused_labels", "pipelined": "vote_fused_labels",
             "sparse": "vote_fused_counts"}.get(mode, "vote_fused_planes")
    n_slab = n if mode not in ("gather", "pipelined") else (n + world - 1) // world
    V_kernel = total_views if mode in (None, "gather", "pipelined") else V
    roofline = None
    kernels_ms = None
    vis_frac = None
    if rank == 0:
        # visible fraction of the (Gaussian, view) pairs, from the device projection kernel itself
        probe = list(range(0, total_views, max(1, total_views // 10)))
        vis = [float((ctx.project_all(cams_all[v])[0] >= 0).mean()) for v in probe]
        vis_frac = float(np.mean(vis))
    early_views = ctx.vote_early_views() if not multi else 0   # option early_vote: the last run's first views were voted while the rest was handed over
    if rank == 0 and not args.no_profile:
        kernels_ms = {}
        for k in ctx.profile_names():
            cnt_k, ms_k = ctx.profile_get(k)
            if cnt_k:
                kernels_ms[k] = {"launches_per_step": round(cnt_k / args.steps, 2), "ms_per_launch": round(ms_k / cnt_k, 4)}
        bins = args.classes + 1
        n_pad = (n + 255) // 256 * 256

        def alg_bytes(name):
            """algorithmic HBM bytes of one launch"""
            if name == "vote_fused_labels":   # positions once (12 B) + one u8 map lookup per visible pair + int32 label
                return 12.0 * n_slab + 1.0 * vis_frac * n_slab * V_kernel + 4.0 * n_slab + 192.0 * V_kernel
            if name == "vote_early_planes":   # + count and first-view planes and the vote record written (u8 each)
                return 12.0 * n + 1.0 * vis_frac * n * early_views + (2.0 * bins + early_views) * n_pad + 192.0 * early_views
            if name == "vote_early_record":   # the early stage of the record-and-replay form: + one record byte per Gaussian and early view
                return 12.0 * n + 1.0 * vis_frac * n * early_views + 1.0 * early_views * n_pad + 192.0 * early_views
            if name == "vote_fused_replay":   # its last stage: the record read back, the remaining views walked, the int32 label
                rest = total_views - early_views
                return 12.0 * n + 1.0 * vis_frac * n * rest + 1.0 * early_views * n_pad + 4.0 * n + 192.0 * rest
            if name == "vote_fused_final":    # + both planes read, one record byte and the int32 label per Gaussian
                rest = total_views - early_views
                return 12.0 * n + 1.0 * vis_frac * n * rest + 2.0 * bins * n_pad + 5.0 * n + 192.0 * rest
            esz = 1 if (mode in ("a2a", "sparse") or total_views <= 255) else 2
            planes = 1.0 if mode == "sparse" else 2.0
            return 12.0 * n + 1.0 * vis_frac * n_slab * V_kernel + planes * esz * bins * n + 192.0 * V

        def roofline_of(name):
            launches, total_ms = ctx.profile_get(name)
            if not launches:
                return None
            k_ms = total_ms / launches
            alg = alg_bytes(name)
            achieved = alg / (k_ms * 1e-3) / 1e9
            cached, stale = {}, None
            tpath = os.path.join(ROOT, "profiles", "counters.json")   # PMC figures of the committed rocprofv3 runs
            if os.path.exists(tpath):
                try:
                    allc = json.load(open(tpath))
                    cached = allc.get(name, {})
                    # the figures belong to ONE build of the kernels and ONE early-vote split: quote them only for those
                    import hashlib
                    src = os.path.join(ROOT, "3d_gaussian_splatting_project_amd", "csrc", "vote.hip")
                    sha = hashlib.sha256(open(src, "rb").read()).hexdigest()[:16]
                    st = allc.get("_stamp", {})
                    if st.get("vote_hip_sha16") != sha:
                        stale = f"profiles/counters.json was taken on another build of csrc/vote.hip ({st.get('vote_hip_sha16')} != {sha})"
                    elif name != "vote_fused_labels" and abs((st.get("early_views") or -99) - int(early_views)) > 0.15 * total_views:
                        stale = f"profiles/counters.json was taken with {st.get('early_views')} early views, this run chose {int(early_views)}"
                    elif name != "vote_fused_labels":
                        cached = dict(cached, source=f"{cached.get('source')} (taken with {st.get('early_views')} early views under the profiler; this run: {int(early_views)})")
                except Exception:
                    cached = {}
            if stale:
                cached = {"stale": stale}
            return {"bound": "hbm", "kernel": name, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                    "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5),
                    "traffic": cached.get("hbm_bytes_per_launch"),
                    "traffic_source": cached.get("stale") or (cached.get("source", None) and f"cached profile figure, not measured in this run: {cached.get('source')}"),
                    "kernel_ms": round(k_ms, 4), "launches": launches, "algorithmic_bytes": int(alg),
                    "second_bound": cached.get("valu_f64")}

        if early_views and mode is None:
            # the run's vote is two kernels: the early stage (second stream, hidden behind the hand-over) and the last stage
            # (between the last map and the labels).  The dominant one by GPU time carries the roofline, the other rides along.
            cand = [r for r in (roofline_of("vote_early_planes"), roofline_of("vote_fused_final"), roofline_of("vote_early_record"),
                                roofline_of("vote_fused_replay")) if r]
            cand.sort(key=lambda r: -r["kernel_ms"])
            roofline = cand[0] if cand else None
            if roofline:
                roofline["other_kernels"] = cand[1:]
                roofline["early_views"] = int(early_views)
                roofline["note"] = ("the vote of a run = an early stage over the first early_views views on a second stream while the host hands "
                                    "over the rest (off the critical path: vote_early_record writes one byte per Gaussian and view; option "
                                    "early_replay=0: vote_early_planes) + a last stage over the remaining views (vote_fused_replay replays the "
                                    "record behind its own walk; vote_fused_final folds the planes in); all walk views with the filtered fp64 "
                                    "projection of the one-piece kernel, which is latency-, not bandwidth-bound (DESIGN.md sections 4, 8)")
        else:
            roofline = roofline_of(kname)
            if roofline:
                roofline["note"] = ("fp64-VALU bound kernel: the HBM fraction is reported as the contract asks, the binding "
                                    "resource is the vector ALU issue rate (second_bound, from PMC counters); DESIGN.md section 8")
        ctx.profile(False)
    culled_frac = ctx.vote_culled() / (max(1, args.steps) * ((n_slab + 63) // 64) * max(1, V_kernel))

    # ---- side measurements at N=1: int32 maps resident in HBM; packed maps resident (kernel only) ------------------------
    side = {}
    if not multi and rank == 0 and args.side_steps > 0:
        S = args.side_steps
        dev_maps = torch.from_numpy(np.stack(host_segs)).cuda()          # (V, H, W) int32 in HBM
        torch.cuda.synchronize()

        def step_resident():
            ctx.vote_begin(args.classes, 0, total_views)
            ctx.vote_views_device(cam_structs, dev_maps)
            ctx.vote_finalize(out=labels_buf)
        step_resident()
        ctx.profile(True)
        t1 = time.perf_counter()
        for _ in range(S):
            step_resident()
        dt = (time.perf_counter() - t1) / S
        pk_cnt, pk_ms = ctx.profile_get("seg_pack")
        ctx.profile(False)
        side["resident_value"] = round(n * total_views / dt, 1)
        side["resident_ms_per_step"] = round(dt * 1e3, 4)
        side["resident_note"] = "int32 maps already in HBM (a segmentation network on the same GPU): pack + vote + labels to the host"
        side["device_map_pack_us_per_map"] = round(pk_ms / max(1, S * V) * 1e3, 3)
        side["resident_labels_equal"] = bool(np.array_equal(labels_buf, labels_main))
        del dev_maps

        def step_kernel():
            ctx.vote_rewind()
            ctx.vote_finalize(to_host=False)
        step_kernel()
        t1 = time.perf_counter()
        for _ in range(4 * S):
            step_kernel()
        ctx.synchronize()
        dt = (time.perf_counter() - t1) / (4 * S)
        side["kernel_resident_value"] = round(n * total_views / dt, 1)
        side["kernel_resident_ms_per_step"] = round(dt * 1e3, 4)
        side["kernel_resident_note"] = "packed maps resident, labels left in HBM: the vote kernel alone (round 1's headline)"

        # host hand-over alone (no vote): how fast the maps get from pageable host memory into the pool
        ctx.vote_begin(args.classes, 0, total_views)
        t1 = time.perf_counter()
        for v in range(V):
            ctx.vote_view(cam_structs[v], host_segs[v])
        t_submit = time.perf_counter() - t1
        ctx.synchronize()
        t_all = time.perf_counter() - t1
        raw = float(sum(s.nbytes for s in host_segs))
        side["host_ingest"] = {"ms_per_map_submit": round(t_submit / V * 1e3, 4), "ms_per_map_done": round(t_all / V * 1e3, 4),
                               "effective_GBps_of_int32_maps": round(raw / t_all / 1e9, 2), "int32_bytes": int(raw),
                               "host_threads": ctx.host_threads(), "pcie_bytes": ctx.vote_link_bytes(),
                               "pcie_note": "bytes of the maps that crossed the link: the compact form (coarse level + the 16-byte blocks of "
                                            "the 4x4 cells whose pixels differ), expanded to the two-level pool form on the GPU"}
        ctx.vote_finalize(out=labels_buf)

        # 8-bit class images (GSX_SEG_U8_LABELS): what a caller hands over whose segmentation network writes uint8 class maps - a
        # quarter of the bytes the host pass has to read.  u8 holds 0 .. classes-1 only, so the maps' -1 pixels become class 0 here;
        # the labels are checked against a run over the SAME maps handed over as int32.
        if args.classes <= 255 and V <= 255:
            segs_u8 = [np.where(sg < 0, 0, sg).astype(np.uint8) for sg in host_segs]

            def step_u8(maps):
                ctx.vote_begin(args.classes, 0, total_views)
                for v in range(V):
                    ctx.vote_view(cam_structs[v], maps[v])
                ctx.vote_finalize(out=labels_buf)
            step_u8([m8.astype(np.int32) for m8 in segs_u8])
            want_u8 = labels_buf.copy()
            for _ in range(2):
                step_u8(segs_u8)
            ctx.synchronize()
            t1 = time.perf_counter()
            for _ in range(S + 2):
                step_u8(segs_u8)
            ctx.synchronize()
            dt = (time.perf_counter() - t1) / (S + 2)
            side["u8_class_images"] = {"value": round(n * total_views / dt, 1), "ms_per_step": round(dt * 1e3, 4),
                                       "map_bytes": int(sum(m8.nbytes for m8 in segs_u8)),
                                       "labels_equal_int32_run_on_the_same_maps": bool(np.array_equal(labels_buf, want_u8)),
                                       "note": "the metric's span with uint8 class images handed over instead of int32 maps (-1 pixels set to class 0): "
                                               "not the headline (SURVEY 8(d) hands over int32), the lever left in the hand-over"}
            del segs_u8

    # ---- CPU baseline: the oracle (C port of the reference loop) on bounded samples of the same workload ----------------
    cpu = None
    if rank == 0 and not multi and args.cpu_sample > 0:
        import oracle
        sizes = [(W, H)] * V
        cams_cpu = cams_all[:V]
        threads = min(oracle.max_threads(), len(os.sched_getaffinity(0)), args.cpu_threads)
        m = min(n, args.cpu_sample, max(20_000, int(4e9 // max(1, V))))   # bounded: <= ~30 s of CPU work whatever the shape
        sub = np.ascontiguousarray(pos[:m])
        t1 = time.perf_counter()
        want = oracle.assign_labels(sub, cams_cpu, host_segs, sizes, threads=threads)
        dt_all = time.perf_counter() - t1
        cores = int(oracle.assign_labels.threads_used)
        parity = bool(np.array_equal(labels_main[:m], want))
        m1 = max(1, min(m, m // max(1, threads)))                       # single thread: 1/threads of the sample
        t1 = time.perf_counter()
        want1 = oracle.assign_labels(np.ascontiguousarray(pos[:m1]), cams_cpu, host_segs, sizes, threads=1)
        dt_1 = time.perf_counter() - t1
        ref_py = None
        rpath = os.path.join(ROOT, "profiles", "r01", "cpu_reference_python.json")
        if os.path.exists(rpath):
            try:
                r = json.load(open(rpath))
                ref_py = {"value": r["gaussian_views_per_s"], "unit": "Gaussian·views/s", "cores": 1,
                          "sample": f"{r['gaussians']} Gaussians x {r['views']} views of this scene, measured in the BUILD container "
                                    "(the reference cannot travel to the GPU box)",
                          "extrapolation_factor_to_workload": round(n * total_views / (r["gaussians"] * r["views"]), 1),
                          "extrapolated_seconds_for_workload": round(n * total_views / r["gaussian_views_per_s"], 1),
                          "gpu_speedup": round(value / r["gaussian_views_per_s"], 1)}
            except Exception:
                ref_py = None
        cpu = {"value": round(m * V / dt_all, 1), "unit": "Gaussian·views/s", "cores": cores, "kind": "port",
               "sample": f"first {m} Gaussians x {V} views @{W}x{H} (same scene and maps), oracle/vote_oracle.c, OpenMP",
               "seconds": round(dt_all, 2), "labels_match_gpu": parity and bool(np.array_equal(want[:m1], want1)),
               "single_thread": {"value": round(m1 * V / dt_1, 1), "cores": 1, "seconds": round(dt_1, 2),
                                 "sample": f"first {m1} Gaussians x {V} views"},
               "cpu_model": cpu_model(), "cpus_usable": len(os.sched_getaffinity(0)),
               "gpu_speedup_all_cores": round(value / (m * V / dt_all), 1),
               "reference_python": ref_py}

    # ---- rasterizer leg (reported beside the headline metric, never part of `value`) ---------------------
    render = None
    if rank == 0 and not multi and args.render_views > 0:
        render = render_leg(pkg, ctx, args, W, H)

    if rank == 0:
        shape = (n, total_views, W, H, args.classes)
        cfg = {(3_000_000, 200, 1920, 1080, 150): "BASELINE configs[2]" if world == 1 else f"BASELINE configs[3]: the 200 views sharded over {world} GPUs",
               (500_000, 16, 1280, 720, 150): "BASELINE configs[1]" + ("" if world == 1 else f" sharded over {world} GPUs"),
               (10_000_000, 1000, 3840, 2160, 150): "BASELINE configs[4]" + (" on ONE GPU" if world == 1 else f": the 1000 views sharded over {world} GPUs")
               }.get(shape, "not a BASELINE config (custom shape)")
        out = {
            "metric": "Gaussians·views/sec labelled (3M G, 1080p), majority vote; first vote_view submit -> labels on the host",
            "value": round(value, 1), "unit": "Gaussian·views/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True,
            "scaling": "weak" if weak else "strong",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": (f"{n} Gaussians x {total_views} views in total @{W}x{H}, {args.classes} classes + (-1), "
                                    + (f"weak scaling: {args.weak_views} views per GPU (not a BASELINE config)" if weak else cfg)),
                       "timed_span": "gsx_vote_begin + one gsx_vote_view per view (host int32 map, pageable) + vote + arg-max + labels D2H "
                                     "into a host array (SURVEY 8d); positions resident",
                       "gaussians": n, "views_total": total_views, "views_this_rank": V, "width": W, "height": H,
                       "classes": args.classes,
                       "seg_maps": f"Voronoi, 400 sites, evaluated on a {args.seg_cell}-px grid" + (" (pixel-accurate boundaries)" if args.seg_cell == 1 else ""),
                       "parallelism": f"views sharded x{world}" + (", Gaussians sharded for the vote" if mode in ("gather", "pipelined") else ""),
                       "exchange": {None: None, "gather": "all_gather(packed u8 maps) + per-rank Gaussian slab vote + all_gather(labels)",
                                    "pipelined": "chunked all_gather(packed u8 maps) overlapped with the hand-over + per-rank Gaussian slab vote + all_gather(labels)",
                                    "sparse": "all_to_all(counts) + sparse tie pass + all_gather(labels)",
                                    "a2a": "all_to_all + slab arg-max + all_gather(labels)",
                                    "allreduce": "all_reduce(SUM) of the histogram + all_reduce(MAX) of tie keys"}[mode],
                       "camera_convention": "w2c (labeler's R@(x-p) looks at the scene)",
                       "visible_fraction": None if vis_frac is None else round(vis_frac, 4),
                       "wave_views_culled_fraction": round(culled_frac, 4),
                       "early_vote_views": int(early_views),   # voted on a second stream while the rest of the run was handed over (0: one-piece vote)
                       "rehearsal": "N > 1 code path with a ONE-rank RCCL group (GSX_DIST_FORCE_COLLECTIVES=1)" if rehearsal else None,
                       "bound_to_gpu_numa_node": None if bound is None else f"{len(bound)} CPUs",
                       "setup_seconds": round(setup_s, 1),
                       "step_ms_median_min_max": [round(float(np.median(step_ms)), 3), round(min(step_ms), 3), round(max(step_ms), 3)],
                       "step_ms": [round(x, 2) for x in step_ms],
                       "cpu_quota_throttling_in_timed_region": None if thr0 is None or thr1 is None else
                       {"scheduler_periods": thr1[0] - thr0[0], "throttled_periods": thr1[1] - thr0[1],
                        "throttled_ms": round((thr1[2] - thr0[2]) / 1e3, 1)},
                       "exchanged_labels_equal_single_gpu_vote": labels_check,
                       "rccl_world": (world if args.backend == "nccl" else None) if multi else None,
                       "collective_backend": args.backend if multi else None,
                       "phases_ms": phases,
                       "library_options": dict(kv.split("=") for kv in args.opt) or None},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "vs_cpu_baseline": None if cpu is None else cpu["gpu_speedup_all_cores"],
            "side": side or None,
            "kernels_ms": kernels_ms,
            "render": render,
        }
        print(json.dumps(out), flush=True)
    ctx.close()
    if multi:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
