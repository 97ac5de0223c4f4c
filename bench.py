#!/usr/bin/env python3
"""bench.py — Gaussian·views/s labelled by the MI355X majority-vote labeler.

One "step" = one full labelling pass of the hot path: every staged view (camera + segmentation map,
already resident in HBM as the library's u8 maps) is voted for every Gaussian and the int32 label of
each Gaussian is left in HBM.  Workload = BASELINE.json configs[2]: 3 M Gaussians, 200 views @1080p,
150 classes (+ label -1).  With N GPUs every rank votes its own 200 views (weak scaling: the job is
200*N views over the same 3 M Gaussians) and the per-Gaussian vote histogram is all-reduced (RCCL).

  python bench.py --gpus 1 --steps 5 --warmup 2
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  See DESIGN.md §6 for the byte model behind "roofline".
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
    ap.add_argument("--views", type=int, default=200, help="views per GPU")
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--classes", type=int, default=150)
    ap.add_argument("--seg-cell", type=int, default=4,
                    help="synthetic maps: Voronoi regions evaluated on a grid of this many pixels (4 = the round's workload; "
                         "1 = pixel-accurate boundaries, the hard case for the coarse map level)")
    ap.add_argument("--cpu-threads", type=int, default=16, help="cap on CPU-baseline OpenMP threads (box CPU share)")
    ap.add_argument("--cpu-sample", type=int, default=3_000_000, help="Gaussians in the CPU-baseline sample (0 = skip)")
    ap.add_argument("--order", default="input", choices=["input", "morton_host"],
                    help="experiment: pre-sort the Gaussians on the host before upload")
    ap.add_argument("--render-views", type=int, default=4, help="rasterizer leg on rank 0 at N=1: views to render (0 = skip)")
    ap.add_argument("--render-splats", type=int, default=3_000_000)
    ap.add_argument("--exchange", default="sparse", choices=["sparse", "a2a", "allreduce"],
                    help="multi-GPU protocol: counts-only all-to-all + sparse tie pass (v3), all-to-all of both planes "
                         "(v2) or all-reduce of the histogram (v1)")
    ap.add_argument("--force-exchange-path", action="store_true",
                    help="N=1 only: run the multi-GPU code path (planes kernel + slab reduce) on one GPU to time its kernels")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend; gloo + several ranks on one GPU is a functional rehearsal only")
    ap.add_argument("--opt", action="append", default=[], help="library tuning option name=value (gsx_set_option)")
    ap.add_argument("--no-profile", action="store_true", help="skip the HIP-event kernel timing")
    return ap.parse_args()


def render_leg(pkg, ctx, args, W, H):
    """Forward rasterizer on the same kind of scene (SH degree 3, viewer camera convention): views/s,
    per-kernel HIP-event times and the blend kernel's algorithmic bytes / time."""
    scene = pkg.scene
    n = args.render_splats
    seed = scene.BASE_SEED + 3
    xyz = scene.make_positions(n, seed)
    a = scene.make_splat_attributes(n, seed, sh_degree=3)
    cams = scene.make_cameras(max(args.render_views, 8), W, H, convention="c2w")[:args.render_views]
    ctx.upload_splats(xyz, a["scale"], a["rot"], a["opacity"], a["f_dc"])
    ctx.upload_sh(a["f_rest"], 3)
    ctx.render_view(cams[0], W, H, to_host=False)          # warm-up
    ctx.profile(True)
    t0 = time.perf_counter()
    pairs = consumed = 0
    for cam in cams:
        ctx.render_view(cam, W, H, to_host=False)
        pairs += ctx.render_num_pairs()
        consumed += ctx.render_num_pairs_consumed()
    ctx.synchronize()
    dt = time.perf_counter() - t0
    names = ["render_sh", "render_depth", "render_preprocess", "scan", "render_emit", "radix_hist", "radix_rowscan",
             "radix_scatter", "render_ranges", "render_blend"]
    k_ms = {k: round(ctx.profile_get(k)[1] / len(cams), 4) for k in names}
    ctx.profile(False)
    P = pairs / len(cams)
    blend_ms = k_ms["render_blend"]
    # blend: 4 B sorted index + 40 B record per (tile, splat) pair it actually reads + 16 B/pixel out
    Pc = consumed / len(cams)
    alg = Pc * 44.0 + W * H * 16.0
    achieved = alg / (blend_ms * 1e-3) / 1e9 if blend_ms > 0 else None
    return {"views": len(cams), "splats": n, "sh_degree": 3, "width": W, "height": H,
            "views_per_s": round(len(cams) / dt, 2), "gaussian_views_per_s": round(n * len(cams) / dt, 1),
            "tile_splat_pairs_per_view": int(P), "pairs_consumed_per_view": int(Pc), "kernel_ms_per_view": k_ms,
            "blend_roofline": {"bound": "hbm", "achieved": None if achieved is None else round(achieved, 2), "peak": HBM_PEAK_GBS,
                               "unit": "GB/s", "frac": None if achieved is None else round(achieved / HBM_PEAK_GBS, 5),
                               "algorithmic_bytes": int(alg),
                               "note": "blend is VALU/exp-bound at 16x16 tiles (SURVEY 7.6); see DESIGN.md"}}


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

    device = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(device)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", device))
        else:
            dist.init_process_group("gloo")

    n, V, W, H = args.gaussians, args.views, args.width, args.height
    total_views = V * world
    first = rank * V

    # ---- inputs -> HBM (outside the timed region) --------------------------------------------------
    t0 = time.time()
    pos = scene.make_positions(n, scene.BASE_SEED + 3)
    if args.order == "morton_host":
        q = ((pos - pos.min(0)) / (pos.max(0) - pos.min(0)) * 1023.0).astype(np.uint32)
        def spread(v):
            v = (v | (v << 16)) & 0x030000FF
            v = (v | (v << 8)) & 0x0300F00F
            v = (v | (v << 4)) & 0x030C30C3
            return (v | (v << 2)) & 0x09249249
        code = spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2)
        pos = np.ascontiguousarray(pos[np.argsort(code, kind="stable")])
    cams_all = scene.make_cameras(total_views, W, H, convention="w2c")
    ctx = pkg.Context(device)
    for kv in args.opt:
        k, v = kv.split("=")
        ctx.set_option(k, int(v))
    use_a2a = (world > 1 or args.force_exchange_path) and args.exchange in ("a2a", "sparse") and V <= 255
    use_sparse = use_a2a and args.exchange == "sparse"
    if use_a2a:
        pkg.dist.configure_a2a(ctx, world)
    ctx.upload_positions(pos)
    ctx.vote_begin(args.classes, first, total_views)
    host_segs = []
    keep_host = rank == 0 and world == 1 and args.cpu_sample > 0
    ingest_s = 0.0
    for v in range(V):
        seg = scene.make_segmap(H, W, args.classes, 3000 + first + v, cell=args.seg_cell)
        t1 = time.perf_counter()
        ctx.vote_view(cams_all[first + v], seg)          # host int32 map -> PCIe -> u8 tiles in HBM (synchronous)
        ingest_s += time.perf_counter() - t1
        if keep_host:
            host_segs.append(seg)
    ctx.synchronize()
    setup_s = time.time() - t0

    shard = (pkg.dist.GpuSparseShard(ctx) if use_sparse else pkg.dist.GpuSlabShard(ctx)) if use_a2a else pkg.dist.GpuVoteShard(ctx)

    def step():
        ctx.vote_rewind()
        if world == 1 and not args.force_exchange_path:
            ctx.vote_finalize(to_host=False)      # fused kernel -> int32 labels in HBM
        elif use_sparse:
            pkg.dist.exchange_labels_sparse(shard, to_host=False)
        elif use_a2a:
            pkg.dist.exchange_labels_a2a(shard, to_host=False)
        else:
            pkg.dist.exchange_labels(shard, to_host=False)

    def fence():
        ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ctx.vote_culled(reset=True)
    if not args.no_profile:
        ctx.profile(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = n * total_views / (elapsed / args.steps)
    # share of the (wave of 64 Gaussians, view) pairs the kernels skipped as provably invisible (wave culling)
    culled_frac = ctx.vote_culled() / (args.steps * ((n + 63) // 64) * V) if n and V else 0.0

    # ---- dominant kernel: HIP-event time on the ctx stream, algorithmic bytes / time ------------------
    kname = "vote_fused_labels" if (world == 1 and not args.force_exchange_path) else ("vote_fused_counts" if use_sparse else "vote_fused_planes")
    roofline = None
    vis_frac = None
    if rank == 0:
        # visible fraction of the (Gaussian, view) pairs, from the device projection kernel itself
        probe = list(range(0, V, max(1, V // 10)))
        vis = [float((ctx.project_all(cams_all[first + v])[0] >= 0).mean()) for v in probe]
        vis_frac = float(np.mean(vis))
    if rank == 0 and not args.no_profile:
        launches, total_ms = ctx.profile_get(kname)
        if launches:
            k_ms = total_ms / launches
            n_vis = vis_frac * n * V
            if kname == "vote_fused_labels":
                # positions once (12 B) + one u8 seg gather per visible pair + int32 label
                alg = 12.0 * n + 1.0 * n_vis + 4.0 * n + 192.0 * V
            else:
                esz = 1 if (use_a2a or total_views <= 255) else 2
                planes = 1.0 if use_sparse else 2.0
                alg = 12.0 * n + 1.0 * n_vis + planes * esz * (args.classes + 1) * n + 192.0 * V
            achieved = alg / (k_ms * 1e-3) / 1e9
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "traffic.json")
            if os.path.exists(tpath):
                try:
                    traffic = json.load(open(tpath)).get(kname)
                except Exception:
                    traffic = None
            roofline = {"bound": "hbm", "kernel": kname, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                        "kernel_ms": round(k_ms, 4), "launches": launches, "algorithmic_bytes": int(alg),
                        "note": "fp64-VALU bound kernel (vector ALUs ~72 % busy); HBM traffic 1.3x the algorithmic bytes; see DESIGN.md §6"}

    exchange_ms = None
    if rank == 0 and not args.no_profile and (world > 1 or args.force_exchange_path):
        exchange_ms = {}
        for k in ("vote_fused_counts", "vote_fused_planes", "vote_slab_totals", "vote_tie", "vote_tie_resolve", "vote_slab_reduce",
                  "vote_keys", "vote_labels"):
            cnt_k, ms_k = ctx.profile_get(k)
            if cnt_k:
                exchange_ms[k] = round(ms_k / cnt_k, 4)

    # ---- CPU baseline: the oracle (C port of the reference loop) on a bounded sample ---------------------
    cpu = None
    if keep_host:
        import oracle
        m = min(n, args.cpu_sample)
        sub = np.ascontiguousarray(pos[:m])
        sizes = [(W, H)] * V
        t0 = time.perf_counter()
        threads = min(oracle.max_threads(), len(os.sched_getaffinity(0)), args.cpu_threads)
        want = oracle.assign_labels(sub, cams_all[:V], host_segs, sizes, threads=threads)
        dt = time.perf_counter() - t0
        cores = oracle.assign_labels.threads_used
        ctx.vote_rewind()
        labels_gpu = ctx.vote_finalize(to_host=True)
        parity = bool(np.array_equal(labels_gpu[:m], want))
        cpu = {"value": round(m * V / dt, 1), "unit": "Gaussian·views/s", "cores": int(cores), "kind": "port",
               "sample": f"first {m} Gaussians x {V} views @{W}x{H} (same scene), oracle/vote_oracle.c, OpenMP",
               "seconds": round(dt, 2), "labels_match_gpu": parity}

    # ---- rasterizer leg (reported beside the headline metric, never part of `value`) ---------------------
    render = None
    if rank == 0 and world == 1 and args.render_views > 0:
        render = render_leg(pkg, ctx, args, W, H)

    if rank == 0:
        out = {
            "metric": "Gaussians·views/sec labelled (3M G, 1080p), majority vote",
            "value": round(value, 1), "unit": "Gaussian·views/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"{n} Gaussians x {V} views/GPU @{W}x{H}, {args.classes} classes + (-1), "
                                   f"BASELINE configs[{2 if world == 1 else 3}]",
                       "gaussians": n, "views_per_gpu": V, "views_total": total_views, "width": W, "height": H,
                       "classes": args.classes, "seg_maps": f"Voronoi, 400 sites, evaluated on a {args.seg_cell}-px grid",
                       "parallelism": f"views sharded x{world}",
                       "exchange": None if world == 1 else ("all_to_all(counts) + sparse tie pass + all_gather(labels)" if use_sparse else
                                                            "all_to_all + slab arg-max + all_gather(labels)" if use_a2a else
                                                            "all_reduce(SUM) of the histogram + all_reduce(MAX) of tie keys"),
                       "camera_convention": "w2c (labeler's R@(x-p) looks at the scene)",
                       "visible_fraction": None if vis_frac is None else round(vis_frac, 4),
                       "wave_views_culled_fraction": round(culled_frac, 4),
                       "setup_seconds": round(setup_s, 1),
                       "host_map_ingest_ms_per_view": round(ingest_s / max(1, V) * 1e3, 3),
                       "pcie_inclusive_value": round(n * total_views / (ingest_s + elapsed / args.steps), 1) if world == 1 else None},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "render": render,
            "exchange_kernels_ms": exchange_ms,
        }
        print(json.dumps(out), flush=True)
    ctx.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
