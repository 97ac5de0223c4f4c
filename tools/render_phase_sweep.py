#!/usr/bin/env python3
"""GPU box: views/s of gsx_render_views (four frames in flight) by the depth-phase options (render_phases, render_phase_ratio,
exact_cull, tile_lpt), bench scene (3 M splats / 1080p / SH 3); pairs sorted / consumed per view alongside."""
import importlib, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = importlib.import_module("3d_gaussian_splatting_project_amd")
scene = pkg.scene
n, W, H = 3_000_000, 1920, 1080
if len(sys.argv) > 4:      # another shape: render_phase_sweep.py <mode> <splats> <width> <height>
    n, W, H = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
seed = scene.BASE_SEED + 3
xyz = scene.make_positions(n, seed)
a = scene.make_splat_attributes(n, seed, sh_degree=3)
cams = scene.make_cameras(24, W, H, convention="c2w")
with pkg.Context(0) as c:
    c.upload_splats(xyz, a["scale"], a["rot"], a["opacity"], a["f_dc"])
    c.upload_sh(a["f_rest"], 3)
    sweep = ({}, {"render_phases": 3}, {"render_phases": 3, "render_phase_ratio": 3}, {"render_phases": 2, "render_phase_ratio": 8},
             {"render_phases": 2, "render_phase_ratio": 3}, {"render_phases": 4, "render_phase_ratio": 3}, {"exact_cull": 1},
             {"render_phases": 3, "exact_cull": 1}, {"tile_lpt": 1}, {})
    if len(sys.argv) > 1 and sys.argv[1] == "ab":      # a clean A/B of the two candidates, interleaved three times
        sweep = ({}, {"render_phases": 3, "render_phase_ratio": 3}, {"render_phases": 3}, {"render_phases": 3, "render_phase_ratio": 5},
                 {"render_phases": 4, "render_phase_ratio": 3}, {"render_phases": 4, "render_phase_ratio": 2}, {"tile_lpt": 1}) * 2 + ({},)
    if len(sys.argv) > 1 and sys.argv[1] == "blend":   # blend kernels: 1 = two pixels per thread, 2 = four (one wave per tile)
        sweep = ({"blend_pk2": 1}, {"blend_pk2": 2}) * 3
        for mode in (1, 2):
            c.set_option("blend_pk2", mode)
            c.render_view(cams[0], W, H, to_host=False)
            c.profile(True)
            for cam in cams[:8]:
                c.render_view(cam, W, H, to_host=False)
            cnt, ms = c.profile_get("render_blend")
            c.profile(False)
            print(f"blend_pk2={mode}: render_blend {ms / 8:.4f} ms per view ({cnt} launches)", flush=True)
    if len(sys.argv) > 1 and sys.argv[1] == "bin32":   # depth phases again, now that a pair costs less (32x32-pixel bins); per-kernel times first
        sweep = ({}, {"render_phases": 1}, {"render_phases": 3, "render_phase_ratio": 3}, {"render_phases": 2, "render_phase_ratio": 8},
                 {"render_phases": 2, "render_phase_ratio": 2}, {"render_phases": 3}, {"render_bin32": 0}) * 2 + ({},)
        for b in (0, 1):
            c.set_option("render_bin32", b)
            c.render_view(cams[0], W, H, to_host=False)
            c.profile(True)
            for cam in cams[:8]:
                c.render_view(cam, W, H, to_host=False)
            names = ("render_pre", "render_bucket", "radix_hist", "radix_rowscan", "radix_scatter", "render_bin_count", "scan", "render_bin_emit", "render_ranges", "render_blend")
            ms = {k: c.profile_get(k)[1] / 8 for k in names}
            c.profile(False)
            print(f"render_bin32={b}: kernel ms per view " + " ".join(f"{k}={v:.4f}" for k, v in ms.items()) + f"  sum {sum(ms.values()):.4f}", flush=True)
    if len(sys.argv) > 1 and sys.argv[1] == "compact":  # the phase cut once the level-1 sort holds only the splats a tile sees (render_compact)
        sweep = ({"render_compact": 0}, {}, {"render_phase_ratio": 3}, {"render_phase_ratio": 2}, {"render_phase_ratio": 5},
                 {"render_phases": 3, "render_phase_ratio": 3}, {"render_phases": 3, "render_phase_ratio": 2}) * 2 + ({"render_compact": 0}, {})
        if "ratio" in sys.argv[2:]:
            sweep = ({}, {"render_phase_ratio": 5}, {"render_phase_ratio": 6}, {"render_phase_ratio": 7}, {"render_phase_ratio": 8}, {"render_phase_ratio": 10},
                     {"render_phases": 3, "render_phase_ratio": 4}) * 2 + ({},)
    if len(sys.argv) > 1 and sys.argv[1] == "one":     # the defaults, three times (one pass per variant library: GSX_LIBRARY)
        sweep = ({},) * 3
    if len(sys.argv) > 1 and sys.argv[1] == "wide":    # the pair sorts in one 11-bit pass (ranges included) against two passes + ranges kernel
        sweep = ({"render_wide_sort": 0}, {"render_wide_sort": 2}, {"render_wide_sort": 1}) * 3
        for b in (0, 1):
            c.set_option("render_wide_sort", b)
            c.render_view(cams[0], W, H, to_host=False)
            c.profile(True)
            for cam in cams[:8]:
                c.render_view(cam, W, H, to_host=False)
            names = ("render_pre", "render_bucket", "radix_hist", "radix_rowscan", "radix_scatter", "render_bin_count", "scan", "render_bin_emit", "render_ranges", "render_blend")
            have = set(c.profile_names())
            ms = {k: (c.profile_get(k)[1] / 8 if k in have else 0.0) for k in names}
            c.profile(False)
            print(f"render_wide_sort={b}: kernel ms per view " + " ".join(f"{k}={v:.4f}" for k, v in ms.items()) + f"  sum {sum(ms.values()):.4f}", flush=True)
    if len(sys.argv) > 1 and sys.argv[1] == "cull":    # bounding-box binning against the exact ellipse test, interleaved
        sweep = ({}, {"exact_cull": 1}) * 3
    if len(sys.argv) > 1 and sys.argv[1] == "blend1":  # one pass over both kernels (tools/blend_chunks.sh runs it per variant library)
        sweep = ({"blend_pk2": 1}, {"blend_pk2": 2})
    for opts in sweep:
        base = {"render_phases": 2, "render_phase_ratio": 6, "exact_cull": 0, "tile_lpt": 0, "blend_pk2": 2, "render_bin32": 1, "render_compact": 1, "render_wide_sort": 1}
        # (render_wide_sort 1: the one-pass pair sort for a frame on its own only, 2: also with frames in flight)
        base.update(opts)
        for k, v in base.items():
            c.set_option(k, v)
        c.render_views(cams, W, H, to_host=False)
        t0 = time.perf_counter()
        for rep in range(5):
            c.render_views(cams, W, H, to_host=False)
        dt = (time.perf_counter() - t0) / (5 * len(cams))
        print(f"{opts}: {dt * 1e3:.3f} ms/view = {1 / dt:.0f} views/s   pairs sorted {c.render_num_pairs() // len(cams)}  consumed {c.render_num_pairs_consumed() // len(cams)}", flush=True)
