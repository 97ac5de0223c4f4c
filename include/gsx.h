/*
 * gsx.h — C ABI of libgsx.so: MI355X (gfx950) Gaussian-splat majority-vote labeler + rasterizer.
 *
 * Plain C types only (pointers, sizes, PODs); no C++ or torch types cross this boundary.
 * The reference (GloireLINVANI/3D_Gaussian_Splatting_Project) has no FFI: its hot path lives
 * behind Python functions.  Each entry point names the reference lines it replaces
 * (dls.py = deep_learning_segmentation.py, gs.js = Web_Viewer_Gaussians_Selection/gaussians_selection.js).
 *
 * Conventions
 *   - every function returns GSX_OK (0) or a negative gsx_status; gsx_last_error() has the text.
 *   - "not visible" / "no vote" are ordinary results, never errors (dls.py:73,82,306).
 *   - host pointers are read/written during the call only; the library never keeps them.
 *   - device pointers returned by *_device() accessors stay valid until the next gsx_vote_begin /
 *     gsx_upload_* on the same ctx, and belong to the ctx.
 *   - one ctx = one GPU = one HIP stream (gsx_render_views and the early vote add internal ones); a ctx is used from one host
 *     thread at a time; the library runs its own worker threads for host-side packing and the second render stream.
 *   - there is NO CPU fallback: without a gfx950 device gsx_create fails with GSX_E_HIP.
 */
#ifndef GSX_H
#define GSX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GSX_ABI_VERSION 2

typedef enum gsx_status {
    GSX_OK = 0,
    GSX_E_INVALID = -1,     /* bad argument (NULL, negative size, unknown enum)            */
    GSX_E_HIP = -2,         /* a HIP runtime call failed / no usable device               */
    GSX_E_STATE = -3,       /* call order violated (e.g. vote_view before vote_begin)     */
    GSX_E_RANGE = -4,       /* a seg-map label outside [-1, n_classes-1], too many views  */
    GSX_E_UNSUPPORTED = -5, /* configuration outside what the kernels are built for       */
    GSX_E_IO = -6           /* PLY file could not be read / written                       */
} gsx_status;

typedef struct gsx_ctx gsx_ctx;

/* One entry of cameras.json (dls.py:17-22, 54-63; gs.js:81-107).  All fp64, as JSON gives. */
typedef struct gsx_camera {
    double fx, fy;
    int32_t width, height;
    double R[9]; /* "rotation", row-major 3x3 */
    double p[3]; /* "position" */
} gsx_camera;

/* element type of a segmentation map handed to gsx_vote_view* (dls.py:124,142,158) */
typedef enum gsx_seg_dtype {
    GSX_SEG_I32 = 0, /* int32, values in [-1, n_classes-1]  (YOLO / Mask2Former maps)      */
    GSX_SEG_I64 = 1, /* int64, same range                    (SegFormer argmax)             */
    GSX_SEG_U8 = 2,  /* uint8 holding label+1 (0 = label -1): the compact on-device form   */
    GSX_SEG_U8_LABELS = 3 /* uint8 holding the label itself, 0 .. n_classes-1 (a class map stored
                        * as an 8-bit image; it cannot express -1)                          */
} gsx_seg_dtype;

/* ---------------------------------------------------------------------------------------------
 * context
 * ------------------------------------------------------------------------------------------- */
int gsx_abi_version(void);
/* number of HIP devices visible to this process (0 if none / no driver) */
int gsx_device_count(void);
/* device_id: HIP device ordinal of this process's GPU.  Fails (GSX_E_HIP) if the device is
 * missing or is not gfx950. */
int gsx_create(int device_id, gsx_ctx** out);
void gsx_destroy(gsx_ctx* ctx);
/* ctx may be NULL: returns the calling thread's last error raised without a ctx */
const char* gsx_last_error(const gsx_ctx* ctx);
/* the ctx's hipStream_t (as void*) on which every kernel of this ctx is launched */
void* gsx_stream(gsx_ctx* ctx);
int gsx_synchronize(gsx_ctx* ctx);
/* tuning knobs; none of them changes any result.
 *   "spatial_sort" (default 1)  Morton-order the Gaussians on the GPU at upload: a wave's 64 Gaussians
 *                               then project to neighbouring pixels (seg-map gathers hit few lines)
 *   "xcd_swizzle"  (default 32) which workgroups of the vote kernels share an XCD and its L2: 0 = hardware order,
 *                               1 = XCD x takes the x-th contiguous eighth of the Morton curve, C >= 2 = the curve
 *                               is cut into chunks of C workgroups dealt round-robin to the XCDs (compact pieces
 *                               of space per L2, and all XCDs finish together)
 *   "vote_unroll"  (default 8)  views whose seg-map gathers are in flight together: 2, 4, 8
 *   "flat_project" (default 1)  the projection as one straight-line block (all rows, both divisions, one predicate at
 *                               the end, one wave-uniform depth early-out) instead of the reference's three early
 *                               returns as divergent branches; same operations on the same operands
 *   "wave_cull"    (default 1)  skip a view for a whole wave when the bounding sphere of its 64 Gaussians lies
 *                               outside the view's frustum by a safety margin (see gsx_debug_cull_planes).  Skips
 *                               33 % of the (wave, view) pairs of the benchmark scene, bit-identical results.  On its
 *                               own it measured 3-5 % slower (the kernel waited on seg-map gathers, which invisible
 *                               pairs never issued); together with "seg_coarse" it is worth 12 %
 *   "lds_batch"    (default 0)  read the LDS counters of a whole chunk of views in one round trip and
 *                               resolve repeated bins in registers (measured 2.6 % slower: VALU-bound)
 *   "fast_div"     (default 0)  projection through ONE reciprocal with a certified margin; lanes within 2^-20 of a
 *                               pixel boundary (and any non-finite case) take the exact IEEE divisions.  Bit-identical
 *                               (tested on 7e7 pairs and at pixel boundaries) but measured 1 % SLOWER: 15 fewer fp64
 *                               instructions per visible pair, yet both axes are evaluated before the first early-out
 *   "filter_project" (default 1) the two IEEE divisions of the projection are preceded by an fp32 filter (one v_rcp_f32, two
 *                               v_fma_f32) with a proven error bound: a wave whose lanes are all farther than the bound from
 *                               every pixel boundary takes floor() of the filter's result, any other wave the exact
 *                               divisions for that view (about one wave and view in ten at 1080p).  Bit-identical by
 *                               construction (csrc/vote.hip: project_filtered; gsx_debug_filter_check)
 *   "host_compact" (default 1)  host maps (gsx_vote_view) cross PCIe as their coarse level plus the 16-byte blocks of the
 *                               mixed 4x4 cells only, and a kernel behind the DMA rebuilds the pool form (the link, not
 *                               the host pass, is what the hand-over of 200 1080p maps waits for); 0 = the pool form
 *                               itself crosses the link.  Same pool bytes, same labels
 *   "early_vote"   (default 1)  a run whose views all arrive on this context through gsx_vote_view (first_view 0, at most
 *                               255 announced, >= 32 of them, >= 2^18 Gaussians) votes its first views on a second stream
 *                               while the host is still handing over the rest; gsx_vote_finalize then walks only the views
 *                               behind them on top of the early counts (with more than 255 announced views: every batch of
 *                               <= 255 views but the last starts its count kernel when it is staged).  Same labels, bit for bit.  0 = off, 2 = whatever
 *                               the run's size (tests).  "early_vote_at": the stage starts when this many permille of the
 *                               announced views are staged; 0 (default): chosen from the run's own hand-over rate so
 *                               that the stage ends as the last map arrives (between 50 and 88 %)
 *   "early_replay" (default 1)  the form of the early vote: 1 = the early stage only RECORDS every vote (no histogram, no LDS:
 *                               0.73 ms of GPU time for 160 of 200 views at 3 M Gaussians), the last stage replays the record
 *                               behind its own walk in the one-piece kernel's order; 0 = the early stage keeps count and
 *                               first-view planes (1.85 ms, 0.9 GB more memory) and the last stage folds them in.  Same
 *                               labels, same span within the run-to-run spread (round 3 made the cheaper one the default)
 *   "labels_u8"    (default 1)  labels leave the device as one byte each (label + 1) and are widened on the host
 *                               (gsx_vote_finalize); 0 = int32 over the link
 *   "exchange_slabs" / "exchange_local"  plane layout of exchange protocol v2, see gsx_vote_slab_reduce
 *   "blend_pk2"    (default 2)  rasterizer: 0 = one pixel per thread, 1 = two (packed fp32), 2 = four (one wave per tile,
 *                               staging chunks of 64 records, the "tile opaque" vote every 16)
 *   "render_phases" (default 2) rasterizer: a frame is binned, sorted and blended front to back in this many DEPTH
 *                               PHASES (1..8); a phase skips the tiles the earlier ones left opaque (1 - alpha <
 *                               1e-5 on every pixel), so a dense scene sorts a few times the (tile, splat) pairs the
 *                               blend consumes instead of all of them (3 M splats @1080p: 23.3 M pairs -> 7.2 M with 2
 *                               phases, 2.2 M consumed; with "render_bin32" and "render_compact": 2.75 M (bin, splat) pairs for
 *                               1.2 M records evaluated).  What is skipped could not have moved a channel by 1e-5
 *   "render_phase_ratio" (default 6)  phase p ends after nvis / ratio^(K-1-p) splats of the depth order (2..64), nvis = the
 *                               splats the level-1 sort keeps ("render_compact")
 *   "render_compact" (default 1) rasterizer: the splats without a tile rectangle in the view (behind the camera, off the frame,
 *                               between the pixel centres, dropped by the JS sort's quirk) get a key that the FIRST pass of
 *                               the level-1 radix sort leaves out - a stable LSD pass is an order-preserving partition
 *                               anyway - so the second pass, the bin kernels and their scans see only the others; the
 *                               depth phases are cut on the device from their count.  0 = every splat is sorted (the
 *                               phases are then cut from n: the best ratio was 4)
 *   "render_frames" (default 4) gsx_render_views: frames in flight, each on a HIP stream of its own (1..6; 935 / 1252 /
 *                               1359 / 1396 / 1296 / 1349 views/s with 1..6 at 3 M splats @1080p SH 3: four streams for the
 *                               four hardware queues)
 *   "render_multi_pre" (default 1)  gsx_render_views: ONE pre pass (depth keys, vertex shader, colours, tile rectangles) per
 *                               group of frames in flight instead of one per frame: a splat's texel pair and SH coefficients
 *                               are read once for all views of the group (168 MB per view instead of 816 at 3 M splats, SH
 *                               3).  Bit-identical frames
 *   "render_bin32" (default 1)  rasterizer: the (list, splat) pairs are binned, sorted and ranged by 32x32-pixel BINS (2x2 tiles);
 *                               a pair's value carries the mask of the bin's tiles the splat's rectangle covers (and that
 *                               are not opaque yet), and a tile's blend wave takes the entries of its bin's list that name
 *                               it, in list order: 7.45 M -> 2.87 M pairs per view at 3 M splats @1080p, 1610-1650 ->
 *                               1910-1960 views/s, frames bit-identical.  Needs "blend_pk2" = 2, "exact_cull" = 0 and
 *                               fewer than 2^28 splats; otherwise (or with 0) the lists are per 16x16 tile
 *   "render_wide_sort" (default 1) rasterizer: with at most 2048 lists (32x32-pixel bins up to about 1440p) the pair sort is ONE radix
 *                               pass over the whole 11-bit key whose digit bases are the lists' ranges (no second pass, no ranges
 *                               kernel, no keys written): 1 = for a frame on its own (gsx_render_view: 6 % less kernel time), 2 =
 *                               also with several frames in flight (measured 1 % slower there: its stores are scattered), 0 = never.
 *                               The same lists, bit-identical frames
 *   "render_share_stream" (default 1)  gsx_render_views: the first extra frame runs on the context's second stream (the
 *                               early vote's) instead of one more stream - a context that has labelled before would
 *                               otherwise hold five streams for four hardware queues (1120 instead of 1340 views/s)
 *   "exact_cull"   (default 0)  rasterizer: bin a splat only into the tiles its |vPosition| <= 2 ellipse reaches
 *                               (minimum of the quadratic over the tile), not its whole bounding box.  The binning
 *                               is wave-cooperative (no lane walks a rectangle on its own), yet on the 3 M-splat
 *                               scene the test still costs more (+0.18 ms) than the 23 % fewer pairs save (-0.07)
 *   "tile_lpt"     (default 0)  rasterizer: blend the tiles with the longest splat lists first (measured:
 *                               blend -3 %, paid back by the extra ordering launches; per-tile lists only: ignored
 *                               while "render_bin32" is in effect)
 *   "seg_tiled"    (default 1)  keep the u8 seg maps as strips of 16 pixel columns (16x8 pixels per 128-B line;
 *                               applies to the views staged after the call)
 *   "batched_counts" (default 1) more than 255 views on one GPU: one fast u8-histogram launch per batch of <= 255 views
 *                               into its own count plane, then the sparse tie pass of exchange protocol v3 across
 *                               the batches (0: accumulate 16-bit count and first-view planes instead)
 *   "seg_coarse"   (default 1)  keep a second, 4x4-coarsened level of every strip-stored map (a cell holds the label
 *                               its 16 pixels share, or 255) and look a vote up there first; only lanes that hit a
 *                               mixed cell read the full-resolution map.  Same labels; a wave then touches ~3 cache
 *                               lines per view instead of ~16 (HBM traffic per launch 5.2 GB -> 0.5 GB).  Needs
 *                               n_classes <= 254, unit scale and "seg_tiled"; otherwise the one-level path runs */
int gsx_set_option(gsx_ctx* ctx, const char* name, int64_t value);

/* ---------------------------------------------------------------------------------------------
 * scene upload — replaces load_gaussians (dls.py:25-40), which keeps only x,y,z
 * ------------------------------------------------------------------------------------------- */
/* SoA host arrays of n floats each */
int gsx_upload_positions(gsx_ctx* ctx, int64_t n, const float* x, const float* y, const float* z);
/* AoS rows (e.g. the vertex rows of a 3DGS PLY): float at base + i*stride_bytes + off_{x,y,z} */
int gsx_upload_positions_strided(gsx_ctx* ctx, int64_t n, const void* base, int64_t stride_bytes,
                                 int64_t off_x, int64_t off_y, int64_t off_z);
int64_t gsx_num_gaussians(const gsx_ctx* ctx);

/* ---------------------------------------------------------------------------------------------
 * parity probe for project_gaussian (dls.py:43-82): one position through the device kernel.
 * *visible = 0 where the reference returns None; then *x = *y = -1.
 * ------------------------------------------------------------------------------------------- */
int gsx_project_one(gsx_ctx* ctx, const float pos[3], const gsx_camera* cam, int32_t* x, int32_t* y,
                    int32_t* visible);
/* batch form over the uploaded positions: x,y are host arrays of n int32 (-1 = not visible) */
int gsx_project_all(gsx_ctx* ctx, const gsx_camera* cam, int32_t* x, int32_t* y);

/* ---------------------------------------------------------------------------------------------
 * majority vote — replaces the loop body and arg-max of assign_labels (dls.py:252-308)
 *
 *   gsx_vote_begin        gaussian_votes = {}                                   dls.py:252
 *   gsx_vote_view*        one iteration of `for camera in cameras` whose PNG exists.  The caller
 *                         keeps the skip of missing images (dls.py:256-259): views are numbered
 *                         by the caller in processing order.                    dls.py:255-295
 *   gsx_vote_finalize     arg-max, first-inserted label wins ties, -1 if never visible
 *                                                                               dls.py:297-308
 * Views are staged in device memory (compact u8 maps) and consumed by ONE fused kernel launch at
 * flush/finalize time; they stay resident until the next gsx_vote_begin so that a run can be
 * repeated (gsx_vote_rewind) without re-uploading.
 * ------------------------------------------------------------------------------------------- */
/* n_classes: labels are -1 .. n_classes-1 (1 <= n_classes <= 255: the on-device maps are u8).
 * first_view / total_views: this ctx will receive the global view indices
 * [first_view, first_view + k); total_views is the number of views over ALL ranks (sizes the
 * counters: <= 65535).  Single GPU: first_view = 0, total_views = number of views (upper bound ok). */
int gsx_vote_begin(gsx_ctx* ctx, int32_t n_classes, int32_t first_view, int32_t total_views);
/* seg: HOST pointer, seg_h x seg_w row-major.  img_w,img_h: the PIL image size (dls.py:261-263).
 * The map is read during the call (worker threads narrow it to u8 bins and check the label range: GSX_E_RANGE fails THIS
 * call and stages nothing) and written into pinned memory in a compact transfer form - the 4x4-coarsened level plus one
 * 16-byte block per cell whose pixels differ (option "host_compact"); one asynchronous DMA per group of up to 16 maps moves
 * the records and a kernel queued behind it rebuilds the two-level on-device map.  Nothing is synchronised: 200 calls
 * cost the host pass over the maps (the link carries ~0.3 MB per 1080p map of an ordinary segmentation, 2.2 MB at worst).
 * Option "host_threads" sizes the worker pool; default 0 = gsx_default_host_threads(): min(16, usable CPUs), where the usable
 * CPUs are the affinity mask capped by the cgroup's CPU quota and divided by LOCAL_WORLD_SIZE (the ranks torch.distributed.run
 * started on this node: one process per GPU share the node's CPUs); env GSX_HOST_THREADS overrides it.
 * Option "host_pack" = 0 selects the alternative hand-over: the workers only copy the raw map into pinned memory, the raw
 * bytes cross PCIe (4x as many for int32) and the fused kernel of gsx_vote_view_device packs them; the range check is
 * then the deferred device-side one.  Measured slower on the GPU box; for hosts short of cores. */
int gsx_vote_view(gsx_ctx* ctx, const gsx_camera* cam, const void* seg, int32_t seg_dtype, int32_t seg_w,
                  int32_t seg_h, int32_t img_w, int32_t img_h);
/* same, seg is a DEVICE pointer (e.g. the segmentation model's output tensor on this GPU): one fused kernel on
 * the ctx stream (gsx_stream) reads it once and writes both map levels; the caller orders that stream after the
 * producer of the map and keeps the map alive until the stream has passed (gsx_synchronize, an event, ...).
 * Nothing is read back here: a label outside [-1, n_classes-1] is recorded on the device and fails the call that
 * hands out the labels (gsx_vote_finalize, gsx_vote_labels_from_*) with GSX_E_RANGE, naming the view. */
int gsx_vote_view_device(gsx_ctx* ctx, const gsx_camera* cam, const void* seg_dev, int32_t seg_dtype,
                         int32_t seg_w, int32_t seg_h, int32_t img_w, int32_t img_h);
/* n device maps of one geometry and dtype, views in array order: 16 maps per kernel launch (a 1080p map is ~2 us
 * of HBM time, less than a launch).  cams[n], segs_dev[n]. */
int gsx_vote_views_device(gsx_ctx* ctx, int32_t n, const gsx_camera* cams, const void* const* segs_dev, int32_t seg_dtype,
                          int32_t seg_w, int32_t seg_h, int32_t img_w, int32_t img_h);
int32_t gsx_vote_num_views(const gsx_ctx* ctx);
/* forget accumulated votes but keep the staged views: the next flush/finalize votes them again */
int gsx_vote_rewind(gsx_ctx* ctx);
/* single-GPU end: runs the fused kernel (+ arg-max) and leaves int32 labels on the device.
 * labels_out: host array of n int32, or NULL to skip the device-to-host copy.  A label is -1 .. 254, so the copy moves
 * ONE byte per Gaussian (label + 1) into pinned memory and the worker threads widen it into labels_out (option
 * "labels_u8" = 0: int32 over the link, the A/B of DESIGN.md section 3). */
int gsx_vote_finalize(gsx_ctx* ctx, int32_t* labels_out);
/* device pointer of the n int32 labels written by gsx_vote_finalize / gsx_vote_labels_from_keys */
void* gsx_vote_labels_device(gsx_ctx* ctx);

/* ---- multi-GPU exchange (views sharded over ranks; one process per GPU) ----------------------
 * rank-local:  gsx_vote_flush            fused kernel -> per-Gaussian vote histogram planes
 * exchange 1:  all-reduce SUM (int32 words) over gsx_vote_counts_device()   [the histogram]
 * rank-local:  gsx_vote_tiebreak_keys    per Gaussian: max-count bins -> (earliest view, bin) key
 * exchange 2:  all-reduce MAX (int32) over gsx_vote_keys_device()           [n words]
 * rank-local:  gsx_vote_labels_from_keys
 * The counts plane packs 8- or 16-bit counters into int32 words so that an int32 SUM is exact
 * (total votes per bin <= total_views, which fits the counter). */
int gsx_vote_flush(gsx_ctx* ctx);
void* gsx_vote_counts_device(gsx_ctx* ctx, int64_t* n_int32_words);
int gsx_vote_tiebreak_keys(gsx_ctx* ctx);
void* gsx_vote_keys_device(gsx_ctx* ctx, int64_t* n_int32_words);
int gsx_vote_labels_from_keys(gsx_ctx* ctx, int32_t* labels_out);
/* ---- multi-GPU exchange, protocol v2: all-to-all instead of all-reduce (SURVEY.md section 5's preferred shape)
 * Set the options "exchange_slabs" = world size and "exchange_local" = 1 BEFORE gsx_vote_begin: the planes
 * become u8 [slab][bins][sn] with per-rank counters (<= 255 views per rank) and LOCAL first-view codes.
 *   rank-local:  gsx_vote_flush
 *   exchange 1:  all_to_all (equal splits) over gsx_vote_counts_device() and over gsx_vote_first_device():
 *                rank j receives slab j of every rank, [src rank][bins][sn]           -- 2 x bins*n/world bytes/peer
 *   rank-local:  gsx_vote_slab_reduce(recv_counts, recv_first): sums the counters and resolves ties by
 *                (lowest rank, earliest local view) = globally earliest view, since ranks own contiguous
 *                rank-ordered view blocks -> sn int32 labels of this rank's slab at gsx_vote_keys_device()
 *   exchange 2:  all_gather of the slab labels -> slabs*sn int32 in Morton (upload) order
 *   rank-local:  gsx_vote_labels_from_sorted(all_labels_dev, labels_out): back to the caller's order */
void* gsx_vote_first_device(gsx_ctx* ctx, int64_t* n_int32_words);
/* ---- protocol v3: counts-only all-to-all + a sparse tie pass (same options as v2) -------------------------
 * Only the COUNT plane crosses the fabric (half of v2's bytes) and it comes out of the fast u8-histogram
 * kernel; first-view information is computed afterwards, for the tied Gaussians only.
 *   rank-local:  gsx_vote_flush_counts        u8 count plane [slab][bins][sn] at gsx_vote_counts_device()
 *   exchange 1:  all_to_all over the count plane
 *   rank-local:  gsx_vote_slab_totals(recv)   unique maximum -> label; otherwise label -2 and the set of
 *                                             max-count bins as a bit mask, u32 [8][sn] at gsx_vote_cand_device()
 *   exchange 2:  all_gather of the masks -> [slab][8][sn]
 *   rank-local:  gsx_vote_tie_codes(masks)    tied Gaussians only: walk this rank's views in forward order, stop
 *                                             at the first one voting a candidate -> u16 codes [slab][sn]
 *                                             ((255 - local view) << 8 | bin) at gsx_vote_codes_device()
 *   exchange 3:  all_to_all over the codes (2 bytes per Gaussian)
 *   rank-local:  gsx_vote_tie_resolve(recv)   lowest rank with a code wins = globally earliest view
 *   exchange 4:  all_gather of the slab labels (gsx_vote_keys_device), then gsx_vote_labels_from_sorted */
int gsx_vote_flush_counts(gsx_ctx* ctx);
int gsx_vote_slab_totals(gsx_ctx* ctx, const void* recv_counts_dev);
void* gsx_vote_cand_device(gsx_ctx* ctx, int64_t* n_int32_words);
int gsx_vote_tie_codes(gsx_ctx* ctx, const void* cand_all_dev);
void* gsx_vote_codes_device(gsx_ctx* ctx, int64_t* n_int32_words);
int gsx_vote_tie_resolve(gsx_ctx* ctx, const void* recv_codes_dev);
int64_t gsx_vote_slab_size(const gsx_ctx* ctx);
int gsx_vote_slab_reduce(gsx_ctx* ctx, const void* recv_counts_dev, const void* recv_first_dev);
int gsx_vote_labels_from_sorted(gsx_ctx* ctx, const void* sorted_labels_dev, int32_t* labels_out);
/* ---- protocol v4: views sharded for the hand-over, GAUSSIANS sharded for the vote (bench.py's default for N > 1) ----
 * The packed maps of 200 1080p views are 0.44 GB in total; the dense vote histogram of 3 M Gaussians is 0.45 GB PER
 * RANK.  So the maps cross the fabric, not the votes; no option is needed and the tie rule is the single-GPU one.
 *   rank-local:  gsx_vote_view* for this rank's contiguous, rank-ordered block of views
 *   exchange 0:  every rank learns all (view count, pool bytes) pairs and all view blobs  [tiny; see gsx_vote_export]
 *   rank-local:  gsx_vote_export(chunk): chunk = the largest rank's pool bytes; the rank's pool now spans >= chunk
 *   exchange 1:  all_gather of the pools, chunk bytes per rank, into one caller-owned device buffer
 *   rank-local:  gsx_vote_import: the ctx now holds ALL views, their maps read from the gathered buffer
 *                gsx_vote_slab_labels(rank, world): the fused single-GPU kernel over this rank's slab of the Gaussians
 *                (Morton order) -> slab_size int32 labels at gsx_vote_keys_device()
 *   exchange 2:  all_gather of the slab labels, then gsx_vote_labels_from_sorted
 * A view blob is GSX_VIEW_BLOB_BYTES opaque bytes (the library's view descriptor: camera, map geometry, offset inside
 * the exporting rank's pool); it is only meaningful to another instance of the same library build. */
#define GSX_VIEW_BLOB_BYTES 256
/* reserve_bytes: grow this rank's pool to at least that many bytes (the all-gather reads `chunk` bytes from it).
 * blobs_out (may be NULL): gsx_vote_num_views() * GSX_VIEW_BLOB_BYTES bytes on the host.  *pool_dev: the pool,
 * valid until the next gsx_vote_view* / gsx_vote_begin; *pool_bytes: bytes in use (a multiple of 256). */
int gsx_vote_export(gsx_ctx* ctx, int64_t reserve_bytes, void* blobs_out, void** pool_dev, int64_t* pool_bytes);
/* bytes of this rank's pool in use (what gsx_vote_export reports), without touching the stream: host maps that are packed
 * but whose grouped DMA has not been queued yet are counted; gsx_vote_export queues them */
int64_t gsx_vote_pool_bytes(const gsx_ctx* ctx);
/* statistics: bytes of HOST maps (gsx_vote_view) sent over PCIe since gsx_vote_begin - compact records (option
 * "host_compact") or maps in pool form */
int64_t gsx_vote_link_bytes(const gsx_ctx* ctx);
/* statistics: the run's first views that were voted on a second stream while the rest was still being handed over
 * (option "early_vote"; 0: none, the run is voted in one piece by gsx_vote_finalize) */
int64_t gsx_vote_early_views(const gsx_ctx* ctx);
/* part r contributed part_views[r] views (blobs in part order) whose maps start at byte part_offsets[r] of
 * pool_all_dev (pool_all_bytes long; caller-owned, must stay alive and unchanged until the labels have been fetched).
 * Replaces the views staged so far; global view order = part order.  Blobs are validated against the pool size. */
int gsx_vote_import(gsx_ctx* ctx, int32_t n_parts, const int32_t* part_views, const int64_t* part_offsets, const void* blobs,
                    const void* pool_all_dev, int64_t pool_all_bytes);
/* gsx_vote_import for a run whose maps all share ONE geometry (seg_w x seg_h, images img_w x img_h: a capture from one
 * camera model), without the blobs: every rank holds the whole camera list (load_cameras, dls.py:17-22) and derives the
 * descriptors itself, exactly as gsx_vote_view does - cams[k] is the camera of global view k (part order), view v of part r
 * lies at part_offsets[r] + v * (the pool stride gsx_vote_view gives a map of this geometry under this context's options:
 * all ranks must run the same build with the same options).  No exchange of view blobs is needed then. */
int gsx_vote_import_uniform(gsx_ctx* ctx, int32_t n_parts, const int32_t* part_views, const int64_t* part_offsets, const gsx_camera* cams,
                            int32_t seg_w, int32_t seg_h, int32_t img_w, int32_t img_h, const void* pool_all_dev, int64_t pool_all_bytes);
/* takes the last gsx_vote_import / gsx_vote_import_uniform back: the context holds this rank's own views in its own pool again
 * (a protocol that imports before it knows whether every rank's pool was what the schedule assumed can fall back) */
int gsx_vote_import_undo(gsx_ctx* ctx);
/* votes the Gaussians [slab * S, min(n, (slab+1) * S)) of the Morton order over all staged views, S = *slab_size =
 * ceil(n / slabs) rounded up to 256; labels (int32, Morton order) at gsx_vote_keys_device()[0 .. S). */
int gsx_vote_slab_labels(gsx_ctx* ctx, int32_t slab, int32_t slabs, int64_t* slab_size);
/* copies of the rank-local planes for tests: counts[bins][n] and first-view codes, widened to u16 */
int gsx_vote_debug_planes(gsx_ctx* ctx, uint16_t* counts_out, uint16_t* first_out);

/* ---------------------------------------------------------------------------------------------
 * rasterizer — the viewer's GPU/worker path (gs.js) as tile-based HIP kernels
 *
 *   gsx_upload_splats    processPlyBuffer gs.js:464-585 (importance order, u8 quantisation, 32-byte
 *                        .splat rows) + generateTexture gs.js:286-357 (4*Sigma as truncated fp16)
 *   gsx_render_view      getViewMatrix/calculateProjectionMatrix gs.js:66-107, runSort gs.js:417-462
 *                        (front-to-back 16-bit depth buckets, stable), vertex shader gs.js:696-750,
 *                        fragment shader + blend gs.js:782-799, 1033-1038
 * Inputs are the 3DGS PLY attributes as float arrays (AoS per attribute, row i = vertex i):
 *   xyz n x 3, scale n x 3 (log), rot n x 4 (w first), opacity n (logit), f_dc n x 3, labels n (or NULL).
 *   scale == NULL selects the viewer's fallback (scale 0.01, identity rotation, gs.js:559-563);
 *   opacity == NULL -> alpha 255 (gs.js:576).
 * rgba_out: height x width x 4 floats, row 0 = top row, premultiplied colour exactly as the fragment
 * shader emits it (float, not the canvas's 8-bit quantisation); NULL leaves the image on the device.
 * ------------------------------------------------------------------------------------------- */
int gsx_upload_splats(gsx_ctx* ctx, int64_t n, const float* xyz, const float* scale, const float* rot,
                      const float* opacity, const float* f_dc, const int32_t* labels);
/* Optional view-dependent colour: spherical harmonics of degree 0..3 (standard 3DGS real-SH basis,
 * dir = normalize(position - camera position), rgb = clamp(0.5 + SH, 0, 1) in float).  NOT part of the
 * reference (gs.js:565-569 reads only f_dc and quantises it to 8 bits): once this is called the float
 * SH colour replaces rgba8/255; geometry, alpha and depth fade stay the reference's.  f_rest: n x
 * 3*((d+1)^2-1) floats in 3DGS PLY order (f_rest_[c*K1 + k-1]); NULL for degree 0.  Call after
 * gsx_upload_splats; a new gsx_upload_splats switches it off again. */
int gsx_upload_sh(gsx_ctx* ctx, const float* f_rest, int32_t sh_degree);
int64_t gsx_num_splats(const gsx_ctx* ctx);
int gsx_render_view(gsx_ctx* ctx, const gsx_camera* cam, int32_t width, int32_t height, float* rgba_out);
/* n views of one size, F frames in flight (option "render_frames", default 4, 1..6; five and six measured slower than four): the context keeps F - 1 further HIP
 * streams with their own per-frame buffers that share the uploaded scene, and renders view k on stream k % F, each further
 * stream from a host thread of its own - the memory-bound front of one frame overlaps the VALU-bound blend tail of the
 * others (3 M splats / 1080p / SH 3: 935 views/s one at a time, 1252 / 1359 / 1396 with 2 / 3 / 4 in flight).  Same pixels
 * as gsx_render_view.  rgba_out: NULL, or n pointers (each NULL or height x width x 4 floats).  Afterwards
 * gsx_render_num_pairs* report the SUMS over the n views and gsx_render_image_device the last view of the context's own
 * stream (the last k with k % F == 0).  With profiling enabled the views are rendered one at a time. */
int gsx_render_views(gsx_ctx* ctx, int32_t n, const gsx_camera* cams, int32_t width, int32_t height, float* const* rgba_out);
void* gsx_render_image_device(gsx_ctx* ctx);
/* number of (tile, splat) pairs the last gsx_render_view sorted and blended */
int64_t gsx_render_num_pairs(const gsx_ctx* ctx);
/* ... and how many of them the blend kernel actually read (a tile stops once it is opaque) */
int64_t gsx_render_num_pairs_consumed(const gsx_ctx* ctx);
/* Selection hit test — performHitTesting, gs.js:361-395: the label of the splat whose projected centre is
 * nearest to (x, y) (canvas pixels, y measured from the BOTTOM as the viewer's click handler does,
 * gs.js:1377-1378) within 10 px, nearer NDC depth breaking exact ties; NO_SELECTION (-999999) if none.
 * *index_out = importance-order row of that splat or -1.  Uses the splats of gsx_upload_splats. */
int gsx_hit_test(gsx_ctx* ctx, const gsx_camera* cam, int32_t width, int32_t height, double x, double y,
                 int32_t* label_out, int64_t* index_out);
/* test hooks (any pointer may be NULL): the packed .splat rows (n x 32 bytes) and importance
 * permutation (gs.js:527), the texture words (n x 8 u32, gs.js:311-353) and the last view's 16-bit
 * depth buckets (65536 = dropped by the JS counting sort, gs.js:443-457) */
int gsx_render_debug(gsx_ctx* ctx, uint8_t* buffer_out, uint32_t* order_out, uint32_t* texdata_out,
                     uint32_t* bucket_out);

/* ---------------------------------------------------------------------------------------------
 * PLY files — replaces the plyfile calls of the reference: PlyData.read (dls.py:29, ply_handler.py:
 * 44-45) and PlyData([vertex], text=False).write (dls.py:331-332, ply_handler.py:35-37).
 * Only the vertex element (scalar properties) is kept, as save_labeled_ply does.  ascii and
 * binary_little_endian are read; rows are exposed in binary little-endian layout either way.
 * These entry points need no GPU and no ctx (errors: gsx_last_error(NULL)).
 * ------------------------------------------------------------------------------------------- */
typedef struct gsx_ply gsx_ply;
typedef enum gsx_ply_type {
    GSX_PLY_CHAR = 0, GSX_PLY_UCHAR = 1, GSX_PLY_SHORT = 2, GSX_PLY_USHORT = 3,
    GSX_PLY_INT = 4, GSX_PLY_UINT = 5, GSX_PLY_FLOAT = 6, GSX_PLY_DOUBLE = 7
} gsx_ply_type;
int gsx_ply_open(const char* path, gsx_ply** out);
void gsx_ply_close(gsx_ply* ply);
int64_t gsx_ply_num_vertices(const gsx_ply* ply);
int32_t gsx_ply_num_properties(const gsx_ply* ply);
/* i-th vertex property: name (owned by ply), gsx_ply_type, byte offset inside a row */
int gsx_ply_property(const gsx_ply* ply, int32_t i, const char** name, int32_t* type, int64_t* offset);
int64_t gsx_ply_row_stride(const gsx_ply* ply);
/* the vertex rows (num_vertices x row_stride bytes): a private copy-on-write mapping of the file or a
 * parsed buffer; writable, never written back to the source file; feeds gsx_upload_positions_strided */
void* gsx_ply_rows(gsx_ply* ply);
/* one property as floats (any scalar type is converted) / overwrite one property from floats */
int gsx_ply_read_f32(const gsx_ply* ply, const char* name, float* out);
int gsx_ply_set_f32(gsx_ply* ply, const char* name, const float* in);
/* writes every vertex property, plus a trailing `property int label` when labels != NULL
 * (save_labeled_ply, dls.py:318-332).  text != 0 writes ascii (k_means.py:193). */
int gsx_ply_write(const gsx_ply* ply, const char* path, const int32_t* labels, int32_t text);

/* ---------------------------------------------------------------------------------------------
 * test hook: the library's stable LSD radix sort of (u32 key, u32 value) pairs by key bits
 * [0, bits) — the primitive behind the Morton ordering and the rasterizer's (tile | depth16) order,
 * which restates the stable counting sort of gs.js:443-457.  Host arrays, sorted in place.
 * ------------------------------------------------------------------------------------------- */
int gsx_debug_sort_pairs(gsx_ctx* ctx, uint32_t* keys, uint32_t* values, int64_t n, int32_t bits);
/* the rasterizer's level-1 sort (csrc/sort.hip: radix_sort_pairs_drop): pairs whose key is 0xffffffff are left out by the first
 * pass; the others come back sorted (stable) in the first *kept_out slots, the slots behind them hold unspecified values */
int gsx_debug_sort_pairs_drop(gsx_ctx* ctx, uint32_t* keys, uint32_t* values, int64_t n, int32_t bits, int64_t* kept_out);
/* test hook, host only (no context, no GPU): the packed form of one map exactly as gsx_vote_view stages it in
 * pinned memory - u8 bins in strips of 16 pixel columns (tiled != 0; row-major otherwise) followed, at *coarse_off
 * (-1: none), by the 4x4-coarsened level.  out == NULL only reports *bytes.  *bad = 1 if a label was out of range. */
int gsx_debug_host_pack(const void* seg, int32_t seg_dtype, int32_t w, int32_t h, int32_t n_classes, int32_t tiled,
                        int32_t coarse, int32_t threads, uint8_t* out, int64_t out_cap, int64_t* bytes, int64_t* coarse_off,
                        int32_t* bad);
/* test hook, host only: the COMPACT transfer form in which gsx_vote_view sends a two-level map over PCIe (option
 * "host_compact", default 1): uint32 first_block[2 * bands of 8 pixel rows] | the coarse level as in the pool | one 16-byte
 * block (4 rows x 4 pixels) per mixed 4x4 cell.  A band holds two rows of cells; entry 2 * band + row says where that cell
 * row's blocks start, inside a cell row they follow the cell columns; the cell rows follow each other in the order their
 * workers reserved room.  out == NULL: *bytes = worst-case size; otherwise bytes in use.
 * Option "host_prefetch" (default 8192): bytes the narrowing loops prefetch ahead of themselves (NTA hint; a negative value
 * selects T0); "host_prefetch_burst" (default 1): a band that starts a new stream asks for that distance at once. */
int gsx_debug_host_pack_compact(const void* seg, int32_t seg_dtype, int32_t w, int32_t h, int32_t n_classes, int32_t threads,
                                uint8_t* out, int64_t out_cap, int64_t* bytes, int64_t* table_bytes, int64_t* stream_off,
                                int32_t* bad);
/* test hook, host only: `runs` fork-joins of pseudo-random size (1..max_parts parts) on ONE worker pool of `threads` threads;
 * returns how many parts did not run exactly once (0 = the pool is sound), -1 if the pool could not be created */
int64_t gsx_debug_workers_stress(int32_t threads, int32_t runs, int32_t max_parts);
/* test hook, host only: the D2H epilogue's widening pass - labels_out[i] = bins[i] - 1 (the labels cross PCIe as one byte
 * each, bin = label + 1), on `threads` worker threads */
int gsx_debug_widen_labels(int32_t threads, const uint8_t* bins, int64_t n, int32_t* labels_out);
/* statistics: (wave of 64 Gaussians, view) pairs the vote kernels skipped through the wave culling since the context
 * was created or since the last call with reset != 0 */
int gsx_vote_culled(gsx_ctx* ctx, int64_t* wave_views, int32_t reset);
/* test hook: what the fp32 filter in front of the projection's divisions (option "filter_project") assumes about the
 * hardware, measured on the device.  out[17]: [0] = the largest relative error of v_rcp_f32 over EVERY float in
 * [2^-41, 2^41], in units of 2^-24 (the filter's proof needs <= 3); [1..8] = v_fract_f32 and [9..16] =
 * v_cvt_flr_i32_f32 of +inf, -inf, -1e-10, NaN, 1920.5, -0.25, 3e38, -3e38. */
int gsx_debug_filter_check(gsx_ctx* ctx, double* out);
/* test hook, host only (no context, no GPU): the five world-space culling planes the vote kernels use to skip whole
 * waves for a view (option "wave_cull"), as out[5][5] = unit normal A, offset B, margin slope M: a sphere (c, r) with
 * A.c + B > r + M (|c|_1 + r) holds no Gaussian that project_gaussian (deep_learning_segmentation.py:43-82) would
 * accept (the margin is a million times the rounding error of the fp64 projection of any point of the sphere). */
int gsx_debug_cull_planes(const gsx_camera* cam, double* out);

/* ---------------------------------------------------------------------------------------------
 * k-means labeler: 3D_clustering/k_means.py:107-151 k_means_with_color (the reference's other producer of the
 * `label` field).  points / colors: n x 3 float32 each, row-major, on the host (x y z and
 * f_dc_0..2, k_means.py:17-28).  init_index[k]: the rows used as initial centroids - the reference draws them
 * with an unseeded np.random.choice (k_means.py:111), so the caller supplies them.  Runs at most max_iter
 * rounds of assign (nearest centroid by float64 squared distance in scipy's summation order, :116-122) and
 * update (float32 mean with numpy's in-order accumulation, :125-128), stops early when the centroids moved by
 * less than tol (:132-136), and labels every row by the centroids in hand (:142-147).
 * labels_out int32[n] (required); centroids_out float32[k*6], iterations_out, converged_out may be NULL.
 * 1 <= k <= min(n, 2048).
 * ------------------------------------------------------------------------------------------- */
int gsx_kmeans(gsx_ctx* ctx, int64_t n, const float* points, const float* colors, int32_t k, const int64_t* init_index,
               int32_t max_iter, double tol, int32_t* labels_out, float* centroids_out, int32_t* iterations_out,
               int32_t* converged_out);

/* ---------------------------------------------------------------------------------------------
 * profiling hooks (HIP events on the ctx stream around each kernel launch)
 * ------------------------------------------------------------------------------------------- */
int gsx_profile_enable(gsx_ctx* ctx, int on);
int gsx_profile_reset(gsx_ctx* ctx);
/* the index-th kernel name seen since profiling was first enabled, or NULL past the end */
const char* gsx_profile_name(gsx_ctx* ctx, int32_t index);
/* worker threads (including the caller) gsx_vote_view packs host maps with; starts the pool if need be */
int gsx_host_threads(gsx_ctx* ctx);
/* the pool size a context would choose on its own in this process right now (no context, no GPU needed) */
int gsx_default_host_threads(void);
/* name: "vote_fused_labels", "seg_pack", ...; returns launches and total milliseconds since reset */
int gsx_profile_get(gsx_ctx* ctx, const char* name, int64_t* launches, double* total_ms);

#ifdef __cplusplus
}
#endif
#endif /* GSX_H */
