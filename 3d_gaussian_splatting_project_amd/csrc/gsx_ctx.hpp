// Internal (non-ABI) state of libgsx.so.  Everything here is private to the library.
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <map>
#include <cstddef>
#include <string>
#include <vector>

#include "../../include/gsx.h"
#include "host_pack.hpp"

namespace gsx {

// ---- error plumbing ----------------------------------------------------------------------------
void set_global_error(const char* fmt, ...);
struct Ctx;
int fail(Ctx* c, int code, const char* fmt, ...);

#define GSX_HIP(ctx, call)                                                                         \
    do {                                                                                           \
        hipError_t e__ = (call);                                                                   \
        if (e__ != hipSuccess)                                                                     \
            return gsx::fail((ctx), GSX_E_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e__), \
                             __FILE__, __LINE__);                                                  \
    } while (0)

// ---- device buffer: grows, never shrinks ---------------------------------------------------------
struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    bool owned = true;  // false: an alias of another context's buffer (gsx_render_views' twin): never freed or regrown here
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    DevBuf(DevBuf&& o) noexcept : p(o.p), cap(o.cap), owned(o.owned) { o.p = nullptr; o.cap = 0; o.owned = true; }
    DevBuf& operator=(DevBuf&& o) noexcept {
        if (this != &o) {
            release();
            p = o.p;
            cap = o.cap;
            owned = o.owned;
            o.p = nullptr;
            o.cap = 0;
            o.owned = true;
        }
        return *this;
    }
    void alias(const DevBuf& o) {  // view of o's memory; o must outlive this
        release();
        p = o.p;
        cap = o.cap;
        owned = false;
    }
    ~DevBuf() { release(); }
    hipError_t ensure(size_t bytes) {  // contents are NOT preserved on growth
        if (bytes <= cap) return hipSuccess;
        if (!owned) return hipErrorInvalidValue;  // an alias cannot grow
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        hipError_t e = hipMalloc(&p, bytes);
        if (e == hipSuccess) cap = bytes;
        return e;
    }
    void release() {
        if (p && owned) (void)hipFree(p);
        p = nullptr;
        cap = 0;
        owned = true;
    }
    template <class T>
    T* as() const { return reinterpret_cast<T*>(p); }
};

// ---- per-view descriptor read by the kernels through scalar (wave-uniform) loads ------------------
struct alignas(64) ViewDesc {
    // ---- core, 128 B = two s_load_dwordx16: everything that differs from view to view in an ordinary capture ----
    double R[9];        // cameras.json rotation, row-major                         (dls.py:60)
    double t[3];        // (-R) @ p in the dgemv association of the oracle           (dls.py:66)
    double fx, fy;      //                                                           (dls.py:54-55)
    long long seg_off;  // host copy: byte offset of this view's u8 map in the seg pool; device copy
                        // (sync_views): the map's absolute address
    int seg_row_bytes;  // bytes per strip of 16 pixel columns (0: row-major map)
    int unit_scale;     // both scales are exactly 1.0 and the camera frame fits the map: skip scale + clamp
    // ---- frame, 48 B: identical for all views when the cameras share one resolution (then kernel constants) ----
    double half_w, half_h;  // width/2, height/2                                    (dls.py:76-77)
    double width, height;   // bounds of the visibility test                        (dls.py:80)
    int seg_w, seg_h;
    int cam_w, cam_h;   // camera width / height as integers (visibility test of the certified path)
    int coarse_row_bytes;   // bytes per strip of 16 coarse cells (0: this view has no coarse level)
    unsigned coarse_delta;  // byte offset of the coarse level from the map's own start
    float hw32, hh32;       // half_w, half_h as floats (exact: the frame is <= 65535 pixels a side): the fp32 filter of the projection
    // ---- cold part: only read on the scale + clamp path (dls.py:270-286) ----
    double wscale, hscale;  // seg_w/img_w, seg_h/img_h                             (dls.py:270-271)
    float xf[12];           // experiments (GSX_ABLATE & 64, timing only): fx R0, fy R1, R2 rows and fx t0, fy t1, t2 as floats
};
static_assert(sizeof(ViewDesc) == 256 && offsetof(ViewDesc, wscale) == 192, "ViewDesc layout");

// one slot of the pinned staging ring of gsx_vote_view (host maps): packed by the host workers, DMA'd to the pool
struct PinSlot {
    void* p = nullptr;
    size_t cap = 0;
    hipEvent_t ev = nullptr;  // recorded after the slot's H2D copy
    bool busy = false;
};
static constexpr int kPinSlots = 4;
static constexpr int kDmaBatch = 4;    // packed host maps per H2D copy (one hipMemcpyAsync + event record costs ~6 us of host time)
static constexpr int kCompactBatch = 16;  // compact records per H2D copy and expansion launch (as many as fit into a group's part of the ring)
static constexpr int kDmaGroups = 3;   // ring = kDmaGroups groups of kDmaBatch slots: one filling, one or two in flight
static constexpr int kLabelChunks = 4;
static constexpr int kConsumedSlots = 64;  // rasterizer statistics: the blend kernels' "pairs staged" counter, one slot per 128-B line
static constexpr int kNoBadView = 0x7f7f7f7f;  // errflag value meaning "every device-side map was in range"

struct ProfEvent {
    int name_id;
    hipEvent_t start, stop;
};

struct Ctx {
    int device = -1;
    hipStream_t stream = nullptr;
    std::string err;

    // scene
    int64_t n = 0;
    int64_t n_pad = 0;  // n rounded up to 256: row pitch of the vote planes
    DevBuf x, y, z;     // SoA f32 positions (Morton order when `sorted`)
    DevBuf perm;        // u32[n]: original index of slot i (valid when `sorted`)
    bool sorted = false;
    DevBuf sort_hist;   // radix-sort scratch

    // options (gsx_set_option)
    int opt_spatial_sort = 1;  // Morton-order the Gaussians at upload (results do not depend on it)
    int opt_xcd_swizzle = 32;  // 0: off, 1: contiguous eighths of the Morton curve per XCD, C: chunks of C workgroups round-robin
    int opt_vote_unroll = 8;   // views whose seg gathers are in flight together: 1, 2, 4 or 8
    int opt_slabs = 1;         // see Ctx::slabs (takes effect at the next vote_begin)
    int opt_local_codes = 0;   // see Ctx::local_codes
    int opt_lds_batch = 0;     // read a chunk's LDS counters in one round trip (repeats resolved in registers)
    int opt_blend_pk2 = 2;     // rasterizer: 0 = one pixel per thread, 1 = two (packed fp32), 2 = four, one wave per tile (default since round 3)
    int opt_exact_cull = 0;    // rasterizer: keep only the tiles the splat's ellipse really reaches (pairs -23 %; the test costs more than the sort saves)
    int opt_tile_lpt = 0;      // rasterizer: launch the tiles with the longest lists first (blend -3 %, but net 0)
    int opt_seg_tiled = 1;     // store seg maps as 16x8-pixel tiles of 128 B
    int opt_fast_div = 0;      // certified single-reciprocal projection with exact fallback (bit-identical, not faster)
    int opt_batched_counts = 1; // > 255 views on one GPU: per-batch count planes + sparse tie pass instead of 16-bit planes
    int opt_seg_coarse = 1;    // keep a 4x4-coarsened level of every tiled map (views staged after the call)
    int opt_flat_project = 1;  // branchless projection block (exact divisions, one predicate at the end)

    // vote
    bool vote_begun = false;
    int n_classes = 0, bins = 0;
    int first_view = 0, total_views = 0;
    bool wide = false;  // 16-bit plane counters (total_views > 255)
    bool local_codes = false;  // planes hold per-rank u8 counters and LOCAL first-view codes (all-to-all exchange)
    int slabs = 1;             // planes are [slab][bins][sn]; one slab per rank of the all-to-all exchange
    int64_t sn = 0;            // Gaussians per slab (multiple of 256); n_pad = slabs * sn
    std::vector<ViewDesc> views;
    DevBuf d_cull;               // double[25][cull_pitch]: five world-space culling planes per staged view
    int cull_pitch = 0;
    DevBuf d_cull_tally;         // u64: (wave, view) pairs skipped by the culling since the last reset
    int opt_wave_cull = 1;       // skip (wave, view) pairs whose 64 Gaussians provably all miss the frame
    bool views_dirty = true;  // host views newer than d_views
    bool views_coarse = false;  // ... and a coarse level (set by sync_views)
    bool views_simple = false;  // every staged view: unit scale + tiled map (set by sync_views)
    DevBuf d_views;
    DevBuf segpool;
    size_t seg_used = 0;
    DevBuf errflag;  // int: smallest view index whose device-side map held a label out of range (kNoBadView: none)
    // host hand-over (vote.hip): worker pool, pinned ring for the maps, pinned landing zone for the labels
    Workers* workers = nullptr;
    int opt_host_threads = 0;  // 0: default_host_threads()
    int opt_host_compact = 1;  // host maps cross PCIe in the compact form (coarse level + the mixed cells' blocks), expanded on the GPU
#ifdef GSX_EXPERIMENTS  // `make experiments` only (libgsx_experiments.so, used by tools/): timing-only switches, results invalid
    int opt_ablate = 0;        // 1 = host maps are packed but not copied, 2 = copied but not packed; << 4: last-stage ablations (vote.hip)
#else
    static constexpr int opt_ablate = 0;  // the product library has no such option: every branch on it folds away
#endif
    int opt_labels_u8 = 1;     // 1: the labels cross PCIe as one byte each (bin = label + 1) and are widened by the workers; 0: as int32
    int opt_host_pack = 1;     // 1: host maps are narrowed to u8 by the workers (2.2 MB/map over PCIe); 0: raw copy + GPU pack kernel (8.3 MB/map)
    DevBuf dstage[kPinSlots];  // host_pack = 0: device-side landing zone of the raw map of each ring slot
    PinSlot ring[kPinSlots];   // host_pack = 0: raw maps
    int ring_next = 0;
    // host_pack = 1: ONE pinned buffer of kDmaGroups * kDmaBatch slots, a slot = the pool stride of the map geometry, so that
    // consecutive packed maps are consecutive both here and in the pool and a whole group goes up in one DMA
    void* hring = nullptr;
    size_t hring_bytes = 0, hring_slot = 0;
    hipEvent_t hring_ev[kDmaGroups] = {nullptr, nullptr, nullptr};
    bool hring_busy[kDmaGroups] = {false, false, false};
    int hring_next = 0;
    int pend_first = 0, pend_count = 0;  // packed maps of the filling group whose DMA has not been queued yet
    size_t pend_dst = 0;
    // compact transfer form: the records of a group lie back to back in its part of the ring
    bool hring_compact = false;
    size_t grp_used = 0;                 // bytes of the filling group in use
    size_t pend_lo = 0;                  // where the pending records start in the group
    size_t pend_rec[kCompactBatch] = {}; // their offsets in the group
    size_t pend_map[kCompactBatch] = {}; // their maps' offsets in the pool
    int cgrp = 0, grp_recs = 0;          // the filling group and the records it holds
    int pend_group = 0;                  // group of the pending records
    MapLayout pend_L;                    // their geometry (one expansion launch = one geometry)
    DevBuf cstage;                       // device staging of one group of records
    void* h_scratch = nullptr;           // host: the narrowed strips of the map being packed (ordinary memory)
    size_t h_scratch_cap = 0;
    long long compact_bytes = 0;         // statistics: bytes of host maps sent over PCIe since vote_begin (compact records or pool form)
    void* h_labels = nullptr;  // pinned: n u8 bins (label + 1) + one int (the error flag) land here before the caller's array
    size_t h_labels_cap = 0;
    hipEvent_t h_ev[kLabelChunks] = {nullptr, nullptr, nullptr, nullptr};
    void* h_views = nullptr;   // pinned: view descriptors + culling planes on their way to d_views / d_cull
    size_t h_views_cap = 0;
    hipEvent_t h_views_ev = nullptr;
    std::vector<ViewDesc> own_views;  // gsx_vote_import: this rank's own views, for gsx_vote_import_undo
    int own_first_view = 0;
    const void* pool_base = nullptr;  // gsx_vote_import: the maps live in a caller-owned gathered pool, not in segpool
    int n_flushed = 0;         // views [0, n_flushed) are already in the planes
    bool planes_valid = false;  // planes hold votes (zeroed at begin/rewind)
    bool planes_zero = false;   // planes are known to be all-zero
    bool planes_stale = false;  // planes hold a rewound run's votes; the next fresh flush overwrites them
    DevBuf cnt, fv;             // [bins][n_pad] counters / first-view codes (u8 or u16)
    DevBuf keys, labels;        // [n_pad] int32
    DevBuf labels8;             // [n] u8: bin = label + 1, what labels_to_host sends over PCIe
    DevBuf bcnt, bcodes;        // > 255 views on one GPU: per-batch u8 count planes [S][bins][n_pad], tie codes u16 [S][n_pad]
    DevBuf cand, codes;         // exchange v3: candidate masks u32[8][sn] of this slab; tie codes u16[n_pad]
    bool labels_valid = false;
    // early vote (vote.hip: early_vote_stage): the views [0, early_done) are voted on a second stream while the host is
    // still handing over the rest of the run; vote_finalize then only walks the views behind them
    int opt_filter_project = 1;  // fp32 filter in front of the two fp64 divisions of the projection (vote.hip: project_filtered); exact by construction
    int opt_early_vote = 1;      // 0: off, 1: for runs worth it (one rank holds all <= 255 views, a large scene), 2: whenever possible (tests)
    int opt_early_at = 0;        // the stage starts when this many permille of the announced views are staged; 0: chosen from the run's own hand-over rate
    std::chrono::steady_clock::time_point early_t0;  // first gsx_vote_view of the run
    int opt_early_replay = 1;    // 1 (default since round 3): the early stage only records the votes, the last stage replays them (vote_record_kernel / vote_fused_replay_kernel); 0: count + first-view planes and the fold
    bool early_replayed = false; // this run's early stage was a record-only one
    int early_state = 0;         // 0: not started in this run, 1: started, -1: not available any more (rewind, pool moved)
    int early_done = 0;          // views [0, early_done) are in ecnt / efv (or, > 255 announced views, in the first early_batches planes of bcnt)
    int early_batches = 0;       // > 255 announced views: batches of labels_batched() whose count kernels already ran on stream2
    hipStream_t stream2 = nullptr;
    bool stream_borrowed = false; // (a twin of gsx_render_views) `stream` is the parent context's second stream: not this context's to destroy
    int opt_render_share_stream = 1;  // gsx_render_views: the first extra frame runs on the context's second stream instead of one more stream
    bool early_inflight = false; // early_done_ev was recorded on stream2 and c->stream has not been ordered behind it yet (early_join)
    hipEvent_t early_maps_ev = nullptr, early_done_ev = nullptr, early_up_ev = nullptr;
    DevBuf ecnt, efv;            // u8 [wave][bin][64]: counts / first-view codes of the early views
    DevBuf erec;                 // u8 [wave][view][64]: bin + 1 voted by each Gaussian in each early view
    DevBuf e_views, e_cull;      // descriptors and culling planes (pitch kEarlyPitch) of the run's views, filled stage by stage
    void* h_early = nullptr;     // pinned staging of both
    size_t h_early_cap = 0;

    // rasterizer (render.hip / blend.hip)
    int64_t rn = 0;                      // splats uploaded
    DevBuf r_order, r_buffer, r_tex;     // importance permutation, .splat rows, texel pairs
    DevBuf r_sh;                         // per-view SH colour (n x 3 f32)
    DevBuf r_fdc, r_shc;                 // f_dc (source order); SH coefficients coef[k][i][3] (importance order)
    bool r_sh_valid = false, r_sh_on = false;
    int r_sh_deg = 0;
    DevBuf r_image;                      // float4[H][W] of the last view
    int r_W = 0, r_H = 0;
    DevBuf r_ranges, r_small, r_scan;
    DevBuf r_depth, r_bucket, r_rect, r_count, r_offset, r_rec;
    DevBuf r_pre;                        // int[4]: min depth, max depth, splat 0's tile rectangle (written by the pre pass)
    // gsx_render_views with several frames in flight: ONE pre pass per group of frames (pre_multi_kernel) writes the per-view
    // records of all of them; a frame's records live in one of three rotating sets so that the pass for group g + 1 can run
    // while the frames of group g (and stragglers of g - 1) still read theirs
    struct PreSet {
        DevBuf depth, rect, rec, pre;
    };
    static constexpr int kPreSets = 3;
    PreSet r_sets[kPreSets];
    DevBuf r_pre_args;                   // pre_multi_kernel's per-view arguments, one slot per record set
    std::vector<unsigned long long> r_pre_args_host;  // ... and their host copies (8-byte aligned; alive while the H2D copies run)
    hipEvent_t r_pre_ev[kPreSets] = {nullptr, nullptr, nullptr};  // recorded behind the pre pass that filled set s
    int r_pre_ext = -1;                  // >= 0: this frame's pre pass has been run for it into r_sets[r_pre_ext] (render_view skips its own)
    int opt_render_multi_pre = 1;        // gsx_render_views: one pre pass per group of frames in flight (0: every frame its own)
    DevBuf r_keys0, r_keys1, r_vals0, r_vals1;
    DevBuf r_tile_order;                 // blend launch order (longest list first)
    int r_sorted_in = 0;
    DevBuf r_sat;                        // u8 per tile: every pixel is opaque (1 - alpha < 1e-5): later depth phases skip it
    int opt_render_phases = 2;           // depth phases per frame (1 = bin and sort every pair at once)
    int opt_render_phase_ratio = 6;      // phase p ends at nvis / ratio^(K-1-p) splats (front to back; nvis = the splats with a rectangle in the view; 4 until the level-1 sort left the others out)
    int opt_render_bin32 = 1;            // bin, sort and range the splats by 32x32-pixel BINS (2x2 tiles); a pair carries the mask of the bin's tiles
                                         // the splat's rectangle covers (one-wave blend kernel, no exact_cull, < 2^28 splats; else 16x16)
    int opt_render_compact = 1;          // the level-1 sort leaves out the splats without a rectangle in the view (its first pass is the partition)
    int opt_render_wide_sort = 1;        // the pair sorts in ONE pass when the list index fits 11 bits (sort.hip: radix_sort_values_wide; the ranges come with it):
                                         // 0 never, 1 for a frame on its own (gsx_render_view), 2 also with several frames in flight
    bool r_in_flight = false;            // this context renders one of several frames in flight (set by render_views for the call)
    int r_bin32 = 0;                     // the frame being rendered uses bins (set by render_view, read by launch_blend)
    unsigned long long r_P = 0;          // (tile, splat) pairs of the last view (all phases)
    static constexpr int kMaxFrames = 6;
    Ctx* twins[kMaxFrames - 1] = {nullptr, nullptr, nullptr, nullptr, nullptr};  // gsx_render_views: further streams + per-frame buffers, aliasing this context's scene
    int opt_render_frames = 4;           // frames in flight in gsx_render_views (1 .. kMaxFrames): 935 / 1252 / 1359 / 1396 views/s with 1 / 2 / 3 / 4
    size_t r_pair_cap = 0;               // capacity (pairs) of r_keys*/r_vals*: grown when a phase overflows it, the frame is redone
    unsigned long long r_consumed = 0;   // pairs the blend kernel actually staged (early-out leaves the rest unread)
    DevBuf r_d0, r_d1, r_d2, r_d3;       // level-1 sort ping-pong (bucket, splat)
    DevBuf r_rects;                      // the tile rectangles of a phase's splats in the phase's order (bin COUNT -> EMIT)

    // profiling
    bool prof_on = false;
    std::vector<std::string> prof_names;
    std::vector<ProfEvent> prof_events;
    std::vector<hipEvent_t> event_pool;
    std::map<std::string, std::pair<int64_t, double>> prof_acc;
};

// RAII: brackets one kernel launch with events when profiling is on
struct ProfScope {
    Ctx* c;
    ProfEvent ev{};
    bool active = false;
    hipStream_t st = nullptr;
    ProfScope(Ctx* ctx, const char* name, hipStream_t on = nullptr);  // on: the stream the kernel is launched on (default: ctx->stream)
    ~ProfScope();
};

// vote.hip
int vote_begin(Ctx* c, int n_classes, int first_view, int total_views);
int vote_view(Ctx* c, const gsx_camera* cam, const void* seg, int seg_dtype, int seg_w, int seg_h, int img_w, int img_h);
int vote_views_device(Ctx* c, int n, const gsx_camera* cams, const void* const* segs, int seg_dtype, int seg_w, int seg_h,
                      int img_w, int img_h);
int vote_export(Ctx* c, int64_t reserve_bytes, void* blobs_out, void** pool_dev, int64_t* pool_bytes);
int vote_import(Ctx* c, int n_parts, const int32_t* part_views, const int64_t* part_offsets, const void* blobs,
                const void* pool_all_dev, int64_t pool_all_bytes);
int vote_import_uniform(Ctx* c, int n_parts, const int32_t* part_views, const int64_t* part_offsets, const gsx_camera* cams,
                        int seg_w, int seg_h, int img_w, int img_h, const void* pool_all_dev, int64_t pool_all_bytes);
int vote_import_undo(Ctx* c);
int vote_slab_labels(Ctx* c, int slab, int slabs, int64_t* slab_size);
int host_threads(Ctx* c);
int vote_flush_pending(Ctx* c);  // queue the DMA of packed host maps that are still waiting for their group to fill
void vote_release_host(Ctx* c);  // pinned buffers, events, worker pool (gsx_destroy)
int debug_host_pack(const void* seg, int seg_dtype, int w, int h, int n_classes, int tiled, int coarse, int threads,
                    uint8_t* out, int64_t out_cap, int64_t* bytes, int64_t* coarse_off, int32_t* bad);
int debug_host_pack_compact(const void* seg, int seg_dtype, int w, int h, int n_classes, int threads, uint8_t* out, int64_t out_cap,
                            int64_t* bytes, int64_t* table_bytes, int64_t* stream_off, int32_t* bad);
int vote_rewind(Ctx* c);
int vote_finalize(Ctx* c, int32_t* labels_out);
int vote_flush(Ctx* c);
int vote_tiebreak_keys(Ctx* c);
int vote_labels_from_keys(Ctx* c, int32_t* labels_out);
int vote_debug_planes(Ctx* c, uint16_t* counts_out, uint16_t* first_out);
int vote_slab_reduce(Ctx* c, const void* recv_cnt, const void* recv_fv);
int vote_flush_counts(Ctx* c);
int vote_slab_totals(Ctx* c, const void* recv_cnt);
int vote_tie_codes(Ctx* c, const void* cand_all);
int vote_tie_resolve(Ctx* c, const void* recv_codes);
int vote_labels_from_sorted(Ctx* c, const void* sorted_labels_dev, int32_t* labels_out);
int project_all(Ctx* c, const gsx_camera* cam, const float* dx, const float* dy, const float* dz, int64_t n,
                int32_t* x_host, int32_t* y_host, const uint32_t* perm);
// sort.hip
int radix_sort_pairs(Ctx* c, uint32_t* k0, uint32_t* v0, uint32_t* k1, uint32_t* v1, long long n, int bits,
                     int* result_in);
int radix_sort_pairs_dev(Ctx* c, uint32_t* k0, uint32_t* v0, uint32_t* k1, uint32_t* v1, long long n,
                         const unsigned long long* n_dev, int bits, int* result_in);
int radix_sort_pairs_drop(Ctx* c, uint32_t* k0, uint32_t* v0, uint32_t* k1, uint32_t* v1, long long n,
                          const unsigned long long* n_dev, int bits, int* result_in, unsigned long long* n_kept);  // elements with key 0xffffffff are left out
int radix_sort_values_wide(Ctx* c, const uint32_t* k0, const uint32_t* v0, uint32_t* v1, long long n, const unsigned long long* n_dev,
                           int bits, int2* ranges, int nranges);  // one pass over <= 11 key bits; ranges[d] = slots of key d
int spatial_sort_positions(Ctx* c);
int vote_culled(Ctx* c, int64_t* out, bool reset);
int second_stream(Ctx* c);
int filter_check(Ctx* c, double* out);
int kmeans(Ctx* c, int64_t n, const float* points, const float* colors, int k, const int64_t* init_index, int max_iter,
           double tol, int32_t* labels_out, float* centroids_out, int32_t* iterations_out, int32_t* converged_out);
void debug_cull_planes(const gsx_camera* cam, double* out);
// render.hip
int upload_splats(Ctx* c, int64_t n, const float* xyz, const float* scale, const float* rot, const float* opacity,
                  const float* f_dc, const int32_t* labels);
int upload_sh(Ctx* c, const float* f_rest, int deg);
int render_view(Ctx* c, const gsx_camera* cam, int W, int H, float* rgba_out);
int render_views(Ctx* c, int n, const gsx_camera* cams, int W, int H, float* const* rgba_out);
void render_release_twin(Ctx* c);
int hit_test(Ctx* c, const gsx_camera* cam, int W, int H, double x, double y, int32_t* label_out, int64_t* index_out);
int render_debug(Ctx* c, uint8_t* buffer_out, uint32_t* order_out, uint32_t* tex_out, uint32_t* bucket_out);
void fill_view_desc(ViewDesc& vd, const gsx_camera* cam, int seg_w, int seg_h, int img_w, int img_h);

}  // namespace gsx
