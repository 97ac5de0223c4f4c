// libgsx.so — C ABI entry points (include/gsx.h): context, scene upload, profiling, dispatch.
// No exception crosses the boundary; every entry returns a gsx_status.
#include <cstring>
#include <exception>
#include <new>

#include "gsx_ctx.hpp"

namespace gsx {

static thread_local std::string g_err;

void set_global_error(const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
}

int fail(Ctx* c, int code, const char* fmt, ...) {
    char buf[768];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    else g_err = buf;
    return code;
}

ProfScope::ProfScope(Ctx* ctx, const char* name, hipStream_t on) : c(ctx), st(on ? on : ctx->stream) {
    if (!c->prof_on) return;
    int id = -1;
    for (size_t i = 0; i < c->prof_names.size(); ++i)
        if (c->prof_names[i] == name) id = (int)i;
    if (id < 0) {
        c->prof_names.emplace_back(name);
        id = (int)c->prof_names.size() - 1;
    }
    auto get = [&](hipEvent_t& e) {
        if (!c->event_pool.empty()) {
            e = c->event_pool.back();
            c->event_pool.pop_back();
            return true;
        }
        return hipEventCreate(&e) == hipSuccess;
    };
    if (!get(ev.start) || !get(ev.stop)) return;
    ev.name_id = id;
    active = hipEventRecord(ev.start, st) == hipSuccess;
}

ProfScope::~ProfScope() {
    if (!active) return;
    (void)hipEventRecord(ev.stop, st);
    c->prof_events.push_back(ev);
}

static void prof_drain(Ctx* c) {
    for (auto& e : c->prof_events) {
        float ms = 0.f;
        if (hipEventSynchronize(e.stop) == hipSuccess && hipEventElapsedTime(&ms, e.start, e.stop) == hipSuccess) {
            auto& acc = c->prof_acc[c->prof_names[e.name_id]];
            acc.first += 1;
            acc.second += ms;
        }
        c->event_pool.push_back(e.start);
        c->event_pool.push_back(e.stop);
    }
    c->prof_events.clear();
}

// No exception may cross the C boundary: std::bad_alloc from a host-side vector sized by the caller's counts,
// std::system_error from a worker thread that cannot be started, ... become a status + gsx_last_error text.
template <class F>
static int guard(Ctx* c, const char* who, F f) {
    try {
        return f();
    } catch (const std::bad_alloc&) {
        return fail(c, GSX_E_INVALID, "%s: out of host memory", who);
    } catch (const std::exception& e) {
        return fail(c, GSX_E_INVALID, "%s: %s", who, e.what());
    } catch (...) {
        return fail(c, GSX_E_INVALID, "%s: unknown failure", who);
    }
}

}  // namespace gsx

using gsx::Ctx;

#define CTX_OR_FAIL(ctx)                                                     \
    Ctx* c = reinterpret_cast<Ctx*>(ctx);                                    \
    if (!c) return gsx::fail(nullptr, GSX_E_INVALID, "%s: ctx is NULL", __func__)

extern "C" {

int gsx_abi_version(void) { return GSX_ABI_VERSION; }

int gsx_device_count(void) {
    int count = 0;
    return hipGetDeviceCount(&count) == hipSuccess ? count : 0;
}

int gsx_create(int device_id, gsx_ctx** out) {
    if (!out) return gsx::fail(nullptr, GSX_E_INVALID, "gsx_create: out is NULL");
    *out = nullptr;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        return gsx::fail(nullptr, GSX_E_HIP, "gsx_create: no HIP device (%s); libgsx has no CPU fallback",
                         e == hipSuccess ? "device count 0" : hipGetErrorString(e));
    if (device_id < 0 || device_id >= count)
        return gsx::fail(nullptr, GSX_E_INVALID, "gsx_create: device %d out of range [0,%d)", device_id, count);
    hipDeviceProp_t prop;
    GSX_HIP(nullptr, hipGetDeviceProperties(&prop, device_id));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return gsx::fail(nullptr, GSX_E_HIP, "gsx_create: device %d is %s; this library carries gfx950 code only",
                         device_id, prop.gcnArchName);
    GSX_HIP(nullptr, hipSetDevice(device_id));
    Ctx* c = new (std::nothrow) Ctx();
    if (!c) return gsx::fail(nullptr, GSX_E_INVALID, "gsx_create: out of host memory");
    c->device = device_id;
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        delete c;
        return gsx::fail(nullptr, GSX_E_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
    }
    *out = reinterpret_cast<gsx_ctx*>(c);
    return GSX_OK;
}

void gsx_destroy(gsx_ctx* ctx) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->stream2) (void)hipStreamSynchronize(c->stream2);  // an early vote stage may still read the buffers freed below
    gsx::prof_drain(c);
    for (auto ev : c->event_pool) (void)hipEventDestroy(ev);
    for (gsx::DevBuf* b : {&c->x, &c->y, &c->z, &c->perm, &c->sort_hist, &c->d_views, &c->d_cull, &c->d_cull_tally, &c->segpool, &c->errflag,
                           &c->cnt, &c->fv, &c->bcnt, &c->bcodes, &c->keys, &c->labels, &c->cand, &c->codes, &c->r_order, &c->r_buffer, &c->r_tex, &c->r_sh, &c->r_fdc, &c->r_shc, &c->r_image,
                           &c->r_ranges, &c->r_small, &c->r_scan, &c->r_depth, &c->r_bucket, &c->r_rect, &c->r_count,
                           &c->r_offset, &c->r_rec, &c->r_keys0, &c->r_keys1, &c->r_vals0, &c->r_vals1, &c->r_tile_order, &c->r_sat, &c->r_d0, &c->r_d1, &c->r_d2, &c->r_d3, &c->r_rects})
        b->release();
    gsx::render_release_twin(c);
    gsx::vote_release_host(c);
    (void)hipStreamDestroy(c->stream);
    delete c;
}

const char* gsx_last_error(const gsx_ctx* ctx) {
    const Ctx* c = reinterpret_cast<const Ctx*>(ctx);
    return c ? c->err.c_str() : gsx::g_err.c_str();
}

void* gsx_stream(gsx_ctx* ctx) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    return c ? (void*)c->stream : nullptr;
}

int gsx_set_option(gsx_ctx* ctx, const char* name, int64_t value) {
    CTX_OR_FAIL(ctx);
    if (!name) return gsx::fail(c, GSX_E_INVALID, "set_option: name is NULL");
    const std::string k(name);
    if (k == "spatial_sort") c->opt_spatial_sort = value != 0;
    else if (k == "xcd_swizzle") {
        if (value < 0 || value > 65536) return gsx::fail(c, GSX_E_INVALID, "set_option: xcd_swizzle must be in [0,65536]");
        c->opt_xcd_swizzle = (int)value;
    }
    else if (k == "seg_tiled") c->opt_seg_tiled = value != 0;
    else if (k == "tile_lpt") c->opt_tile_lpt = value != 0;
    else if (k == "exact_cull") c->opt_exact_cull = value != 0;
    else if (k == "render_phases") {
        if (value < 1 || value > 8) return gsx::fail(c, GSX_E_INVALID, "set_option: render_phases must be in [1,8]");
        c->opt_render_phases = (int)value;
    } else if (k == "render_phase_ratio") {
        if (value < 2 || value > 64) return gsx::fail(c, GSX_E_INVALID, "set_option: render_phase_ratio must be in [2,64]");
        c->opt_render_phase_ratio = (int)value;
    }
    else if (k == "render_multi_pre") c->opt_render_multi_pre = value != 0;
    else if (k == "render_bin32") c->opt_render_bin32 = value != 0;
    else if (k == "render_compact") c->opt_render_compact = value != 0;
    else if (k == "render_wide_sort") c->opt_render_wide_sort = value < 0 ? 0 : (value > 2 ? 2 : (int)value);
    else if (k == "render_share_stream") {
        if ((value != 0) != (c->opt_render_share_stream != 0)) gsx::render_release_twin(c);  // the extra frames' streams are made anew
        c->opt_render_share_stream = value != 0;
    }
    else if (k == "render_frames") {
        if (value < 1 || value > Ctx::kMaxFrames) return gsx::fail(c, GSX_E_INVALID, "set_option: render_frames must be in [1,%d]", (int)Ctx::kMaxFrames);
        c->opt_render_frames = (int)value;
    }
    else if (k == "blend_pk2") c->opt_blend_pk2 = value < 0 ? 0 : (value > 2 ? 2 : (int)value);
    else if (k == "exchange_slabs") {
        if (value < 1 || value > 64) return gsx::fail(c, GSX_E_INVALID, "set_option: exchange_slabs must be in [1,64]");
        c->opt_slabs = (int)value;
    } else if (k == "exchange_local") c->opt_local_codes = value != 0;
    else if (k == "fast_div") c->opt_fast_div = value != 0;
    else if (k == "lds_batch") c->opt_lds_batch = value != 0;
    else if (k == "flat_project") c->opt_flat_project = value != 0;
    else if (k == "seg_coarse") c->opt_seg_coarse = value != 0;
    else if (k == "batched_counts") c->opt_batched_counts = value != 0;
    else if (k == "wave_cull") c->opt_wave_cull = value != 0;
    else if (k == "filter_project") c->opt_filter_project = value != 0;
    else if (k == "host_pack") c->opt_host_pack = value != 0;
    else if (k == "labels_u8") c->opt_labels_u8 = value != 0;
    else if (k == "host_compact") c->opt_host_compact = value != 0;
    else if (k == "early_vote") {
        if (value < 0 || value > 2) return gsx::fail(c, GSX_E_INVALID, "set_option: early_vote must be 0, 1 or 2");
        c->opt_early_vote = (int)value;
    } else if (k == "early_replay") c->opt_early_replay = value != 0;
    else if (k == "early_vote_at") {
        if (value < 0 || value > 1000) return gsx::fail(c, GSX_E_INVALID, "set_option: early_vote_at must be in [0,1000] (permille of the announced views; 0 = from the hand-over rate)");
        c->opt_early_at = (int)value;
    }
    else if (k == "host_prefetch") gsx::set_host_prefetch((int)value);
    else if (k == "host_prefetch_burst") gsx::set_host_prefetch_burst(value != 0);
#ifdef GSX_EXPERIMENTS
    else if (k == "ablate") c->opt_ablate = (int)value;
#endif
    else if (k == "host_threads") {
        if (value < 0 || value > 256) return gsx::fail(c, GSX_E_INVALID, "set_option: host_threads must be in [0,256]");
        if ((int)value != c->opt_host_threads) {
            delete c->workers;  // re-created with the new size at the next host-side hand-over
            c->workers = nullptr;
        }
        c->opt_host_threads = (int)value;
    }
    else if (k == "vote_unroll") {
        if (value != 2 && value != 4 && value != 8)
            return gsx::fail(c, GSX_E_INVALID, "set_option: vote_unroll must be 2, 4 or 8");
        c->opt_vote_unroll = (int)value;
    } else
        return gsx::fail(c, GSX_E_INVALID, "set_option: unknown option '%s'", name);
    return GSX_OK;
}

int gsx_synchronize(gsx_ctx* ctx) {
    CTX_OR_FAIL(ctx);
    GSX_HIP(c, hipSetDevice(c->device));
    {
        const int rc = gsx::vote_flush_pending(c);
        if (rc) return rc;
    }
    GSX_HIP(c, hipStreamSynchronize(c->stream));
    return GSX_OK;
}

// ---- scene ---------------------------------------------------------------------------------------
static int alloc_positions(Ctx* c, int64_t n) {
    GSX_HIP(c, hipSetDevice(c->device));
    const size_t bytes = sizeof(float) * (size_t)(n > 0 ? n : 1);
    GSX_HIP(c, c->x.ensure(bytes));
    GSX_HIP(c, c->y.ensure(bytes));
    GSX_HIP(c, c->z.ensure(bytes));
    c->n = n;
    c->n_pad = (n + 255) / 256 * 256;
    c->vote_begun = false;  // planes are sized by n: a new scene needs a new vote_begin
    c->labels_valid = false;
    return GSX_OK;
}

int gsx_upload_positions(gsx_ctx* ctx, int64_t n, const float* x, const float* y, const float* z) {
    CTX_OR_FAIL(ctx);
    if (n < 0 || (n > 0 && (!x || !y || !z))) return gsx::fail(c, GSX_E_INVALID, "upload_positions: bad arguments");
    if (n > (int64_t)1 << 31) return gsx::fail(c, GSX_E_UNSUPPORTED, "upload_positions: n > 2^31");
    int rc = alloc_positions(c, n);
    if (rc) return rc;
    if (n == 0) return GSX_OK;
    GSX_HIP(c, hipMemcpyAsync(c->x.p, x, sizeof(float) * n, hipMemcpyHostToDevice, c->stream));
    GSX_HIP(c, hipMemcpyAsync(c->y.p, y, sizeof(float) * n, hipMemcpyHostToDevice, c->stream));
    GSX_HIP(c, hipMemcpyAsync(c->z.p, z, sizeof(float) * n, hipMemcpyHostToDevice, c->stream));
    GSX_HIP(c, hipStreamSynchronize(c->stream));
    c->sorted = false;
    if (c->opt_spatial_sort) return gsx::spatial_sort_positions(c);
    return GSX_OK;
}

int gsx_upload_positions_strided(gsx_ctx* ctx, int64_t n, const void* base, int64_t stride_bytes, int64_t off_x,
                                 int64_t off_y, int64_t off_z) {
    CTX_OR_FAIL(ctx);
    if (n < 0 || (n > 0 && !base) || stride_bytes < 4 || off_x < 0 || off_y < 0 || off_z < 0)
        return gsx::fail(c, GSX_E_INVALID, "upload_positions_strided: bad arguments");
    if (n > (int64_t)1 << 31) return gsx::fail(c, GSX_E_UNSUPPORTED, "upload_positions_strided: n > 2^31");
    return gsx::guard(c, __func__, [&] {
        // AoS -> SoA transpose on the host (rows are 248+ bytes in a 3DGS PLY; only 12 are wanted)
        std::vector<float> sx((size_t)n), sy((size_t)n), sz((size_t)n);
        const char* b = static_cast<const char*>(base);
        for (int64_t i = 0; i < n; ++i) {
            const char* row = b + i * stride_bytes;
            std::memcpy(&sx[i], row + off_x, 4);
            std::memcpy(&sy[i], row + off_y, 4);
            std::memcpy(&sz[i], row + off_z, 4);
        }
        return gsx_upload_positions(ctx, n, sx.data(), sy.data(), sz.data());
    });
}

int64_t gsx_num_gaussians(const gsx_ctx* ctx) {
    const Ctx* c = reinterpret_cast<const Ctx*>(ctx);
    return c ? c->n : 0;
}

// ---- projection probe ----------------------------------------------------------------------------
int gsx_project_one(gsx_ctx* ctx, const float pos[3], const gsx_camera* cam, int32_t* x, int32_t* y,
                    int32_t* visible) {
    CTX_OR_FAIL(ctx);
    if (!pos || !cam || !x || !y || !visible) return gsx::fail(c, GSX_E_INVALID, "project_one: NULL argument");
    GSX_HIP(c, hipSetDevice(c->device));
    gsx::DevBuf tmp;
    GSX_HIP(c, tmp.ensure(3 * sizeof(float)));
    hipError_t e = hipMemcpy(tmp.p, pos, 3 * sizeof(float), hipMemcpyHostToDevice);
    int rc = GSX_OK;
    if (e != hipSuccess) rc = gsx::fail(c, GSX_E_HIP, "project_one: H2D failed: %s", hipGetErrorString(e));
    if (!rc) rc = gsx::project_all(c, cam, tmp.as<float>(), tmp.as<float>() + 1, tmp.as<float>() + 2, 1, x, y, nullptr);
    if (!rc) *visible = (*x >= 0) ? 1 : 0;
    return rc;
}

int gsx_project_all(gsx_ctx* ctx, const gsx_camera* cam, int32_t* x, int32_t* y) {
    CTX_OR_FAIL(ctx);
    if (!cam || !x || !y) return gsx::fail(c, GSX_E_INVALID, "project_all: NULL argument");
    return gsx::guard(c, __func__, [&] { return gsx::project_all(c, cam, c->x.as<float>(), c->y.as<float>(), c->z.as<float>(), c->n, x, y,
                            c->sorted ? c->perm.as<uint32_t>() : nullptr); });
}

// ---- vote ----------------------------------------------------------------------------------------
int gsx_vote_begin(gsx_ctx* ctx, int32_t n_classes, int32_t first_view, int32_t total_views) {
    CTX_OR_FAIL(ctx);
    return gsx::guard(c, __func__, [&] { return gsx::vote_begin(c, n_classes, first_view, total_views); });
}
int gsx_vote_view(gsx_ctx* ctx, const gsx_camera* cam, const void* seg, int32_t seg_dtype, int32_t seg_w,
                  int32_t seg_h, int32_t img_w, int32_t img_h) {
    CTX_OR_FAIL(ctx);
    return gsx::guard(c, __func__, [&] { return gsx::vote_view(c, cam, seg, seg_dtype, seg_w, seg_h, img_w, img_h); });
}
int gsx_vote_view_device(gsx_ctx* ctx, const gsx_camera* cam, const void* seg_dev, int32_t seg_dtype, int32_t seg_w,
                         int32_t seg_h, int32_t img_w, int32_t img_h) {
    CTX_OR_FAIL(ctx);
    return gsx::guard(c, __func__, [&] { return gsx::vote_views_device(c, 1, cam, &seg_dev, seg_dtype, seg_w, seg_h, img_w, img_h); });
}
int gsx_vote_views_device(gsx_ctx* ctx, int32_t n, const gsx_camera* cams, const void* const* segs_dev, int32_t seg_dtype,
                          int32_t seg_w, int32_t seg_h, int32_t img_w, int32_t img_h) {
    CTX_OR_FAIL(ctx);
    return gsx::guard(c, __func__, [&] { return gsx::vote_views_device(c, n, cams, segs_dev, seg_dtype, seg_w, seg_h, img_w, img_h); });
}
int64_t gsx_debug_workers_stress(int32_t threads, int32_t runs, int32_t max_parts) {
    try {
        return gsx::workers_stress(threads, runs, max_parts);
    } catch (...) {
        return -1;
    }
}
int gsx_debug_widen_labels(int32_t threads, const uint8_t* bins, int64_t n, int32_t* labels_out) {
    if (n < 0 || (n > 0 && (!bins || !labels_out))) return GSX_E_INVALID;
    try {
        if (threads > 1) {
            gsx::Workers pool(threads);
            gsx::host_widen_labels(&pool, labels_out, bins, (size_t)n);
        } else {
            gsx::host_widen_labels(nullptr, labels_out, bins, (size_t)n);
        }
        return GSX_OK;
    } catch (...) {
        return GSX_E_HIP;  /* the pool could not be created (thread / memory exhaustion) */
    }
}
int gsx_debug_host_pack_compact(const void* seg, int32_t seg_dtype, int32_t w, int32_t h, int32_t n_classes, int32_t threads, uint8_t* out,
                                int64_t out_cap, int64_t* bytes, int64_t* table_bytes, int64_t* stream_off, int32_t* bad) {
    try {
        return gsx::debug_host_pack_compact(seg, seg_dtype, w, h, n_classes, threads, out, out_cap, bytes, table_bytes, stream_off, bad);
    } catch (...) {
        return GSX_E_HIP;
    }
}
int gsx_debug_host_pack(const void* seg, int32_t seg_dtype, int32_t w, int32_t h, int32_t n_classes, int32_t tiled,
                        int32_t coarse, int32_t threads, uint8_t* out, int64_t out_cap, int64_t* bytes, int64_t* coarse_off,
                        int32_t* bad) {
    return gsx::guard(nullptr, __func__, [&] { return gsx::debug_host_pack(seg, seg_dtype, w, h, n_classes, tiled, coarse, threads, out, out_cap, bytes, coarse_off, bad); });
}
int32_t gsx_vote_num_views(const gsx_ctx* ctx) {
    const Ctx* c = reinterpret_cast<const Ctx*>(ctx);
    return c ? (int32_t)c->views.size() : 0;
}
int gsx_vote_rewind(gsx_ctx* ctx) {
    CTX_OR_FAIL(ctx);
    return gsx::guard(c, __func__, [&] { return gsx::vote_rewind(c); });
}
int gsx_vote_finalize(gsx_ctx* ctx, int32_t* labels_out) {
    CTX_OR_FAIL(ctx);
    return gsx::guard(c, __func__, [&] { return gsx::vote_finalize(c, labels_out); });
}
void* gsx_vote_labels_device(gsx_ctx* ctx) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    return (c && c->labels_valid) ? c->labels.p : nullptr;
}
int gsx_vote_flush(gsx_ctx* ctx) {
    CTX_OR_FAIL(ctx);
    return gsx::vote_flush(c);
}
void* gsx_vote_counts_device(gsx_ctx* ctx, int64_t* n_int32_words) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c || !c->vote_begun) return nullptr;
    if (n_int32_words) *n_int32_words = (int64_t)c->bins * c->n_pad * (c->wide ? 2 : 1) / 4;
    return c->cnt.p;
}
int gsx_vote_tiebreak_keys(gsx_ctx* ctx) {
    CTX_OR_FAIL(ctx);
    return gsx::vote_tiebreak_keys(c);
}
void* gsx_vote_keys_device(gsx_ctx* ctx, int64_t* n_int32_words) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c || !c->vote_begun) return nullptr;
    if (n_int32_words) *n_int32_words = c->n_pad;
    return c->keys.p;
}
int gsx_vote_labels_from_keys(gsx_ctx* ctx, int32_t* labels_out) {
    CTX_OR_FAIL(ctx);
    return gsx::guard(c, __func__, [&] { return gsx::vote_labels_from_keys(c, labels_out); });
}
void* gsx_vote_first_device(gsx_ctx* ctx, int64_t* n_int32_words) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c || !c->vote_begun) return nullptr;
    if (n_int32_words) *n_int32_words = (int64_t)c->bins * c->n_pad * (c->wide ? 2 : 1) / 4;
    return c->fv.p;
}
int64_t gsx_vote_slab_size(const gsx_ctx* ctx) {
    const Ctx* c = reinterpret_cast<const Ctx*>(ctx);
    return (c && c->vote_begun) ? c->sn : 0;
}
int gsx_vote_slab_reduce(gsx_ctx* ctx, const void* recv_counts_dev, const void* recv_first_dev) {
    CTX_OR_FAIL(ctx);
    return gsx::guard(c, __func__, [&] { return gsx::vote_slab_reduce(c, recv_counts_dev, recv_first_dev); });
}
int gsx_vote_flush_counts(gsx_ctx* ctx) {
    CTX_OR_FAIL(ctx);
    return gsx::vote_flush_counts(c);
}
int gsx_vote_slab_totals(gsx_ctx* ctx, const void* recv_counts_dev) {
    CTX_OR_FAIL(ctx);
    return gsx::guard(c, __func__, [&] { return gsx::vote_slab_totals(c, recv_counts_dev); });
}
void* gsx_vote_cand_device(gsx_ctx* ctx, int64_t* n_int32_words) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c || !c->vote_begun) return nullptr;
    if (n_int32_words) *n_int32_words = 8 * c->sn;
    return c->cand.p;
}
int gsx_vote_tie_codes(gsx_ctx* ctx, const void* cand_all_dev) {
    CTX_OR_FAIL(ctx);
    return gsx::guard(c, __func__, [&] { return gsx::vote_tie_codes(c, cand_all_dev); });
}
void* gsx_vote_codes_device(gsx_ctx* ctx, int64_t* n_int32_words) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c || !c->vote_begun) return nullptr;
    if (n_int32_words) *n_int32_words = c->n_pad / 2;
    return c->codes.p;
}
int gsx_vote_tie_resolve(gsx_ctx* ctx, const void* recv_codes_dev) {
    CTX_OR_FAIL(ctx);
    return gsx::guard(c, __func__, [&] { return gsx::vote_tie_resolve(c, recv_codes_dev); });
}
int gsx_vote_labels_from_sorted(gsx_ctx* ctx, const void* sorted_labels_dev, int32_t* labels_out) {
    CTX_OR_FAIL(ctx);
    return gsx::guard(c, __func__, [&] { return gsx::vote_labels_from_sorted(c, sorted_labels_dev, labels_out); });
}
int gsx_vote_export(gsx_ctx* ctx, int64_t reserve_bytes, void* blobs_out, void** pool_dev, int64_t* pool_bytes) {
    CTX_OR_FAIL(ctx);
    return gsx::guard(c, __func__, [&] { return gsx::vote_export(c, reserve_bytes, blobs_out, pool_dev, pool_bytes); });
}
int64_t gsx_vote_pool_bytes(const gsx_ctx* ctx) {
    const Ctx* c = reinterpret_cast<const Ctx*>(ctx);
    return (c && c->vote_begun && !c->pool_base) ? (int64_t)((c->seg_used + 255) / 256 * 256) : 0;
}
int64_t gsx_vote_link_bytes(const gsx_ctx* ctx) {
    const Ctx* c = reinterpret_cast<const Ctx*>(ctx);
    return (c && c->vote_begun) ? (int64_t)c->compact_bytes : 0;
}
int64_t gsx_vote_early_views(const gsx_ctx* ctx) {
    const Ctx* c = reinterpret_cast<const Ctx*>(ctx);
    return c && (c->early_state == 1 || c->early_batches > 0) ? c->early_done : 0;
}
int gsx_vote_import(gsx_ctx* ctx, int32_t n_parts, const int32_t* part_views, const int64_t* part_offsets, const void* blobs,
                    const void* pool_all_dev, int64_t pool_all_bytes) {
    CTX_OR_FAIL(ctx);
    return gsx::guard(c, __func__, [&] { return gsx::vote_import(c, n_parts, part_views, part_offsets, blobs, pool_all_dev, pool_all_bytes); });
}
int gsx_vote_import_uniform(gsx_ctx* ctx, int32_t n_parts, const int32_t* part_views, const int64_t* part_offsets, const gsx_camera* cams,
                            int32_t seg_w, int32_t seg_h, int32_t img_w, int32_t img_h, const void* pool_all_dev, int64_t pool_all_bytes) {
    CTX_OR_FAIL(ctx);
    return gsx::guard(c, __func__, [&] { return gsx::vote_import_uniform(c, n_parts, part_views, part_offsets, cams, seg_w, seg_h, img_w, img_h, pool_all_dev, pool_all_bytes); });
}
int gsx_vote_import_undo(gsx_ctx* ctx) {
    CTX_OR_FAIL(ctx);
    return gsx::guard(c, __func__, [&] { return gsx::vote_import_undo(c); });
}
int gsx_vote_slab_labels(gsx_ctx* ctx, int32_t slab, int32_t slabs, int64_t* slab_size) {
    CTX_OR_FAIL(ctx);
    return gsx::guard(c, __func__, [&] { return gsx::vote_slab_labels(c, slab, slabs, slab_size); });
}
int gsx_host_threads(gsx_ctx* ctx) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c) return 0;
    try {
        return gsx::host_threads(c);  // may create the worker pool
    } catch (...) {
        return 1;  // no pool: maps are packed on the calling thread
    }
}
int gsx_default_host_threads(void) { return gsx::default_host_threads(); }
int gsx_vote_debug_planes(gsx_ctx* ctx, uint16_t* counts_out, uint16_t* first_out) {
    CTX_OR_FAIL(ctx);
    return gsx::guard(c, __func__, [&] { return gsx::vote_debug_planes(c, counts_out, first_out); });
}

// ---- rasterizer -----------------------------------------------------------------------------------
int gsx_upload_splats(gsx_ctx* ctx, int64_t n, const float* xyz, const float* scale, const float* rot,
                      const float* opacity, const float* f_dc, const int32_t* labels) {
    CTX_OR_FAIL(ctx);
    return gsx::guard(c, __func__, [&] { return gsx::upload_splats(c, n, xyz, scale, rot, opacity, f_dc, labels); });
}
int gsx_upload_sh(gsx_ctx* ctx, const float* f_rest, int32_t sh_degree) {
    CTX_OR_FAIL(ctx);
    return gsx::guard(c, __func__, [&] { return gsx::upload_sh(c, f_rest, sh_degree); });
}
int64_t gsx_num_splats(const gsx_ctx* ctx) {
    const Ctx* c = reinterpret_cast<const Ctx*>(ctx);
    return c ? c->rn : 0;
}
int gsx_render_view(gsx_ctx* ctx, const gsx_camera* cam, int32_t width, int32_t height, float* rgba_out) {
    CTX_OR_FAIL(ctx);
    return gsx::guard(c, __func__, [&] { return gsx::render_view(c, cam, width, height, rgba_out); });
}
int gsx_render_views(gsx_ctx* ctx, int32_t n, const gsx_camera* cams, int32_t width, int32_t height, float* const* rgba_out) {
    CTX_OR_FAIL(ctx);
    return gsx::guard(c, __func__, [&] { return gsx::render_views(c, n, cams, width, height, rgba_out); });
}
void* gsx_render_image_device(gsx_ctx* ctx) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    return c ? c->r_image.p : nullptr;
}
int64_t gsx_render_num_pairs(const gsx_ctx* ctx) {
    const Ctx* c = reinterpret_cast<const Ctx*>(ctx);
    return c ? (int64_t)c->r_P : 0;
}
int64_t gsx_render_num_pairs_consumed(const gsx_ctx* ctx) {
    const Ctx* c = reinterpret_cast<const Ctx*>(ctx);
    return c ? (int64_t)c->r_consumed : 0;
}
int gsx_hit_test(gsx_ctx* ctx, const gsx_camera* cam, int32_t width, int32_t height, double x, double y,
                 int32_t* label_out, int64_t* index_out) {
    CTX_OR_FAIL(ctx);
    return gsx::guard(c, __func__, [&] { return gsx::hit_test(c, cam, width, height, x, y, label_out, index_out); });
}
int gsx_render_debug(gsx_ctx* ctx, uint8_t* buffer_out, uint32_t* order_out, uint32_t* texdata_out,
                     uint32_t* bucket_out) {
    CTX_OR_FAIL(ctx);
    return gsx::guard(c, __func__, [&] { return gsx::render_debug(c, buffer_out, order_out, texdata_out, bucket_out); });
}

int gsx_kmeans(gsx_ctx* ctx, int64_t n, const float* points, const float* colors, int32_t k, const int64_t* init_index,
               int32_t max_iter, double tol, int32_t* labels_out, float* centroids_out, int32_t* iterations_out,
               int32_t* converged_out) {
    CTX_OR_FAIL(ctx);
    return gsx::guard(c, __func__, [&] { return gsx::kmeans(c, n, points, colors, k, init_index, max_iter, tol, labels_out, centroids_out, iterations_out,
                       converged_out); });
}

int gsx_vote_culled(gsx_ctx* ctx, int64_t* wave_views, int32_t reset) {
    CTX_OR_FAIL(ctx);
    if (!wave_views) return gsx::fail(c, GSX_E_INVALID, "vote_culled: NULL argument");
    return gsx::guard(c, __func__, [&] { return gsx::vote_culled(c, wave_views, reset != 0); });
}

int gsx_debug_filter_check(gsx_ctx* ctx, double* out) {
    CTX_OR_FAIL(ctx);
    if (!out) return gsx::fail(c, GSX_E_INVALID, "debug_filter_check: NULL argument");
    return gsx::guard(c, __func__, [&] { return gsx::filter_check(c, out); });
}

int gsx_debug_cull_planes(const gsx_camera* cam, double* out) {
    if (!cam || !out) return gsx::fail(nullptr, GSX_E_INVALID, "debug_cull_planes: NULL argument");
    gsx::debug_cull_planes(cam, out);
    return GSX_OK;
}

int gsx_debug_sort_pairs(gsx_ctx* ctx, uint32_t* keys, uint32_t* values, int64_t n, int32_t bits) {
    CTX_OR_FAIL(ctx);
    if (n < 0 || bits < 0 || bits > 32 || (n > 0 && (!keys || !values)))
        return gsx::fail(c, GSX_E_INVALID, "debug_sort_pairs: bad arguments");
    if (n == 0) return GSX_OK;
    GSX_HIP(c, hipSetDevice(c->device));
    gsx::DevBuf k0, v0, k1, v1;
    const size_t nb = sizeof(uint32_t) * (size_t)n;
    GSX_HIP(c, k0.ensure(nb));
    GSX_HIP(c, v0.ensure(nb));
    GSX_HIP(c, k1.ensure(nb));
    GSX_HIP(c, v1.ensure(nb));
    GSX_HIP(c, hipMemcpyAsync(k0.p, keys, nb, hipMemcpyHostToDevice, c->stream));
    GSX_HIP(c, hipMemcpyAsync(v0.p, values, nb, hipMemcpyHostToDevice, c->stream));
    int where = 0;
    int rc = gsx::radix_sort_pairs(c, k0.as<uint32_t>(), v0.as<uint32_t>(), k1.as<uint32_t>(), v1.as<uint32_t>(), n, bits,
                                   &where);
    if (rc) return rc;
    GSX_HIP(c, hipMemcpyAsync(keys, where ? k1.p : k0.p, nb, hipMemcpyDeviceToHost, c->stream));
    GSX_HIP(c, hipMemcpyAsync(values, where ? v1.p : v0.p, nb, hipMemcpyDeviceToHost, c->stream));
    GSX_HIP(c, hipStreamSynchronize(c->stream));
    return GSX_OK;
}

int gsx_debug_sort_pairs_drop(gsx_ctx* ctx, uint32_t* keys, uint32_t* values, int64_t n, int32_t bits, int64_t* kept_out) {
    CTX_OR_FAIL(ctx);
    if (n < 1 || bits < 1 || bits > 32 || !keys || !values || !kept_out)
        return gsx::fail(c, GSX_E_INVALID, "debug_sort_pairs_drop: bad arguments");
    GSX_HIP(c, hipSetDevice(c->device));
    gsx::DevBuf k0, v0, k1, v1, cnt;
    const size_t nb = sizeof(uint32_t) * (size_t)n;
    GSX_HIP(c, k0.ensure(nb));
    GSX_HIP(c, v0.ensure(nb));
    GSX_HIP(c, k1.ensure(nb));
    GSX_HIP(c, v1.ensure(nb));
    GSX_HIP(c, cnt.ensure(8));
    GSX_HIP(c, hipMemcpyAsync(k0.p, keys, nb, hipMemcpyHostToDevice, c->stream));
    GSX_HIP(c, hipMemcpyAsync(v0.p, values, nb, hipMemcpyHostToDevice, c->stream));
    // (the result's slots behind the kept elements are not written: let them read as 0xff bytes, never as stale memory)
    GSX_HIP(c, hipMemsetAsync(k1.p, 0xff, nb, c->stream));
    GSX_HIP(c, hipMemsetAsync(v1.p, 0xff, nb, c->stream));
    int where = 0;
    int rc = gsx::radix_sort_pairs_drop(c, k0.as<uint32_t>(), v0.as<uint32_t>(), k1.as<uint32_t>(), v1.as<uint32_t>(), n, nullptr,
                                        bits, &where, cnt.as<unsigned long long>());
    if (rc) return rc;
    unsigned long long kept = 0;
    GSX_HIP(c, hipMemcpyAsync(&kept, cnt.p, 8, hipMemcpyDeviceToHost, c->stream));
    GSX_HIP(c, hipMemcpyAsync(keys, where ? k1.p : k0.p, nb, hipMemcpyDeviceToHost, c->stream));
    GSX_HIP(c, hipMemcpyAsync(values, where ? v1.p : v0.p, nb, hipMemcpyDeviceToHost, c->stream));
    GSX_HIP(c, hipStreamSynchronize(c->stream));
    *kept_out = (int64_t)kept;
    return GSX_OK;
}

// ---- profiling -----------------------------------------------------------------------------------
int gsx_profile_enable(gsx_ctx* ctx, int on) {
    CTX_OR_FAIL(ctx);
    c->prof_on = on != 0;
    return GSX_OK;
}
int gsx_profile_reset(gsx_ctx* ctx) {
    CTX_OR_FAIL(ctx);
    GSX_HIP(c, hipSetDevice(c->device));
    gsx::prof_drain(c);
    c->prof_acc.clear();
    return GSX_OK;
}
const char* gsx_profile_name(gsx_ctx* ctx, int32_t index) {
    Ctx* c = reinterpret_cast<Ctx*>(ctx);
    if (!c || index < 0 || (size_t)index >= c->prof_names.size()) return nullptr;
    return c->prof_names[(size_t)index].c_str();
}
int gsx_profile_get(gsx_ctx* ctx, const char* name, int64_t* launches, double* total_ms) {
    CTX_OR_FAIL(ctx);
    if (!name) return gsx::fail(c, GSX_E_INVALID, "profile_get: name is NULL");
    GSX_HIP(c, hipSetDevice(c->device));
    gsx::prof_drain(c);
    auto it = c->prof_acc.find(name);
    if (launches) *launches = it == c->prof_acc.end() ? 0 : it->second.first;
    if (total_ms) *total_ms = it == c->prof_acc.end() ? 0.0 : it->second.second;
    return GSX_OK;
}

}  // extern "C"
