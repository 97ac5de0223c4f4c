// vote.hip — majority-vote labeler kernels for gfx950 (MI355X) and their host drivers.
//
// Replaces the double loop of assign_labels (deep_learning_segmentation.py:255-295) and its arg-max
// (:297-308); project() restates project_gaussian (:43-82) operation for operation in fp64.
// This translation unit MUST be compiled with -ffp-contract=off: every fma below is explicit and
// every other product/sum must round separately, exactly as the reference's Python floats do.
//
// Data layout in HBM
//   positions      SoA  x[n], y[n], z[n] f32                      (coalesced 4 B/lane loads)
//   views          ViewDesc[V], 256 B each (176 hot), wave-uniform -> four scalar loads, operands live in SGPRs
//   seg pool       u8 maps, value = label+1 (bin index), one after another, each stored as strips of 16 pixel
//                  columns (rows of a strip back to back, 16 B each; 8 rows = one 128-B L2 line): the pixels
//                  a wave gathers are a compact patch, so they fall into few lines; 1 B gathers
//   planes         cnt[slab][bins][sn], fv[slab][bins][sn]  u8 or u16 (16 bit only for the all-reduce
//                  protocol with > 255 views in total): bin-major so that a wave's 64 Gaussians touch 64
//                  consecutive elements of a row; slab = one rank's share in the all-to-all protocols
//                  (a single slab is plain [bins][n_pad])
//   keys, labels   int32[n_pad]; cand u32[8][sn]; codes u16[n_pad] (exchange protocol v3)
//
// Kernels
//   vote_fused_labels_kernel   single GPU: all staged views in one launch, labels straight out
//   vote_fused_planes_kernel   multi-GPU v1/v2 (and > 255 views): count + first-view planes
//   vote_fused_counts_kernel   multi-GPU v3: count plane only; vote_slab_totals / vote_tie / vote_tie_resolve
//   vote_keys / vote_labels / vote_slab_reduce / unpermute_labels   the rank-local ends of the protocols
//
// Kernel design (Gaussian-major, all staged views in one launch)
//   one thread = one Gaussian; it walks the views in REVERSE order and keeps its private vote
//   histogram in LDS (row stride an odd number of dwords: a wave that votes one label hits 32
//   different banks).  Reverse order makes the reference's tie rule an online rule: a label that
//   reaches a count >= the best count so far takes over; the label whose FIRST vote (in forward
//   order) is earliest among the max-count labels is the last one to do so.  See DESIGN.md §3.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>

#include <type_traits>

#ifndef GSX_ABLATE
#define GSX_ABLATE 0  // product build; tools/ablate.sh builds timing-only variants (results invalid)
#endif
#if GSX_ABLATE && !defined(GSX_EXPERIMENTS)
#error "GSX_ABLATE variants are experiments: build them with -DGSX_EXPERIMENTS (make experiments), never into libgsx.so"
#endif

#include "gsx_ctx.hpp"

namespace gsx {

static constexpr int kBlock = 256;
static constexpr int kMaxBatch = 255;  // views per fused launch: LDS counters are 8 bit
int early_join(Ctx* c);

// -------------------------------------------------------------------------------------------------
// device: project_gaussian + seg-map addressing
// -------------------------------------------------------------------------------------------------
// One view's hot fields, held in SGPRs.  load_view() pins all of them behind ONE batch of scalar loads: left
// to itself hipcc sinks every s_load next to its first use, behind the early-out branches, and each view
// then pays ~5 exposed scalar-cache round trips (the kernel was waiting, not computing).
struct ViewRegs {
    double R[9], t[3], fx, fy, half_w, half_h, width, height;
    long long seg_off;
    int seg_w, unit_scale, seg_row_bytes, cam_w, cam_h;
    int coarse_row_bytes;
    unsigned coarse_delta;
    float hw32, hh32;
};

typedef int v16i __attribute__((ext_vector_type(16)));
typedef int v2i __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double chunk_f64(const v16i& v, int k) {
    v2i t;
    t.x = v[2 * k];
    t.y = v[2 * k + 1];
    return __builtin_bit_cast(double, t);
}

typedef int v8i __attribute__((ext_vector_type(8)));
typedef int v4i __attribute__((ext_vector_type(4)));
// The hot 184 bytes of a ViewDesc as four scalar loads (x16, x16, x8, x8), pinned as whole register tuples: the
// fields below are sub-registers of those tuples, no scalar move is spent on them.  (Pinning the 24 fields one
// by one cost 15 s_mov per view; the scalar unit is shared by the four SIMDs of a CU and every wave issues at
// most one instruction per turn, so scalar bookkeeping was costing as much as the fp64 arithmetic.)
__device__ __forceinline__ void unpack_core(ViewRegs& r, const v16i& a, const v16i& b) {
#pragma unroll
    for (int k = 0; k < 8; ++k) r.R[k] = chunk_f64(a, k);
    r.R[8] = chunk_f64(b, 0);
#pragma unroll
    for (int k = 0; k < 3; ++k) r.t[k] = chunk_f64(b, 1 + k);
    r.fx = chunk_f64(b, 4);
    r.fy = chunk_f64(b, 5);
    r.seg_off = __builtin_bit_cast(long long, chunk_f64(b, 6));
    r.seg_row_bytes = b[14];
    r.unit_scale = b[15];
}

__device__ __forceinline__ ViewRegs load_view(const ViewDesc* __restrict__ vp) {
    static_assert(offsetof(ViewDesc, R) == 0 && offsetof(ViewDesc, t) == 72 && offsetof(ViewDesc, fx) == 96 &&
                      offsetof(ViewDesc, seg_off) == 112 && offsetof(ViewDesc, seg_row_bytes) == 120 &&
                      offsetof(ViewDesc, unit_scale) == 124 && offsetof(ViewDesc, half_w) == 128 &&
                      offsetof(ViewDesc, width) == 144 && offsetof(ViewDesc, seg_w) == 160 &&
                      offsetof(ViewDesc, cam_w) == 168 && offsetof(ViewDesc, coarse_row_bytes) == 176 &&
                      offsetof(ViewDesc, coarse_delta) == 180 && offsetof(ViewDesc, hw32) == 184 && sizeof(ViewDesc) % 64 == 0,
                  "load_view reads the hot part of ViewDesc as 64 + 64 + 32 + 32 bytes");
    const char* q = reinterpret_cast<const char*>(vp);
    v16i a = *reinterpret_cast<const v16i*>(q);
    v16i b = *reinterpret_cast<const v16i*>(q + 64);
    v8i c = *reinterpret_cast<const v8i*>(q + 128);
    v8i d = *reinterpret_cast<const v8i*>(q + 160);
    asm volatile("" : "+s"(a), "+s"(b), "+s"(c), "+s"(d));
    ViewRegs r;
    unpack_core(r, a, b);
    v2i t;
    t.x = c[0], t.y = c[1];
    r.half_w = __builtin_bit_cast(double, t);
    t.x = c[2], t.y = c[3];
    r.half_h = __builtin_bit_cast(double, t);
    t.x = c[4], t.y = c[5];
    r.width = __builtin_bit_cast(double, t);
    t.x = c[6], t.y = c[7];
    r.height = __builtin_bit_cast(double, t);
    r.seg_w = d[0];
    r.cam_w = d[2];
    r.cam_h = d[3];
    r.coarse_row_bytes = d[4];
    r.coarse_delta = (unsigned)d[5];
    const int hw_bits = d[6], hh_bits = d[7];  // (a bit_cast straight from a vector element reads element 0)
    r.hw32 = __builtin_bit_cast(float, hw_bits);
    r.hh32 = __builtin_bit_cast(float, hh_bits);
    return r;
}


// OpenBLAS dgemv association of `R @ v` for a C-contiguous 3x3 (oracle/vote_oracle.c, header)
__device__ __forceinline__ double row_dot(const double* Rr, double v0, double v1, double v2) {
    return __builtin_fma(Rr[2], v2, __builtin_fma(Rr[0], v0, Rr[1] * v1));
}

// project_gaussian (dls.py:43-82).  Returns false where the reference returns None (:72-73, :80-82).
//
// DIV == kDivExact: the two IEEE-754 divisions of dls.py:76-77, operation for operation.
// DIV == kDivCertified: the same results from ONE reciprocal.  r ~ 1/pc2 (v_rcp_f64 + two Newton steps,
//   relative error <= 2^-48), s^ = (f*pc)*r + half is within 2^-30 of the reference's px = fl(fl(f*pc/pc2) + half)
//   whenever |px| <= 2^18, and far outside the frame (<= 2^16 pixels) otherwise.  If s^ is at least 2^-20 away
//   from every integer, floor(px) == floor(s^) is CERTAIN, and so are `0 <= px < width` (integer width) and the
//   truncation int(px).  A lane that is closer than that to an integer (~4e-6 of them), or whose arithmetic left
//   the finite range, or whose depth lies outside [2^-200, 2^200] (reciprocal not safely normal), falls through to
//   the exact divisions.  11 fewer fp64 instructions per visible pair.
// DIV == kDivFlat: the exact divisions again, but as ONE straight-line block: all three rows and both quotients are
//   evaluated unconditionally and the reference's tests (:72, :80) become a single predicate at the end.  The
//   branchy form serialises ~45 dependent fp64 instructions behind three divergent branches; here the three rows
//   and the two quotients are independent chains the scheduler interleaves.  Same operations, same operands,
//   same results; lanes the reference rejects early merely compute values nobody reads.
// DIV == kDivFlatSimple: kDivFlat for a batch whose views ALL have unit scale and tiled maps (the host checks): the
//   two wave-uniform tests per view disappear from the instruction stream.
// DIV == kDivFlatCoarse: kDivFlatSimple for a batch whose maps all carry a coarse level (see gather_chunk).
// DIV == kDivFiltCoarse: kDivFlatCoarse with the fp32 filter of project_filtered() in front of the two divisions.
enum { kDivExact = 0, kDivCertified = 1, kDivFlat = 2, kDivFlatSimple = 3, kDivFlatCoarse = 4, kDivFiltCoarse = 5 };
static constexpr int kDivModes = 6;

// Two IEEE-754 divisions by the same denominator, bit-identical to `ax / b` and `ay / b`.
// hipcc expands an fp64 division into div_scale(den), rcp, two Newton steps, div_scale(num), mul, fma, div_fmas,
// div_fixup.  Everything up to the refined reciprocal depends only on the SCALED denominator, which is the same
// for both quotients unless v_div_scale rescales for an extreme exponent gap; so the reciprocal chain (v_rcp_f64,
// which costs as much as 3.4 fp64 FMAs on gfx950, plus 4 FMAs) is evaluated once.  Lanes whose two scaled
// denominators differ redo the second quotient with the plain division (extreme-exponent tests only).
__device__ __forceinline__ void div2_shared(double ax, double ay, double b, double& qx, double& qy) {
    bool unused, vx, vy;
    const double sdx = __builtin_amdgcn_div_scale(ax, b, false, &unused);
    const double sdy = __builtin_amdgcn_div_scale(ay, b, false, &unused);
    const double nsd = -sdx;
    double r = __builtin_amdgcn_rcp(sdx);
    r = __builtin_fma(r, __builtin_fma(nsd, r, 1.0), r);
    r = __builtin_fma(r, __builtin_fma(nsd, r, 1.0), r);
    const double snx = __builtin_amdgcn_div_scale(ax, b, true, &vx);
    const double mx = snx * r;
    qx = __builtin_amdgcn_div_fixup(__builtin_amdgcn_div_fmas(__builtin_fma(nsd, mx, snx), r, mx, vx), b, ax);
    const double sny = __builtin_amdgcn_div_scale(ay, b, true, &vy);
    const double my = sny * r;
    qy = __builtin_amdgcn_div_fixup(__builtin_amdgcn_div_fmas(__builtin_fma(nsd, my, sny), r, my, vy), b, ay);
    // wave-uniform test on purpose: behind a per-lane `if` the compiler turns the fallback into a select and
    // evaluates the second reciprocal chain unconditionally
    const bool differ = __double_as_longlong(sdx) != __double_as_longlong(sdy);
    if (__builtin_expect(__builtin_amdgcn_ballot_w64(differ) != 0, 0)) {
        const double slow = ay / b;
        qy = differ ? slow : qy;
    }
}

template <int DIV>
__device__ __forceinline__ bool project(const ViewRegs& vd, double X, double Y, double Z, int& xi, int& yi) {
    const double pc2 = row_dot(vd.R + 6, X, Y, Z) + vd.t[2];
    if (DIV == kDivFlat || DIV == kDivFlatSimple || DIV == kDivFlatCoarse) {
        // one wave-uniform early-out keeps what matters of the branchy form: a wave whose 64 Gaussians are all
        // behind the camera (Morton order makes that the common way to be culled) skips the view
        if (__builtin_amdgcn_ballot_w64(pc2 > 0.0) == 0) return false;
        const double q0 = row_dot(vd.R + 0, X, Y, Z) + vd.t[0];
        const double q1 = row_dot(vd.R + 3, X, Y, Z) + vd.t[1];
        double qx, qy;
        div2_shared(vd.fx * q0, vd.fy * q1, pc2, qx, qy);
        const double fpx = qx + vd.half_w;  // dls.py:76
        const double fpy = qy + vd.half_h;  // dls.py:77
        const bool vis = (pc2 > 0.0) & (0.0 <= fpx) & (fpx < vd.width) & (0.0 <= fpy) & (fpy < vd.height);  // :72, :80
        if (!vis) return false;
        xi = (int)fpx;  // :81
        yi = (int)fpy;
        return true;
    }
    if (!(pc2 > 0.0)) return false;  // `pc2 <= 0` -> None; a NaN depth fails the bounds test below anyway
    const double pc0 = row_dot(vd.R + 0, X, Y, Z) + vd.t[0];
    const double pc1 = row_dot(vd.R + 3, X, Y, Z) + vd.t[1];
    const double ax = vd.fx * pc0, ay = vd.fy * pc1;
    // the certificate below assumes a normal reciprocal: 2^-200 <= pc2 <= 2^200 (exponent field 823..1223)
    if (DIV == kDivCertified && (unsigned)(((__double2hiint(pc2) >> 20) & 0x7ff) - 823) <= 400u) {
        double r = __builtin_amdgcn_rcp(pc2);
        r = __builtin_fma(__builtin_fma(-pc2, r, 1.0), r, r);
        r = __builtin_fma(__builtin_fma(-pc2, r, 1.0), r, r);
        const double sx = ax * r + vd.half_w, sy = ay * r + vd.half_h;
        const double flx = floor(sx), fly = floor(sy);
        const double frx = sx - flx, fry = sy - fly;
        const double lo = 9.5367431640625e-07, hi = 1.0 - 9.5367431640625e-07;  // 2^-20
        if ((frx >= lo) & (frx <= hi) & (fry >= lo) & (fry <= hi)) {  // false for NaN / infinity
            const int kx = (int)flx, ky = (int)fly;                    // saturating conversions
            if (((unsigned)kx >= (unsigned)vd.cam_w) | ((unsigned)ky >= (unsigned)vd.cam_h)) return false;
            xi = kx;
            yi = ky;
            return true;
        }
    }
    const double px = ax / pc2 + vd.half_w;  // dls.py:76
    const double py = ay / pc2 + vd.half_h;  // dls.py:77
    if (!((0.0 <= px) && (px < vd.width) && (0.0 <= py) && (py < vd.height))) return false;  // :80
    xi = (int)px;  // int() truncation, :81
    yi = (int)py;
    return true;
}

// ---- filtered exact projection ------------------------------------------------------------------------------------------
// project<kDivFlatSimple>() spends two thirds of its time on the two IEEE-754 divisions of dls.py:76-77 (v_div_scale, v_rcp_f64,
// two Newton steps, v_div_fmas, v_div_fixup: ~43 ns of ~95 per wave and view, tools/valu_rate.hip) to obtain px, py to the last
// bit - of which only floor() and the comparison with the frame are ever used.  Here the camera-space point (pc0, pc1, pc2) and
// the products ax = fx * pc0, ay = fy * pc1 are the reference's own fp64 values as before, but the perspective division runs
// in fp32: one v_rcp_f32 and two v_fma_f32 give px^, py^ with a PROVEN error bound E, and a lane whose px^ and py^ are both
// farther than E from every integer has floor(px^) == floor(px), floor(py^) == floor(py) for certain - which decides the
// visibility test (an integer frame: 0 <= px < W  <=>  0 <= floor(px) <= W - 1) and the pixel.  If any lane of the wave is
// closer than that to an integer (or its depth lies outside [2^-40, 2^40)), the whole wave takes the exact divisions for
// this view behind a wave-uniform branch: ~4 E of the lanes, i.e. one (wave, view) pair in ten at 1080p.  Bit-exact by
// construction; the bound:
//   u = 2^-24.  a^ = fl32(a) = a (1 + e1), z^ = fl32(pc2) = pc2 (1 + e2), |e1|, |e2| <= u (round to nearest; pc2 in
//   [2^-40, 2^40) keeps z^ and r^ normal, and the 2^-126 absolute error of an a below the normal range is < 2^-86 after the
//   multiplication by r^ <= 2^40); r^ = v_rcp_f32(z^) = (1 + e3) / z^ with |e3| <= 3 u (1 ulp by the ISA manual; the
//   exhaustive check gsx_debug_filter_check() measures it on the device, and the GPU suite asserts <= 3 u);
//   px^ = fl32(a^ r^ + hw) (fused, hw = W / 2 exact in fp32) = (a^ r^ + hw)(1 + e4), |e4| <= u.  With P = a / pc2 + hw:
//   |px^ - P| <= |P - hw| ((1 + u)(1 + 3u) / (1 - u) - 1) + u |px^| / (1 - u) <= 5.01 u |P - hw| + 1.01 u |px^|.
//   The reference's px = fl64(fl64(a / pc2) + hw) differs from P by <= 2^-52 (|P| + hw).
//   Inside or within one pixel of the frame, |P - hw| <= W / 2 + 1 and |px^| <= W + 1:  |px^ - px| < 3.6 W u.
//   E = 4 D u (kFilterK; D = the larger side of the largest frame of the batch, at least 64) leaves 10 % - 0.4 D u >= 2^-20 -
//   for the terms dropped above (a few u) and for the rounding of fract() on (-1, 0) and of `fract - 0.5` (<= 2^-25 each).
//   Outside that band no bound is needed: a lane is declared invisible only when floor(px^) lies outside [0, W - 1] while
//   px^ is at least E away from the integers, and then P is on the same side of the frame (for px^ >= W + 1 or px^ <= -1
//   the relative error 6.02 u cannot move P across the frame edge; between, the bound above applies); infinities saturate
//   the conversion and fail the range test, NaNs fail the comparison with E and send the wave to the exact path.
static constexpr double kFilterK = 4.0;
__device__ __forceinline__ int cvt_floor_i32(float v) {
    int k;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(k) : "v"(v));  // floor + saturating conversion in one instruction (NaN -> 0)
    return k;
}

// filt_h = 0.5 - E.  Returns what project<kDivFlatSimple>() returns, bit for bit.
__device__ __forceinline__ bool project_filtered(const ViewRegs& vd, float filt_h, double X, double Y, double Z, int& xi, int& yi) {
    const double pc2 = row_dot(vd.R + 6, X, Y, Z) + vd.t[2];
    const bool pos = pc2 > 0.0;  // dls.py:72
    if (__builtin_amdgcn_ballot_w64(pos) == 0) return false;
    const double q0 = row_dot(vd.R + 0, X, Y, Z) + vd.t[0];
    const double q1 = row_dot(vd.R + 3, X, Y, Z) + vd.t[1];
    const double ax = vd.fx * q0, ay = vd.fy * q1;
    {
        const float zf = (float)pc2, axf = (float)ax, ayf = (float)ay;
        const float r = __builtin_amdgcn_rcpf(zf);
        const float pxf = __builtin_fmaf(axf, r, vd.hw32), pyf = __builtin_fmaf(ayf, r, vd.hh32);
        const float dx = __builtin_amdgcn_fractf(pxf) - 0.5f, dy = __builtin_amdgcn_fractf(pyf) - 0.5f;
        // 2^-40 <= pc2 < 2^40, on the high word (a negative, zero, infinite or NaN depth fails)
        const bool zrange = (unsigned)(__double2hiint(pc2) - 0x3D700000) < (unsigned)(0x42700000 - 0x3D700000);
        const bool cert = zrange & (__builtin_fabsf(dx) <= filt_h) & (__builtin_fabsf(dy) <= filt_h);  // false for NaN
        const int kx = cvt_floor_i32(pxf), ky = cvt_floor_i32(pyf);
        if (__builtin_expect(__builtin_amdgcn_ballot_w64(pos & !cert) == 0, 1)) {
            xi = kx;
            yi = ky;
            return cert & ((unsigned)kx < (unsigned)vd.cam_w) & ((unsigned)ky < (unsigned)vd.cam_h);
        }
    }
    // some lane is too close to a pixel boundary: the exact divisions for the whole wave (dls.py:76-81)
    double qx, qy;
    div2_shared(ax, ay, pc2, qx, qy);
    const double fpx = qx + vd.half_w, fpy = qy + vd.half_h;
    const bool vis = pos & (0.0 <= fpx) & (fpx < vd.width) & (0.0 <= fpy) & (fpy < vd.height);
    if (!vis) return false;
    xi = (int)fpx;
    yi = (int)fpy;
    return true;
}

// The vote of one Gaussian in one view: the map byte (bin = label + 1) under its projection, or -1 where the
// reference skips the pair (dls.py:278-279).  All in-map offsets are 32-bit (maps are <= 65535 x 65535, checked in
// vote_view), added to the view's wave-uniform base pointer: the gather is `global_load_ubyte v, v_off, s[base]`.
template <int DIV>
__device__ __forceinline__ int seg_bin_regs(const ViewRegs& vd, const ViewDesc* __restrict__ vp,
                                            const uint8_t* __restrict__ pool, double X, double Y, double Z) {
    int xi, yi;
    int bin = -1;
    if (project<DIV>(vd, X, Y, Z, xi, yi)) {
        constexpr bool kSimple = DIV == kDivFlatSimple;
        if (!kSimple && !vd.unit_scale) {
            const double xs = trunc((double)xi * vp->wscale);  // :281
            const double ys = trunc((double)yi * vp->hscale);  // :282
            const int seg_h = vp->seg_h;
            xi = xs > (double)(vd.seg_w - 1) ? vd.seg_w - 1 : (int)xs;  // :285 (xs >= 0 always)
            yi = ys > (double)(seg_h - 1) ? seg_h - 1 : (int)ys;        // :286
        }
#if GSX_ABLATE & 32  // timing experiment only: the footprint of a 4x4-coarsened map (upper bound for a coarse level)
        xi >>= 2;
        yi >>= 2;
#endif
        unsigned off;
        if (kSimple || vd.seg_row_bytes) {
            // strips of 16 pixel columns, rows of a strip back to back (16 B each): any 8 consecutive rows of a strip
            // are one 128-B line, so a compact patch of pixels is a compact set of cache lines, and the offset is
            // (xi>>4) * strip_bytes + (yi<<4) + (xi&15): four integer instructions
            off = __umul24((unsigned)xi >> 4, (unsigned)vd.seg_row_bytes) + ((unsigned)xi & 15u) + ((unsigned)yi << 4);
        } else {
            off = __umul24((unsigned)yi, (unsigned)vd.seg_w) + (unsigned)xi;
        }
        // the device copy of the descriptor holds the map's absolute address; say "global" explicitly, a pointer made
        // from an integer would otherwise be a FLAT one (flat loads also count in lgkmcnt and stall the scalar waits)
        typedef const __attribute__((address_space(1))) uint8_t* global_u8;
#if GSX_ABLATE & 2  // timing experiment only (tools/ablate.sh): no gather, a bin made from the offset
        bin = (int)(off & 127u);
#elif GSX_ABLATE & 8  // timing experiment only: every gather falls into the first KiB of the map (always cached)
        bin = ((global_u8)(unsigned long long)vd.seg_off)[off & 1023u];
#elif GSX_ABLATE & 16  // timing experiment only: the first 64 KiB of the map (L2-resident, rarely in L1)
        bin = ((global_u8)(unsigned long long)vd.seg_off)[off & 65535u];
#else
        bin = ((global_u8)(unsigned long long)vd.seg_off)[off];
#endif
    }
    return bin;
}

template <int DIV>
__device__ __forceinline__ int seg_bin(const ViewDesc* __restrict__ vp, const uint8_t* __restrict__ pool, double X, double Y,
                                       double Z) {
    const ViewRegs vd = load_view(vp);
    return seg_bin_regs<DIV>(vd, vp, pool, X, Y, Z);
}

template <int DIV>
__global__ __launch_bounds__(kBlock) void project_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                         const float* __restrict__ z, long long n,
                                                         const ViewDesc* __restrict__ vd,
                                                         const uint32_t* __restrict__ perm, int* __restrict__ ox,
                                                         int* __restrict__ oy, float filt_h) {
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    const bool valid = i < n;  // (no early return: the flat forms ballot over the whole wave)
    int xi, yi;
    const ViewRegs vr = load_view(vd);
    const double X = valid ? (double)x[i] : __builtin_nan(""), Y = valid ? (double)y[i] : 0.0, Z = valid ? (double)z[i] : 0.0;
    const bool vis = DIV == kDivFiltCoarse ? project_filtered(vr, filt_h, X, Y, Z, xi, yi) : project<DIV>(vr, X, Y, Z, xi, yi);
    if (!valid) return;
    const long long o = perm ? (long long)perm[i] : i;  // back to the caller's order
    ox[o] = vis ? xi : -1;
    oy[o] = vis ? yi : -1;
}

// Workgroups b and b+8 run on the same XCD (round-robin dispatch; a speed heuristic, never needed for
// correctness).  swizzle == 1: XCD x takes the x-th contiguous eighth of the Morton curve.  swizzle == C >= 2: the
// curve is cut into chunks of C workgroups and chunk k goes to XCD k % 8: an XCD still works on compact pieces
// of space (its L2 serves compact patches of every segmentation map), but every XCD sees every part of the scene, so
// the eight of them finish together even when visibility differs from one region to the next.  Bijective for any grid.
__device__ __forceinline__ unsigned logical_block(unsigned b, unsigned nwg, int swizzle) {
    if (!swizzle) return b;
    if (swizzle == 1) {
        const unsigned q = nwg >> 3, r = nwg & 7u, xcd = b & 7u;
        return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
    }
    const unsigned C = (unsigned)swizzle, super = 8u * C;
    if (b >= nwg / super * super) return b;  // the ragged tail keeps its order
    const unsigned j = b >> 3;
    return (j / C) * super + (b & 7u) * C + j % C;
}

// -------------------------------------------------------------------------------------------------
// seg-map packing on the device (gsx_vote_view_device / gsx_vote_views_device): ONE pass over the int32 / int64 /
// u8 map writes both levels of the library's form — the u8 bins (label + 1) in strips of 16 pixel columns and the
// 4x4-coarsened level (a cell = the bin its 16 pixels share, 255 where they differ or the cell sticks out of the
// map) — and validates the label range.  One thread = one 4x4 cell: four 16-byte loads (int32), four u32 stores
// (the four lanes of a strip write 64 contiguous bytes) and one coarse byte.  Up to kPackBatch maps of one geometry
// per launch (blockIdx.y): a 1080p map is 8 MB, i.e. ~2 us of HBM time, less than a launch.
// -------------------------------------------------------------------------------------------------
static constexpr int kPackBatch = 16;
struct PackArgs {
    const void* src[kPackBatch];
    uint8_t* dst[kPackBatch];
    int view[kPackBatch];  // index reported through `err` when the map holds a label outside [-1, bins-2]
    int w, h, strip_bytes, cstrip_bytes, cw, ch, bins;
    unsigned coarse_off;
    int* err;              // atomicMin of the offending view indices (INT_MAX = none)
};

template <typename T, int ADD, bool VEC>
__global__ __launch_bounds__(kBlock) void seg_pack_fused_kernel(PackArgs a) {
    const long long q = (long long)blockIdx.x * kBlock + threadIdx.x;
    const int cy = (int)(q / a.cw), cx = (int)(q % a.cw);
    if (cy >= a.ch) return;
    const int job = blockIdx.y;
    const T* __restrict__ in = static_cast<const T*>(a.src[job]);
    uint8_t* __restrict__ out = a.dst[job];
    const int x0 = cx * 4, y0 = cy * 4;
    const unsigned bins = (unsigned)a.bins;
    unsigned bad = 0;
    uint32_t rows[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int y = y0 + r;
        uint32_t packed = 0;
        if (y < a.h) {
            const T* row = in + (long long)y * a.w + x0;
            T v[4];
            bool have[4];
            if (VEC) {  // host checked: w % 4 == 0 and a 16-byte aligned base, so x0 + 4 <= w and the vector is aligned
                typedef T vec4 __attribute__((ext_vector_type(4)));
                const vec4 t = *reinterpret_cast<const vec4*>(row);
                v[0] = t.x, v[1] = t.y, v[2] = t.z, v[3] = t.w;
                have[0] = have[1] = have[2] = have[3] = true;
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    have[k] = x0 + k < a.w;
                    v[k] = have[k] ? row[k] : (T)0;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned long long b = (unsigned long long)(long long)v[k] + (unsigned)ADD;  // label -2 wraps far above bins
                bad |= (unsigned)(have[k] & (b >= bins));
                packed |= (have[k] ? (uint32_t)(b & 0xffu) : 0u) << (8 * k);  // columns past the edge: bin 0
            }
            if (a.strip_bytes) {
                *reinterpret_cast<uint32_t*>(out + (long long)(x0 >> 4) * a.strip_bytes + ((long long)y << 4) + (x0 & 15)) = packed;
            } else {
                const long long o = (long long)y * a.w + x0;
                if (VEC) *reinterpret_cast<uint32_t*>(out + o) = packed;
                else
                    for (int k = 0; k < 4 && x0 + k < a.w; ++k) out[o + k] = (uint8_t)(packed >> (8 * k));
            }
        }
        rows[r] = packed;
    }
    if (a.cstrip_bytes) {
        const uint32_t same = (rows[0] & 0xffu) * 0x01010101u;
        const bool uniform = x0 + 4 <= a.w && y0 + 4 <= a.h && rows[0] == same && rows[1] == same && rows[2] == same && rows[3] == same;
        out[a.coarse_off + (long long)(cx >> 4) * a.cstrip_bytes + (cx & 15) + ((long long)cy << 4)] = uniform ? (uint8_t)(rows[0] & 0xffu) : (uint8_t)255;
    }
    if (bad) atomicMin(a.err, a.view[job]);
}

// -------------------------------------------------------------------------------------------------
// Host maps arrive in the COMPACT form of host_pack.hpp (coarse level + one 16-byte block per mixed 4x4 cell: the PCIe
// link, not the host pass, is what a run's hand-over waits for, and a segmentation map is mostly uniform cells).  This
// kernel, queued behind the DMA of a group of maps on a GPU that has nothing else to do while maps are handed over,
// rebuilds the pool form: one workgroup = one band of 8 pixel rows (two cell rows) of one map; a thread takes the cells of a
// cell row in the stream's order (by cell column), a ballot scan ranks the mixed ones, and every cell is written
// as four 4-byte rows - its block, or its coarse byte four times over.  The coarse level is copied as it is; every
// byte of the map's pool stride is written (cells and rows past the map, alignment gaps: 0), so a pool map stays a pure
// function of the map, which exchange protocol v4 ships.
// -------------------------------------------------------------------------------------------------
struct ExpandArgs {
    const uint8_t* rec[kCompactBatch];  // compact records in the device staging buffer
    uint8_t* map[kCompactBatch];        // their places in the pool
    int w, h, strip_bytes, cstrip_bytes, cw, ch, strips;
    unsigned table_bytes, stream_off, coarse_off, fine_bytes, map_bytes, stride;
    unsigned max_block;  // last block a cell row of this geometry can hold, counted from its first
};

__global__ __launch_bounds__(kBlock) void seg_expand_kernel(ExpandArgs a) {
    __shared__ unsigned wave_total[kBlock / 64];
    const uint8_t* __restrict__ rec = a.rec[blockIdx.y];
    uint8_t* __restrict__ map = a.map[blockIdx.y];
    const int band = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint8_t* __restrict__ coarse = rec + a.table_bytes;
    const uint32_t* __restrict__ table = reinterpret_cast<const uint32_t*>(rec);
    const uint4* __restrict__ stream = reinterpret_cast<const uint4*>(rec + a.stream_off);
    const int per = a.strips * 4;  // cells per cell row (padded to whole strips)
    for (int cyl = 0; cyl < 2; ++cyl) {  // the band's two cell rows: each has its own run of blocks in the stream
        const int cy = band * 2 + cyl;
        const unsigned first = table[2 * band + cyl];
        unsigned running = 0;  // mixed cells of this cell row before the current round
        for (int i0 = 0; i0 < per; i0 += kBlock) {
            const int cx = i0 + tid;
            const bool valid = cx < per;
            const int s = cx >> 2, cc = cx & 3;
            const bool cell = valid && cy < a.ch && cx < a.cw;
            const unsigned cb = cell ? coarse[(size_t)(cx >> 4) * a.cstrip_bytes + (cx & 15) + ((size_t)cy << 4)] : 0u;
            const bool mixed = cell && cb == 255u;
            const unsigned long long m = __builtin_amdgcn_ballot_w64(mixed);
            const unsigned below = (unsigned)__popcll(m & ((1ull << lane) - 1ull));
            __syncthreads();  // the previous round's totals have been read
            if (lane == 0) wave_total[wv] = (unsigned)__popcll(m);
            __syncthreads();
            unsigned before = running;
#pragma unroll
            for (int k = 0; k < kBlock / 64; ++k) {
                before += k < wv ? wave_total[k] : 0u;
                running += wave_total[k];
            }
            if (valid) {
                uint32_t r0, r1, r2, r3;
                if (mixed) {
                    const uint4 b = stream[first + min(before + below, a.max_block)];  // (a record is the library's own: the clamp never acts)
                    r0 = b.x, r1 = b.y, r2 = b.z, r3 = b.w;
                } else {
                    r0 = r1 = r2 = r3 = cb * 0x01010101u;  // a uniform cell; 0 for the cells and rows past the map
                }
                uint8_t* o = map + (size_t)s * a.strip_bytes + ((size_t)(band * 8 + cyl * 4) << 4) + cc * 4;
                *reinterpret_cast<uint32_t*>(o) = r0;
                *reinterpret_cast<uint32_t*>(o + 16) = r1;
                *reinterpret_cast<uint32_t*>(o + 32) = r2;
                *reinterpret_cast<uint32_t*>(o + 48) = r3;
            }
        }
    }
    // the coarse level, verbatim (16-byte pieces dealt over the bands), and the alignment gaps
    const unsigned cbytes = a.map_bytes - a.coarse_off;  // a multiple of 128
    for (unsigned j = (unsigned)band * kBlock + tid; j < cbytes / 16; j += gridDim.x * kBlock)
        reinterpret_cast<uint4*>(map + a.coarse_off)[j] = reinterpret_cast<const uint4*>(coarse)[j];
    if (band == 0) {
        for (unsigned j = a.fine_bytes + tid; j < a.coarse_off; j += kBlock) map[j] = 0;
        for (unsigned j = a.map_bytes + tid; j < a.stride; j += kBlock) map[j] = 0;
    }
}

// -------------------------------------------------------------------------------------------------
// fused vote, single GPU / single batch: labels straight out of the kernel
// -------------------------------------------------------------------------------------------------
struct FusedParams {
    const float* x;
    const float* y;
    const float* z;
    long long n;
    const ViewDesc* views;  // the batch's views
    int nviews;             // <= kMaxBatch
    const uint8_t* pool;
    int bins;
    int stride_dw;         // LDS row stride in dwords (odd)
    int xcd_swizzle;       // see logical_block()
    const uint32_t* perm;  // sorted slot -> caller's index, or nullptr
    const double* cull;    // culling planes of the batch's views, [25][cull_pitch] (nullptr: no culling)
    int cull_pitch;
    unsigned long long* cull_tally;  // [0] += (wave, view) pairs skipped
    float filt_h;                    // kDivFiltCoarse: 0.5 - E of project_filtered() for the largest frame among the staged views
};

// ---- wave-level view culling ------------------------------------------------------------------------------------
// 40 % of the (Gaussian, view) pairs of an ordinary capture are invisible, and in Morton order they come in whole
// waves.  Each wave bounds its 64 Gaussians by a sphere (c, r) once; then LANE l tests view l against the five
// world-space planes the host derived from that view's frustum (gsx::cull_planes): a plane is stored as a unit normal
// A, an offset B and a margin slope M, and `A.c + B > r + M (|c|_1 + r)` proves that every point of the sphere fails the reference's visibility test (dls.py:72, :80) by more than a million times the rounding error of
// the fp64 projection.  One ballot turns 64 such tests into a 64-bit mask; a culled view costs two scalar
// instructions instead of ~50 fp64 ones, and no descriptor load.  Bits are indexed by the REVERSE view index
// (nviews-1-v), so that the U views of a chunk are U adjacent bits.  Pairs that are not provably invisible take
// the exact path as before: the result is unchanged, bit for bit.
static constexpr int kCullPlanes = 5;
static constexpr unsigned kTallySlots = 1024, kTallyStride = 16;  // u64 counters, one per 128-B line
struct CullMasks {
    unsigned long long m[4];  // kMaxBatch <= 256 views
};

__device__ __forceinline__ double wave_min(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmin(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}

// X is NaN on lanes past the end of the scene: fmin / fmax ignore them (and NaN positions, which never vote).
__device__ __forceinline__ CullMasks wave_cull_masks(const double* __restrict__ cull, int pitch, int nviews, double X,
                                                     double Y, double Z, unsigned long long* __restrict__ tally) {
    CullMasks k;
    k.m[0] = k.m[1] = k.m[2] = k.m[3] = 0;
    if (!cull) return k;  // wave-uniform
    const double big = 1.7e308;
    const double lx = wave_min(X == X ? X : big), hx = wave_max(X == X ? X : -big);
    const double ly = wave_min(X == X ? Y : big), hy = wave_max(X == X ? Y : -big);
    const double lz = wave_min(X == X ? Z : big), hz = wave_max(X == X ? Z : -big);
    const double cx = 0.5 * lx + 0.5 * hx, cy = 0.5 * ly + 0.5 * hy, cz = 0.5 * lz + 0.5 * hz;
    const double dx = X - cx, dy = Y - cy, dz = Z - cz;
    // r >= the distance of every member from c; the factor covers the rounding of the distance itself.
    // An infinite coordinate makes c or r non-finite and every test below false: such a wave is never culled.
    const double r = wave_max(sqrt(dx * dx + dy * dy + dz * dz)) * (1.0 + 1e-12);
    const double reach = fabs(cx) + fabs(cy) + fabs(cz) + r;  // >= |p| for every member: scales the rounding margin
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        if (64 * j < nviews) {  // wave-uniform
            const int ri = 64 * j + lane;  // reverse view index
            bool out = false;
            if (ri < nviews) {
                const double* q = cull + (nviews - 1 - ri);
#pragma unroll
                for (int pl = 0; pl < kCullPlanes; ++pl) {
                    const double a0 = q[(5 * pl + 0) * pitch], a1 = q[(5 * pl + 1) * pitch], a2 = q[(5 * pl + 2) * pitch];
                    const double b = q[(5 * pl + 3) * pitch], m = q[(5 * pl + 4) * pitch];
                    out = out | (a0 * cx + a1 * cy + a2 * cz + b > r + m * reach);
                }
            }
            k.m[j] = __builtin_amdgcn_ballot_w64(out);
        }
    }
    if (tally && lane == 0) {  // statistics for gsx_vote_culled(): one atomic per wave, spread over kTallySlots cache lines -
                               // 47 k atomics on ONE address keep an L2 channel busy for ~0.4 ms, longer than a short kernel runs
        const unsigned slot = (blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) & (kTallySlots - 1);
        atomicAdd(tally + slot * kTallyStride, (unsigned long long)(__popcll(k.m[0]) + __popcll(k.m[1]) + __popcll(k.m[2]) + __popcll(k.m[3])));
    }
    return k;
}

// the U bits of the chunk that starts `done` views into the (reverse) walk; 64 % U == 0, so they share a word
template <int U>
__device__ __forceinline__ unsigned cull_bits(const CullMasks& k, int done) {
    const int w = done >> 6;
    const unsigned long long word = w == 0 ? k.m[0] : w == 1 ? k.m[1] : w == 2 ? k.m[2] : k.m[3];
    return (unsigned)(word >> (done & 63)) & ((1u << U) - 1u);
}

#if GSX_ABLATE & 64
// TIMING EXPERIMENT ONLY (results invalid): what an all-fp32 filter would cost - the camera-space point from twelve floats
// (one s_load_dwordx16 instead of 176 bytes of fp64 descriptor), the exact path (full descriptor, fp64) only for the waves
// with a lane near a pixel boundary.  Without the per-wave base point such a filter would need, its error bound is missing.
struct ViewF32 {
    float r0[3], r1[3], r2[3], t0, t1, t2, hw, hh;
    int cam_w, cam_h, coarse_row_bytes;
    unsigned coarse_delta;
    long long seg_off;
};
__device__ __forceinline__ ViewF32 load_view_f32(const ViewDesc* __restrict__ vp) {
    const char* q = reinterpret_cast<const char*>(vp);
    v16i a = *reinterpret_cast<const v16i*>(q + 192);
    v8i d = *reinterpret_cast<const v8i*>(q + 160);
    v2i so = *reinterpret_cast<const v2i*>(q + 112);
    asm volatile("" : "+s"(a), "+s"(d), "+s"(so));
    ViewF32 r;
    int w[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) w[k] = a[k];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        r.r0[k] = __builtin_bit_cast(float, w[4 + k]);
        r.r1[k] = __builtin_bit_cast(float, w[7 + k]);
        r.r2[k] = __builtin_bit_cast(float, w[10 + k]);
    }
    r.t0 = __builtin_bit_cast(float, w[13]);
    r.t1 = __builtin_bit_cast(float, w[14]);
    r.t2 = __builtin_bit_cast(float, w[15]);
    const int d2 = d[2], d3 = d[3], d4 = d[4], d5 = d[5], d6 = d[6], d7 = d[7];
    r.cam_w = d2, r.cam_h = d3, r.coarse_row_bytes = d4, r.coarse_delta = (unsigned)d5;
    r.hw = __builtin_bit_cast(float, d6);
    r.hh = __builtin_bit_cast(float, d7);
    r.seg_off = __builtin_bit_cast(long long, so);
    return r;
}
#endif

// One chunk of U views (vb-1, vb-2, ..): bin[u] = the vote of this lane's Gaussian in view vb-1-u, or -1.
// FULL: all U views exist (no index test).  culled: bit u set = the whole wave provably misses view vb-1-u.
typedef const __attribute__((address_space(1))) uint8_t* global_u8_ptr;

template <int U, int DIV, bool FULL>
__device__ __forceinline__ void gather_chunk(const ViewDesc* __restrict__ views, const uint8_t* __restrict__ pool, int vb,
                                             double X, double Y, double Z, unsigned culled, int (&bin)[U], float filt_h) {
    if (DIV == kDivFlatCoarse || DIV == kDivFiltCoarse) {
        // Two-level lookup.  Every map carries a 4x4-coarsened copy whose cell holds the label shared by its 16
        // pixels, or 255 where they differ.  The patch of pixels a wave gathers is ~30 px wide and costs ~16 L2
        // requests per view in the full-resolution map (one per 128-B line = 16x8 pixels; the L2 request rate, not
        // HBM, is what the kernel waits for) but ~3 in the coarse one (a line = 64x32 pixels).  Phase A gathers
        // the coarse cells of all U views; phase B re-reads the exact pixel only for lanes that hit a mixed cell
        // (segment boundaries).  A uniform cell IS the pixel's label, so results do not change.
        unsigned pixel[U];  // xi | yi << 16 (both < 65536): the full-resolution offset is only worked out if needed
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int v = vb - 1 - u;
            bin[u] = -1;  // (pixel[u] is only read where bin[u] == 255, i.e. where it was set below)
            if (((culled >> u) & 1u) || !(FULL || v >= 0)) continue;  // wave-uniform
#if GSX_ABLATE & 64  // timing experiment only, see load_view_f32
            if (DIV == kDivFiltCoarse) {
                const ViewF32 f = load_view_f32(views + v);
                const float Xf = (float)X, Yf = (float)Y, Zf = (float)Z;
                const float zf = __builtin_fmaf(Xf, f.r2[0], __builtin_fmaf(Yf, f.r2[1], __builtin_fmaf(Zf, f.r2[2], f.t2)));
                const bool pos = zf > 0.f;
                if (__builtin_amdgcn_ballot_w64(pos) == 0) continue;
                const float a0 = __builtin_fmaf(Xf, f.r0[0], __builtin_fmaf(Yf, f.r0[1], __builtin_fmaf(Zf, f.r0[2], f.t0)));
                const float a1 = __builtin_fmaf(Xf, f.r1[0], __builtin_fmaf(Yf, f.r1[1], __builtin_fmaf(Zf, f.r1[2], f.t1)));
                const float r = __builtin_amdgcn_rcpf(zf);
                const float pxf = __builtin_fmaf(a0, r, f.hw), pyf = __builtin_fmaf(a1, r, f.hh);
                const float dx = __builtin_amdgcn_fractf(pxf) - 0.5f, dy = __builtin_amdgcn_fractf(pyf) - 0.5f;
                const bool cert = (zf >= 1e-12f) & (__builtin_fabsf(dx) <= filt_h) & (__builtin_fabsf(dy) <= filt_h);
                int xi = cvt_floor_i32(pxf), yi = cvt_floor_i32(pyf);
                bool vis = cert & ((unsigned)xi < (unsigned)f.cam_w) & ((unsigned)yi < (unsigned)f.cam_h);
                long long seg_off = f.seg_off;
                unsigned cdelta = f.coarse_delta;
                int crb = f.coarse_row_bytes;
                if (__builtin_amdgcn_ballot_w64(pos & !cert) != 0) {  // the exact path for the whole wave, full descriptor
                    const ViewRegs vd = load_view(views + v);
                    vis = project<kDivFlatSimple>(vd, X, Y, Z, xi, yi);
                }
                if (vis) {
                    pixel[u] = (unsigned)xi | ((unsigned)yi << 16);
                    const unsigned cx = (unsigned)xi >> 2, cy = (unsigned)yi >> 2;
                    const unsigned coff = __umul24(cx >> 4, (unsigned)crb) + (cx & 15u) + (cy << 4);
                    bin[u] = ((global_u8_ptr)((unsigned long long)seg_off + cdelta))[coff];
                }
                continue;
            }
#endif
            const ViewRegs vd = load_view(views + v);
            int xi, yi;
            if (DIV == kDivFiltCoarse ? project_filtered(vd, filt_h, X, Y, Z, xi, yi) : project<kDivFlatSimple>(vd, X, Y, Z, xi, yi)) {
                pixel[u] = (unsigned)xi | ((unsigned)yi << 16);
                const unsigned cx = (unsigned)xi >> 2, cy = (unsigned)yi >> 2;
                const unsigned coff = __umul24(cx >> 4, (unsigned)vd.coarse_row_bytes) + (cx & 15u) + (cy << 4);
                bin[u] = ((global_u8_ptr)((unsigned long long)vd.seg_off + vd.coarse_delta))[coff];
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool mixed = bin[u] == 255;
            if (__builtin_amdgcn_ballot_w64(mixed) != 0) {  // wave-uniform; rare away from segment boundaries
                const ViewDesc* __restrict__ vp = views + (vb - 1 - u);
                const global_u8_ptr base = (global_u8_ptr)(unsigned long long)vp->seg_off;
                const unsigned xi = pixel[u] & 0xffffu, yi = pixel[u] >> 16;
                if (mixed) bin[u] = base[__umul24(xi >> 4, (unsigned)vp->seg_row_bytes) + (xi & 15u) + (yi << 4)];
            }
        }
        return;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int v = vb - 1 - u;  // `v >= 0` is wave-uniform
        if ((culled >> u) & 1u) {  // wave-uniform
            bin[u] = -1;
            continue;
        }
#if GSX_ABLATE & 4  // timing experiment only: ONE descriptor for all views (no scalar loads in the loop; NOTE: every
                    // view then votes the same pixel, so the gathers become free as well); the view-dependent
                    // perturbation (it rounds away) keeps the arithmetic inside the loop
        const ViewRegs vd0 = load_view(views);
        bin[u] = (FULL || v >= 0) ? seg_bin_regs<DIV>(vd0, views, pool, __builtin_fma((double)v, 1e-300, X), Y, Z) : -1;
#else
        bin[u] = (FULL || v >= 0) ? seg_bin<DIV>(views + v, pool, X, Y, Z) : -1;
#endif
    }
}

// (The last stage's LDS layout - a wave's histograms as [bin][64 lanes] bytes - was measured here too, round 3: 1.052 ms against
// 1.036-1.043 for the row per thread, profiles/r03/lds_layout_and_replay_ab.txt; removed again.)
template <int U, int DIV, bool LDS_BATCH>
__global__ __launch_bounds__(kBlock) void vote_fused_labels_kernel(FusedParams p, const ViewDesc* __restrict__ views,
                                                                   int* __restrict__ labels) {
    extern __shared__ uint32_t lds[];
    uint32_t* row = lds + threadIdx.x * p.stride_dw;
    uint8_t* h = reinterpret_cast<uint8_t*>(row);

    for (int k = 0; k < p.stride_dw; ++k) row[k] = 0;  // thread-private: no barrier needed
    const long long i = (long long)logical_block(blockIdx.x, gridDim.x, p.xcd_swizzle) * kBlock + threadIdx.x;
    const bool valid = i < p.n;
    // lanes past the end carry NaN: every comparison in project() fails, they never vote
    const double X = valid ? (double)p.x[i] : __builtin_nan("");
    const double Y = valid ? (double)p.y[i] : 0.0;
    const double Z = valid ? (double)p.z[i] : 0.0;
    const uint8_t* __restrict__ pool = p.pool;

    int best = -1;  // bin of the current winner
    int bestc = 0;
    const CullMasks cmask = wave_cull_masks(p.cull, p.cull_pitch, p.nviews, X, Y, Z, p.cull_tally);
    // views vb-1, vb-2, .. (reverse order); only the last, ragged chunk tests its view indices
    auto chunk = [&](auto full, int vb) {
        constexpr bool kFull = decltype(full)::value;
        int bin[U];
        gather_chunk<U, DIV, kFull>(views, pool, vb, X, Y, Z, cull_bits<U>(cmask, p.nviews - vb), bin, p.filt_h);
#if GSX_ABLATE & 1  // timing experiment only: no LDS histogram, the bins are just consumed
#pragma unroll
        for (int u = 0; u < U; ++u) best += bin[u];
        if (false)
#endif
        if (LDS_BATCH) {
            // one LDS round trip for the whole chunk: read the U counters first, resolve repeats of a bin
            // inside the chunk in registers, then apply the votes in (reverse view) order
            int old[U];
#pragma unroll
            for (int u = 0; u < U; ++u) old[u] = bin[u] >= 0 ? (int)h[bin[u]] : 0;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (bin[u] >= 0) {
                    int c = old[u] + 1;  // dls.py:295
#pragma unroll
                    for (int w = 0; w < u; ++w) c += (bin[w] == bin[u]) ? 1 : 0;
                    h[bin[u]] = (uint8_t)c;  // LDS stores of one wave retire in order: the last repeat wins
                    if (c >= bestc) {  // reverse-order tie rule == first-inserted wins (dls.py:303)
                        bestc = c;
                        best = bin[u];
                    }
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (bin[u] >= 0) {
                    const int c = h[bin[u]] + 1;  // dls.py:295
                    h[bin[u]] = (uint8_t)c;
                    if (c >= bestc) {  // reverse-order tie rule == first-inserted wins (dls.py:303)
                        bestc = c;
                        best = bin[u];
                    }
                }
            }
        }
        };
    int vb = p.nviews;
    for (; vb >= U; vb -= U) chunk(std::true_type{}, vb);
    if (vb > 0) chunk(std::false_type{}, vb);
    if (valid) {
        const long long o = p.perm ? (long long)p.perm[i] : i;
        labels[o] = best - 1 + (best < 0);  // bin b -> label b-1; no vote -> -1 (dls.py:306)
    }
}

// -------------------------------------------------------------------------------------------------
// early vote, last stage (host side: early_vote_stage / vote_finalize).  While the host is still handing over the last
// maps of a run, vote_fused_planes_kernel has already voted the views [0, E) on a second stream and left their
// count plane cntA and first-view plane fvA (u8, wave-major [wave][bin][64]; code = 255 - view index, 0 = no vote).  This kernel is
// what remains between the last map and the labels: the walk of vote_fused_labels_kernel over the views [E, V) only,
// on a histogram that STARTS from cntA, and one pass over fvA.
// The reference's winner (dls.py:303) is the bin with the largest total whose first vote is earliest.  A bin with
// votes in [0, E) has its first vote there (fvA); among such bins the pass below maximises (total, fvA).  The reverse
// walk's online rule maximises (total, earliest vote inside [E, V)) over the bins voted in [E, V); if that bin has
// no vote in [0, E) it beats every other bin of its kind, and it wins only with a strictly larger total than the best
// bin that was voted earlier.  If it has votes in [0, E) it is covered by the pass.
// The planes are read as dwords by a transposed lane mapping (lane (g, t) = bins g, g+4, .. of Gaussians 4t .. 4t+3,
// as the planes kernel stores them): 256 B per wave instruction instead of 64 single bytes.
// -------------------------------------------------------------------------------------------------
// ---- early vote by RECORD and REPLAY (option "early_replay") -----------------------------------------------------------------
// The early stage only RECORDS the vote of every Gaussian in every early view (bin + 1, 0 = none; [wave][view][64 lanes]
// bytes): projection, gather, store - no histogram, no LDS, twice the waves per CU.  The last stage is the one-piece kernel
// with one difference: behind its reverse walk over the views [E, V) it does not project the views E-1 .. 0, it reads their
// votes back from the record, in the same reverse order.  Every update of the histogram and of (best, best_count) is the
// one the one-piece kernel would have made: the labels are identical by construction.
template <int U, int DIV>
__global__ __launch_bounds__(kBlock) void vote_record_kernel(FusedParams p, const ViewDesc* __restrict__ views,
                                                             uint8_t* __restrict__ rec) {
    const long long i = (long long)logical_block(blockIdx.x, gridDim.x, p.xcd_swizzle) * kBlock + threadIdx.x;
    const bool valid = i < p.n;
    const long long ic = valid ? i : 0;
    const float xf = p.x[ic], yf = p.y[ic], zf = p.z[ic];
    const double X = valid ? (double)xf : __builtin_nan("");  // lanes past the end never vote
    const double Y = valid ? (double)yf : 0.0;
    const double Z = valid ? (double)zf : 0.0;
    const uint8_t* __restrict__ pool = p.pool;
    const int lane = threadIdx.x & 63;
    uint8_t* r = rec + (i - lane) * p.nviews + lane;
    const CullMasks cmask = wave_cull_masks(p.cull, p.cull_pitch, p.nviews, X, Y, Z, p.cull_tally);
    auto chunk = [&](auto full, int vb) {
        constexpr bool kFull = decltype(full)::value;
        int bin[U];
        gather_chunk<U, DIV, kFull>(views, pool, vb, X, Y, Z, cull_bits<U>(cmask, p.nviews - vb), bin, p.filt_h);
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (kFull || vb - 1 - u >= 0) r[(vb - 1 - u) * 64] = (uint8_t)(bin[u] + 1);
    };
    int vb = p.nviews;
    for (; vb >= U; vb -= U) chunk(std::true_type{}, vb);
    if (vb > 0) chunk(std::false_type{}, vb);
}

template <int U, int DIV>
__global__ __launch_bounds__(kBlock) void vote_fused_replay_kernel(FusedParams p, const ViewDesc* __restrict__ views,
                                                                   const uint8_t* __restrict__ rec, int E,
                                                                   int* __restrict__ labels) {
    extern __shared__ uint32_t lds[];
    uint32_t* row = lds + threadIdx.x * p.stride_dw;
    uint8_t* h = reinterpret_cast<uint8_t*>(row);
    for (int k = 0; k < p.stride_dw; ++k) row[k] = 0;  // thread-private: no barrier needed
    const long long i = (long long)logical_block(blockIdx.x, gridDim.x, p.xcd_swizzle) * kBlock + threadIdx.x;
    const bool valid = i < p.n;
    const long long ic = valid ? i : 0;
    const float xf = p.x[ic], yf = p.y[ic], zf = p.z[ic];
    const double X = valid ? (double)xf : __builtin_nan("");
    const double Y = valid ? (double)yf : 0.0;
    const double Z = valid ? (double)zf : 0.0;
    const uint8_t* __restrict__ pool = p.pool;
    int best = -1, bestc = 0;
    auto vote = [&](int b) {  // dls.py:295, :303 in reverse view order, as in vote_fused_labels_kernel
        const int c = h[b] + 1;
        h[b] = (uint8_t)c;
        if (c >= bestc) {
            bestc = c;
            best = b;
        }
    };
    const CullMasks cmask = wave_cull_masks(p.cull, p.cull_pitch, p.nviews, X, Y, Z, p.cull_tally);
    auto chunk = [&](auto full, int vb) {
        constexpr bool kFull = decltype(full)::value;
        int bin[U];
        gather_chunk<U, DIV, kFull>(views, pool, vb, X, Y, Z, cull_bits<U>(cmask, p.nviews - vb), bin, p.filt_h);
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (bin[u] >= 0) vote(bin[u]);
    };
    int vb = p.nviews;
    for (; vb >= U; vb -= U) chunk(std::true_type{}, vb);
    if (vb > 0) chunk(std::false_type{}, vb);
    // the views E-1 .. 0 from the record, kReplay bytes of it in flight per lane
    constexpr int kReplay = 16;
    const int lane = threadIdx.x & 63;
    const uint8_t* __restrict__ r = rec + (i - lane) * E + lane;
    for (int v0 = E; v0 > 0; v0 -= kReplay) {
        unsigned b[kReplay];
#pragma unroll
        for (int j = 0; j < kReplay; ++j) {
            const int v = v0 - 1 - j;
            const unsigned got = r[max(v, 0) * 64];  // unconditional load + select
            b[j] = v >= 0 ? got : 0u;
        }
        // (Applying the replayed votes eight per LDS round trip - counters read first, repeats of a bin resolved in registers -
        // was measured, round 3: the last stage took 0.495 ms instead of 0.426; the 28 compares per batch cost more than the
        // dependent read-modify-writes they replace.  Removed, like lds_batch in the walk.)
#pragma unroll
        for (int j = 0; j < kReplay; ++j)
            if (b[j]) vote((int)b[j] - 1);
    }
    if (valid) {
        const long long o = p.perm ? (long long)p.perm[i] : i;
        labels[o] = best - 1 + (best < 0);  // bin b -> label b-1; no vote -> -1 (dls.py:306)
    }
}

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pk_max_u16(uint32_t a, uint32_t b) {  // v_pk_max_u16
    return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(u16x2, a), __builtin_bit_cast(u16x2, b)));
}

template <int U, int DIV, int NQ>
__global__ __launch_bounds__(kBlock, 4) void vote_fused_final_kernel(FusedParams p, const ViewDesc* __restrict__ views,
                                                                     const uint8_t* __restrict__ cntA,
                                                                     const uint8_t* __restrict__ fvA,
                                                                     const uint8_t* __restrict__ recA, int E,
                                                                     int* __restrict__ labels, int ablate_arg) {
#ifdef GSX_EXPERIMENTS
    const int ablate = ablate_arg;  // timing experiments (tools/early_probe.py), results invalid
#else
    constexpr int ablate = 0;       // product build: every branch on it folds away
#endif
    extern __shared__ uint32_t lds[];
    const long long i = (long long)logical_block(blockIdx.x, gridDim.x, p.xcd_swizzle) * kBlock + threadIdx.x;
    const bool valid = i < p.n;
    const long long ic = valid ? i : 0;  // unconditional loads: all three in flight together with the planes' rows
    const float xf = p.x[ic], yf = p.y[ic], zf = p.z[ic];
    const double X = valid ? (double)xf : __builtin_nan("");  // lanes past the end never vote
    const double Y = valid ? (double)yf : 0.0;
    const double Z = valid ? (double)zf : 0.0;
    const uint8_t* __restrict__ pool = p.pool;
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4;
    // The wave's histogram block in LDS has the layout of its block in the planes, [bin][64 lanes] bytes (rows padded to
    // a multiple of four bins): the counts of the views [0, E) arrive by a plain coalesced copy, and dword q * 64 + lane
    // of either holds bin 4q + g of the Gaussians 4t .. 4t+3 (g = lane / 16, t = lane % 16) - what the pass over the
    // first-view plane below needs side by side.  Bank behaviour of the walk is that of the row-per-thread layout: lanes
    // voting one bin touch 16 consecutive dwords, lanes voting different bins collide now and then.
    const int nq = (p.bins + 3) >> 2;
    uint32_t* wl = lds + (threadIdx.x >> 6) * (nq * 64);  // nq * 256 bytes per wave
    uint8_t* hw = reinterpret_cast<uint8_t*>(wl) + lane;   // hw[bin * 64]: this lane's counter of a bin
    const long long wave_at = (ablate & 1) ? 0 : (i - lane) * p.bins;  // (ablate: timing experiments only, tools/early_probe.py)
    const uint32_t* __restrict__ csrc = reinterpret_cast<const uint32_t*>(cntA + wave_at) + lane;
    const uint32_t* __restrict__ fsrc = reinterpret_cast<const uint32_t*>(fvA + wave_at) + lane;
    // unconditional load + select (a conditional load costs a branch per row; the rows past the last bin lie in the next
    // wave's block or in the slack behind the planes, kEarlySlack)
    auto plane_dword = [&](const uint32_t* __restrict__ src, int q) {
        const uint32_t v = src[q * 64];
        return 4 * q + g < p.bins ? v : 0u;
    };
    // A wave lives through a chain of memory round trips and only 16 waves fit a CU (the histograms): the fewer round
    // trips, the shorter the kernel.  NQ > 0 (bins <= 4 NQ): the first-view rows are fetched right behind the counts and
    // wait in NQ registers until the walk is over; NQ == 0: any bin count, planes in rounds of kRound rows.
    constexpr int kRound = 13;
    uint32_t fd[NQ > 0 ? NQ : 1];
    if (NQ > 0) {
        uint32_t w[NQ > 0 ? NQ : 1];
#pragma unroll
        for (int q = 0; q < NQ; ++q) w[q] = plane_dword(csrc, q);
#pragma unroll
        for (int q = 0; q < NQ; ++q) fd[q] = plane_dword(fsrc, q);  // right behind the counts: ONE round trip for both planes
        __builtin_amdgcn_sched_barrier(0);                          // (keep all 2 NQ loads ahead of the first LDS write)
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            if (q < nq) wl[q * 64 + lane] = w[q];
    } else {
        for (int q0 = 0; q0 < nq; q0 += kRound) {
            uint32_t w[kRound];
#pragma unroll
            for (int j = 0; j < kRound; ++j) w[j] = plane_dword(csrc, q0 + j);
#pragma unroll
            for (int j = 0; j < kRound; ++j)
                if (q0 + j < nq) wl[(q0 + j) * 64 + lane] = w[j];
        }
    }
    __builtin_amdgcn_wave_barrier();  // the block is written and read by the lanes of ONE wave: LDS keeps a wave's order

    int best = -1, bestc = 0;
    const CullMasks cmask = wave_cull_masks((ablate & 32) ? nullptr : p.cull, p.cull_pitch, p.nviews, X, Y, Z, p.cull_tally);
    auto chunk = [&](auto full, int vb) {
        constexpr bool kFull = decltype(full)::value;
        int bin[U];
        gather_chunk<U, DIV, kFull>(views, pool, vb, X, Y, Z, cull_bits<U>(cmask, p.nviews - vb), bin, p.filt_h);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (bin[u] >= 0) {
                const int c = hw[bin[u] * 64] + 1;  // dls.py:295
                hw[bin[u] * 64] = (uint8_t)c;
                if (c >= bestc) {  // reverse order: the bin whose earliest vote in [E, V) comes first survives a tie
                    bestc = c;
                    best = bin[u];
                }
            }
        }
    };
    int vb = p.nviews;
    for (; vb >= U; vb -= U) chunk(std::true_type{}, vb);
    if (vb > 0) chunk(std::false_type{}, vb);
    __builtin_amdgcn_wave_barrier();

    // Largest 16-bit key total << 8 | first-view code over ALL bins, per Gaussian: two packed maxima per lane for its four
    // Gaussians, v_perm_b32 pairs a dword of totals with a dword of codes.  A bin without a vote in [0, E) has code 0 and
    // loses against an equal total with one: the order the tie rule asks for.
    uint32_t m01 = 0u, m23 = 0u;
    auto fold = [&](uint32_t f4, int q) {
        const uint32_t hd = wl[q * 64 + lane];
        m01 = pk_max_u16(m01, __builtin_amdgcn_perm(hd, f4, 0x05010400u));  // [T1 f1 T0 f0]
        m23 = pk_max_u16(m23, __builtin_amdgcn_perm(hd, f4, 0x07030602u));  // [T3 f3 T2 f2]
    };
    if (ablate & 4) {  // timing experiment: no pass over the first-view rows (they are still fetched)
#pragma unroll
        for (int q = 0; q < (NQ > 0 ? NQ : 1); ++q) m01 |= fd[q] & 1u;
    } else if (NQ > 0) {
#pragma unroll
        for (int q = 0; q < NQ; ++q)
            if (q < nq) fold(fd[q], q);
    } else {
        for (int q0 = 0; q0 < nq; q0 += kRound) {
            uint32_t f[kRound];
#pragma unroll
            for (int j = 0; j < kRound; ++j) f[j] = plane_dword(fsrc, q0 + j);
#pragma unroll
            for (int j = 0; j < kRound; ++j)
                if (q0 + j < nq) fold(f[j], q0 + j);
        }
    }
#pragma unroll
    for (int o = 16; o <= 32; o <<= 1) {  // the four lanes (0..3, t) share their Gaussians
        m01 = pk_max_u16(m01, (uint32_t)__shfl_xor((int)m01, o));
        m23 = pk_max_u16(m23, (uint32_t)__shfl_xor((int)m23, o));
    }
    // Gaussian `lane` = 4 t' + k' is key k' of the lanes with t = t'
    const uint32_t s01 = (uint32_t)__shfl((int)m01, lane >> 2), s23 = (uint32_t)__shfl((int)m23, lane >> 2);
    const uint32_t pair = (lane & 2) ? s23 : s01;
    const unsigned mine = (lane & 1) ? pair >> 16 : pair & 0xffffu;
    if (valid) {
        // code 0: no bin was voted in [0, E) and reached the total of the walk's winner -> the walk's winner (-1: no vote at
        // all).  Otherwise the winner is the bin this Gaussian voted in view 255 - code: the early stage kept that record.
        int win = best;
        const unsigned code = mine & 0xffu;
        if (code && !(ablate & 8)) win = (int)recA[(i - lane) * E + (255 - (int)code) * 64 + lane] - 1;
        const long long o = (p.perm && !(ablate & 16)) ? (long long)p.perm[i] : i;
        labels[o] = win - 1 + (win < 0);  // bin b -> label b-1; no vote -> -1 (dls.py:306)
    }
}

// -------------------------------------------------------------------------------------------------
// exchange protocol v3, rank-local part 1: the same walk as the labels kernel (u8 counters, 16 waves/CU)
// but the only output is this rank's COUNT plane, u8 [slab][bins][sn].  No first-view plane: ties are
// resolved later, for the tied Gaussians only, by vote_tie_kernel.
// -------------------------------------------------------------------------------------------------
template <int U, int DIV>
__global__ __launch_bounds__(kBlock) void vote_fused_counts_kernel(FusedParams p, const ViewDesc* __restrict__ views,
                                                                   uint8_t* __restrict__ cnt, long long sn) {
    extern __shared__ uint32_t lds[];
    uint32_t* row = lds + threadIdx.x * p.stride_dw;
    for (int k = 0; k < p.stride_dw; ++k) row[k] = 0;
    uint8_t* h = reinterpret_cast<uint8_t*>(row);
    const long long i = (long long)logical_block(blockIdx.x, gridDim.x, p.xcd_swizzle) * kBlock + threadIdx.x;
    const bool valid = i < p.n;
    const double X = valid ? (double)p.x[i] : __builtin_nan("");
    const double Y = valid ? (double)p.y[i] : 0.0;
    const double Z = valid ? (double)p.z[i] : 0.0;
    const uint8_t* __restrict__ pool = p.pool;
    const CullMasks cmask = wave_cull_masks(p.cull, p.cull_pitch, p.nviews, X, Y, Z, p.cull_tally);
    // views vb-1, vb-2, .. (reverse order); only the last, ragged chunk tests its view indices
    auto chunk = [&](auto full, int vb) {
        constexpr bool kFull = decltype(full)::value;
        int bin[U];
        gather_chunk<U, DIV, kFull>(views, pool, vb, X, Y, Z, cull_bits<U>(cmask, p.nviews - vb), bin, p.filt_h);
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (bin[u] >= 0) h[bin[u]] = (uint8_t)(h[bin[u]] + 1);
        };
    int vb = p.nviews;
    for (; vb >= U; vb -= U) chunk(std::true_type{}, vb);
    if (vb > 0) chunk(std::false_type{}, vb);
    // transposed store (see vote_fused_planes_kernel): lane (g, t) packs bin 4*it+g of Gaussians 4t..4t+3
    const int lane = threadIdx.x & 63;
    const int g = lane >> 4, t = lane & 15;
    const long long i0 = i - lane;
    const long long slab = i0 / sn;
    const long long base = slab * p.bins * sn + (i0 - slab * sn) + 4 * t;
    const uint32_t* wrow = lds + (threadIdx.x - lane + 4 * t) * p.stride_dw;
    for (int b = g; b < p.bins; b += 4) {
        uint32_t pc = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) pc |= (uint32_t)reinterpret_cast<const uint8_t*>(wrow + k * p.stride_dw)[b] << (8 * k);
        *reinterpret_cast<uint32_t*>(cnt + base + (long long)b * sn) = pc;
    }
}

// v3, slab owner: sum the S ranks' counters per bin; a unique maximum is the label, otherwise the Gaussian is
// TIED (label -2 for now) and the set of max-count bins goes out as a bit mask, cand[word][sn], 8 words.
static constexpr int kCandWords = 8;  // bins <= 256
// One thread owns 4 consecutive Gaussians: every plane row is read as coalesced dwords (256 B per wave
// instruction instead of 64 single bytes).  (Two passes - maximum, then masks - read the slab twice: 0.37 ms;
// parking the totals in LDS instead was 2x slower still, 77 KB per wave leaves 2 waves per CU.)
__device__ __forceinline__ void slab_totals4(const uint8_t* __restrict__ rcnt, int S, int bins, long long sn, long long i4, int b,
                                             unsigned t[4]) {
    t[0] = t[1] = t[2] = t[3] = 0;
    for (int r = 0; r < S; ++r) {
        const uint32_t w = *reinterpret_cast<const uint32_t*>(rcnt + ((long long)r * bins + b) * sn + i4);
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] += (w >> (8 * k)) & 0xffu;
    }
}

__global__ __launch_bounds__(kBlock) void vote_slab_totals_kernel(const uint8_t* __restrict__ rcnt, int S, int bins,
                                                                  long long sn, int* __restrict__ slab_labels,
                                                                  uint32_t* __restrict__ cand) {
    const long long i4 = ((long long)blockIdx.x * kBlock + threadIdx.x) * 4;
    if (i4 >= sn) return;
    // ONE pass over the planes: the mask of max-count bins is built against the running maximum; a word of 32 bins
    // remembers the maximum its bits refer to, and only the words whose maximum is the final one survive.
    unsigned M[4] = {0, 0, 0, 0};
    uint32_t word[kCandWords][4];
    unsigned wmax[kCandWords][4];
#pragma unroll
    for (int w = 0; w < kCandWords; ++w) {
#pragma unroll
        for (int k = 0; k < 4; ++k) word[w][k] = 0, wmax[w][k] = 0;
        if (w * 32 < bins) {  // wave-uniform
            const int nb = min(32, bins - w * 32);
#pragma unroll 4
            for (int bb = 0; bb < nb; ++bb) {
                unsigned t[4];
                slab_totals4(rcnt, S, bins, sn, i4, w * 32 + bb, t);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (t[k] > M[k]) {  // a new maximum: the bits gathered so far in this word are void
                        M[k] = t[k];
                        word[w][k] = 0;
                    }
                    if (t[k] == M[k]) word[w][k] |= 1u << bb;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) wmax[w][k] = M[k];
        }
    }
    int n_max[4] = {0, 0, 0, 0}, first_bin[4] = {-1, -1, -1, -1};
#pragma unroll
    for (int w = 0; w < kCandWords; ++w) {
        uint32_t out[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            out[k] = (M[k] > 0 && wmax[w][k] == M[k]) ? word[w][k] : 0u;
            if (out[k] && n_max[k] == 0) first_bin[k] = w * 32 + __ffs(out[k]) - 1;
            n_max[k] += __popc(out[k]);
        }
        *reinterpret_cast<uint4*>(cand + (long long)w * sn + i4) = make_uint4(out[0], out[1], out[2], out[3]);
    }
    int lab[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) lab[k] = n_max[k] == 0 ? -1 : (n_max[k] == 1 ? first_bin[k] - 1 : -2);
    *reinterpret_cast<int4*>(slab_labels + i4) = make_int4(lab[0], lab[1], lab[2], lab[3]);
}

// (A wave-ballot compaction of the tied Gaussians in front of this kernel was measured: 0.90 ms vs 0.59 ms —
// the atomically ordered list loses the Morton locality of the gathers; not kept.)
// v3, every rank: for each TIED Gaussian (two or more candidate bins) walk this rank's views in FORWARD order
// and stop at the first one that votes a candidate: code = (255 - local view index) << 8 | bin, 0 if none.
// cand_all: [S][kCandWords][sn] (all-gathered masks); codes: u16 [S][sn], slab-major.
// earlier: codes of n_earlier batches that come BEFORE this one in view order and were walked before it (one GPU playing the
// ranks one after the other): a Gaussian one of them has resolved is not walked again - the earliest batch wins anyway.
__global__ __launch_bounds__(kBlock) void vote_tie_kernel(FusedParams p, const ViewDesc* __restrict__ views,
                                                          const uint32_t* __restrict__ cand_all, long long sn,
                                                          uint16_t* __restrict__ codes, const uint16_t* __restrict__ earlier,
                                                          int n_earlier) {
    const long long i = (long long)logical_block(blockIdx.x, gridDim.x, p.xcd_swizzle) * kBlock + threadIdx.x;
    if (i >= p.n) return;
    for (int e = 0; e < n_earlier; ++e)
        if (earlier[(long long)e * sn + i]) {  // (single slab: codes are [batch][sn])
            codes[i] = 0;
            return;
        }
    const long long slab = i / sn, j = i - slab * sn;
    uint32_t mask[kCandWords];
    int pop = 0;
#pragma unroll
    for (int w = 0; w < kCandWords; ++w) {
        mask[w] = cand_all[(slab * kCandWords + w) * sn + j];
        pop += __popc(mask[w]);
    }
    unsigned code = 0;
    if (pop >= 2) {
        const double X = (double)p.x[i], Y = (double)p.y[i], Z = (double)p.z[i];
        for (int v = 0; v < p.nviews; ++v) {
            const int sb = seg_bin<kDivFlat>(views + v, p.pool, X, Y, Z);
            if (sb < 0) continue;
            const unsigned b = (unsigned)sb;
            uint32_t word = 0;
#pragma unroll
            for (int w = 0; w < kCandWords; ++w) word = (b >> 5) == (unsigned)w ? mask[w] : word;
            if ((word >> (b & 31u)) & 1u) {
                code = ((unsigned)(255 - v) << 8) | b;
                break;
            }
        }
    }
    codes[i] = (uint16_t)code;  // == codes[slab][j]: n_pad = S * sn
}

// v3, slab owner: a tied Gaussian takes the candidate first voted by the LOWEST rank that voted one (ranks own
// contiguous rank-ordered view blocks), i.e. the globally earliest view.  rcodes: u16 [S(src)][sn].
__global__ __launch_bounds__(kBlock) void vote_tie_resolve_kernel(const uint16_t* __restrict__ rcodes, int S, long long sn,
                                                                  int* __restrict__ slab_labels) {
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= sn || slab_labels[i] != -2) return;
    int label = -1;
    for (int r = 0; r < S; ++r) {
        const unsigned c = rcodes[(long long)r * sn + i];
        if (c) {
            label = (int)(c & 0xffu) - 1;
            break;
        }
    }
    slab_labels[i] = label;
}

// -------------------------------------------------------------------------------------------------
// fused vote, planes mode (multi-GPU exchange, or more than kMaxBatch views): the batch's votes
// are merged into the global planes cnt[bins][n_pad] / fv[bins][n_pad].
// LDS word per bin: count << 8 | local index of the earliest view that voted it.
// fv code = FVMAX - global view index of the first vote (larger = earlier; 0 = no vote).
// -------------------------------------------------------------------------------------------------
template <int U, typename PT, int DIV>
__global__ __launch_bounds__(kBlock) void vote_fused_planes_kernel(FusedParams p, const ViewDesc* __restrict__ views,
                                                                   PT* __restrict__ cnt,
                                                                   PT* __restrict__ fv, long long sn,
                                                                   int view_base, int fresh, int local_codes,
                                                                   uint8_t* __restrict__ rec) {
    constexpr int FVMAX = sizeof(PT) == 1 ? 255 : 65535;
    extern __shared__ uint32_t lds[];
    uint32_t* row = lds + threadIdx.x * p.stride_dw;
    for (int k = 0; k < p.stride_dw; ++k) row[k] = 0;
    uint16_t* h = reinterpret_cast<uint16_t*>(row);

    const long long i = (long long)logical_block(blockIdx.x, gridDim.x, p.xcd_swizzle) * kBlock + threadIdx.x;
    const bool valid = i < p.n;
    const double X = valid ? (double)p.x[i] : __builtin_nan("");
    const double Y = valid ? (double)p.y[i] : 0.0;
    const double Z = valid ? (double)p.z[i] : 0.0;
    const uint8_t* __restrict__ pool = p.pool;

    const CullMasks cmask = wave_cull_masks(p.cull, p.cull_pitch, p.nviews, X, Y, Z, p.cull_tally);
    // views vb-1, vb-2, .. (reverse order); only the last, ragged chunk tests its view indices
    auto chunk = [&](auto full, int vb) {
        constexpr bool kFull = decltype(full)::value;
        int bin[U];
        gather_chunk<U, DIV, kFull>(views, pool, vb, X, Y, Z, cull_bits<U>(cmask, p.nviews - vb), bin, p.filt_h);
        if (rec) {  // early vote: the record of every vote, [wave][view][64 lanes] bytes (bin + 1, 0 = none) - the last stage
                    // looks the winner's bin up by the view of its first vote
            uint8_t* r = rec + (i - (threadIdx.x & 63)) * p.nviews + (threadIdx.x & 63);
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (kFull || vb - 1 - u >= 0) r[(vb - 1 - u) * 64] = (uint8_t)(bin[u] + 1);
        }
        unsigned old[U];  // one LDS round trip per chunk, repeats of a bin resolved in registers
#pragma unroll
        for (int u = 0; u < U; ++u) old[u] = bin[u] >= 0 ? (unsigned)h[bin[u]] : 0u;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (bin[u] >= 0) {
                const int v = vb - 1 - u;
                unsigned cnt = (old[u] >> 8) + 1u;
#pragma unroll
                for (int w = 0; w < u; ++w) cnt += (bin[w] == bin[u]) ? 1u : 0u;
                h[bin[u]] = (uint16_t)((cnt << 8) | (unsigned)v);  // reverse order: the last store = the earliest view
            }
        }
        };
    int vb = p.nviews;
    for (; vb >= U; vb -= U) chunk(std::true_type{}, vb);
    if (vb > 0) chunk(std::false_type{}, vb);
    // planes are slab-major: [slab][bin][sn] with slab = i / sn (one slab = one rank's share in the
    // all-to-all exchange; a single slab is the plain [bin][n_pad] layout).  sn is a multiple of the block
    // size, so a workgroup never straddles two slabs.
    const int vb0 = local_codes ? 0 : view_base;  // local codes: 255 - index inside this rank's batch
    if (sizeof(PT) == 1 && fresh) {
        // Transposed store through the wave's own LDS rows: lane (g, t) = (lane / 16, lane % 16) packs bin
        // 4*it + g of Gaussians 4t .. 4t+3 into one dword per plane, so a wave writes 4 bins x 64 B per
        // instruction pair instead of 1 bin x 64 single bytes.  Rows of lanes past n hold zeros.
        const int lane = threadIdx.x & 63;
        const int g = lane >> 4, t = lane & 15;
        const long long i0 = i - lane;  // first Gaussian of this wave
        const long long slab = i0 / sn;
        const long long base = slab * p.bins * sn + (i0 - slab * sn) + 4 * t;
        const uint32_t* wrow = lds + (threadIdx.x - lane + 4 * t) * p.stride_dw;
        for (int b = g; b < p.bins; b += 4) {
            uint32_t pc = 0, pf = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const unsigned w = reinterpret_cast<const uint16_t*>(wrow + k * p.stride_dw)[b];
                const unsigned c = w >> 8;
                const unsigned code = c ? (unsigned)(FVMAX - (vb0 + (int)(w & 0xffu))) : 0u;
                pc |= c << (8 * k);
                pf |= code << (8 * k);
            }
            const long long at = base + (long long)b * sn;
            *reinterpret_cast<uint32_t*>(cnt + at) = pc;
            *reinterpret_cast<uint32_t*>(fv + at) = pf;
        }
        return;
    }
    if (!valid) return;
    const long long slab = i / sn;
    const long long base = slab * p.bins * sn + (i - slab * sn);
    for (int b = 0; b < p.bins; ++b) {
        const unsigned w = h[b];
        const unsigned c = w >> 8;
        const unsigned code = c ? (unsigned)(FVMAX - (vb0 + (int)(w & 0xffu))) : 0u;
        const long long at = base + (long long)b * sn;
        if (fresh) {
            cnt[at] = (PT)c;
            fv[at] = (PT)code;
        } else if (c) {
            cnt[at] = (PT)(cnt[at] + c);
            const unsigned old = fv[at];
            if (code > old) fv[at] = (PT)code;
        }
    }
}

// per Gaussian: among the bins holding the (global) maximum count, the one this rank saw first
template <typename PT>
__global__ __launch_bounds__(kBlock) void vote_keys_kernel(const PT* __restrict__ cnt, const PT* __restrict__ fv,
                                                           long long n, long long sn, int bins,
                                                           int* __restrict__ keys) {
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const long long slab = i / sn;
    const long long base = slab * bins * sn + (i - slab * sn);
    unsigned M = 0;
    int key = 0;
    for (int b = 0; b < bins; ++b) {
        const unsigned c = cnt[base + (long long)b * sn];
        if (c == 0 || c < M) continue;
        const unsigned f = fv[base + (long long)b * sn];
        const int k = f ? (int)((f << 8) | (unsigned)b) : 0;
        if (c > M) {
            M = c;
            key = k;
        } else if (k > key) {
            key = k;
        }
    }
    keys[i] = key;
}

// ---- exchange v2: after the all-to-all this rank holds, for ITS slab of Gaussians, every rank's u8 count and
// first-view planes: recv[r][bin][sn].  Sum the counts, and among the bins holding the maximum pick the one
// whose first vote is globally earliest: ranks own contiguous, rank-ordered view blocks, so that is the
// lowest rank that voted it, then its local view code (255 - local index; larger = earlier).
__global__ __launch_bounds__(kBlock) void vote_slab_reduce_kernel(const uint8_t* __restrict__ rcnt,
                                                                  const uint8_t* __restrict__ rfv, int S, int bins,
                                                                  long long sn, int* __restrict__ slab_labels) {
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= sn) return;
    unsigned M = 0;
    int best_bin = -1;
    unsigned best_key = 0;  // (S - r) << 8 | code : larger = earlier
    for (int b = 0; b < bins; ++b) {
        unsigned total = 0;
        unsigned key = 0;
        for (int r = 0; r < S; ++r) {
            const long long at = ((long long)r * bins + b) * sn + i;
            const unsigned c = rcnt[at];
            total += c;
            if (c && key == 0) key = ((unsigned)(S - r) << 8) | (unsigned)rfv[at];
        }
        if (total == 0 || total < M) continue;
        if (total > M || key > best_key) {
            M = total;
            best_key = key;
            best_bin = b;
        }
    }
    slab_labels[i] = best_bin - 1 + (best_bin < 0);  // bin b -> label b-1; never visible -> -1
}

__global__ __launch_bounds__(kBlock) void unpermute_labels_kernel(const int* __restrict__ sorted, long long n,
                                                                  const uint32_t* __restrict__ perm,
                                                                  int* __restrict__ labels) {
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) labels[perm ? (long long)perm[i] : i] = sorted[i];
}

__global__ __launch_bounds__(kBlock) void vote_labels_kernel(const int* __restrict__ keys, long long n,
                                                             const uint32_t* __restrict__ perm,
                                                             int* __restrict__ labels) {
    const long long i = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const int k = keys[i];
    labels[perm ? (long long)perm[i] : i] = k ? (k & 0xff) - 1 : -1;
}

// -------------------------------------------------------------------------------------------------
// host side
// -------------------------------------------------------------------------------------------------
void fill_view_desc(ViewDesc& vd, const gsx_camera* cam, int seg_w, int seg_h, int img_w, int img_h) {
    std::memset(&vd, 0, sizeof vd);
    double nR[9];
    for (int k = 0; k < 9; ++k) {
        vd.R[k] = cam->R[k];
        nR[k] = -cam->R[k];  // t = -R @ p: unary minus first (dls.py:66)
    }
    for (int r = 0; r < 3; ++r)
        vd.t[r] = std::fma(nR[3 * r + 2], cam->p[2], std::fma(nR[3 * r + 0], cam->p[0], nR[3 * r + 1] * cam->p[1]));
    vd.fx = cam->fx;
    vd.fy = cam->fy;
    vd.half_w = (double)cam->width / 2.0;
    vd.half_h = (double)cam->height / 2.0;
    vd.width = (double)cam->width;
    vd.height = (double)cam->height;
    vd.cam_w = cam->width;
    vd.cam_h = cam->height;
    vd.hw32 = (float)vd.half_w;  // exact for frames up to 2^24 pixels a side
    vd.hh32 = (float)vd.half_h;
    for (int k = 0; k < 3; ++k) {
        vd.xf[k] = (float)(vd.fx * vd.R[k]);
        vd.xf[3 + k] = (float)(vd.fy * vd.R[3 + k]);
        vd.xf[6 + k] = (float)vd.R[6 + k];
    }
    vd.xf[9] = (float)(vd.fx * vd.t[0]);
    vd.xf[10] = (float)(vd.fy * vd.t[1]);
    vd.xf[11] = (float)vd.t[2];
    vd.wscale = (double)seg_w / (double)img_w;
    vd.hscale = (double)seg_h / (double)img_h;
    vd.seg_w = seg_w;
    vd.seg_h = seg_h;
    // int(x*1.0) == x; the clamp (dls.py:285-286) is a no-op only if the camera frame fits the map
    vd.unit_scale = (vd.wscale == 1.0 && vd.hscale == 1.0 && cam->width <= seg_w && cam->height <= seg_h) ? 1 : 0;
}

// call after sync_views(): views_simple describes the staged batch
static inline int div_mode(const Ctx* c) {
    if (!c->opt_flat_project) return c->opt_fast_div ? kDivCertified : kDivExact;
    if (!c->views_simple) return kDivFlat;
    return c->views_coarse && c->opt_seg_coarse ? (c->opt_filter_project ? kDivFiltCoarse : kDivFlatCoarse) : kDivFlatSimple;
}
// 0.5 - E of project_filtered() for the staged views (E grows with the frame: the largest one decides)
static float filter_half_width(const Ctx* c) {
    int dim = 64;
    for (const ViewDesc& v : c->views) dim = std::max(dim, std::max(v.cam_w, v.cam_h));
    return (float)(0.5 - kFilterK * (double)dim * 5.9604644775390625e-08);
}
static inline unsigned grid_for(long long n) { return (unsigned)((n + kBlock - 1) / kBlock); }

int project_all(Ctx* c, const gsx_camera* cam, const float* dx, const float* dy, const float* dz, int64_t n,
                int32_t* x_host, int32_t* y_host, const uint32_t* perm) {
    if (n <= 0) return GSX_OK;
    GSX_HIP(c, hipSetDevice(c->device));
    ViewDesc vd;
    fill_view_desc(vd, cam, 1, 1, 1, 1);
    DevBuf dvd, ox, oy;
    GSX_HIP(c, dvd.ensure(sizeof vd));
    GSX_HIP(c, ox.ensure(sizeof(int) * n));
    GSX_HIP(c, oy.ensure(sizeof(int) * n));
    int rc = GSX_OK;
    hipError_t e = hipMemcpyAsync(dvd.p, &vd, sizeof vd, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        ProfScope ps(c, "project");
        // the filtered form is the vote kernels' (kDivFiltCoarse): frames up to 65535 pixels a side (hw32 exact, E < 1/64)
        const int dim = std::max(cam->width, cam->height);
        const float filt_h = (float)(0.5 - kFilterK * (double)std::max(dim, 64) * 5.9604644775390625e-08);
        if (c->opt_flat_project && c->opt_filter_project && dim <= 65535)
            hipLaunchKernelGGL(project_kernel<kDivFiltCoarse>, dim3(grid_for(n)), dim3(kBlock), 0, c->stream, dx, dy, dz, (long long)n,
                               dvd.as<ViewDesc>(), perm, ox.as<int>(), oy.as<int>(), filt_h);
        else if (c->opt_flat_project)
            hipLaunchKernelGGL(project_kernel<kDivFlat>, dim3(grid_for(n)), dim3(kBlock), 0, c->stream, dx, dy, dz, (long long)n,
                               dvd.as<ViewDesc>(), perm, ox.as<int>(), oy.as<int>(), 0.f);
        else if (c->opt_fast_div)
            hipLaunchKernelGGL(project_kernel<kDivCertified>, dim3(grid_for(n)), dim3(kBlock), 0, c->stream, dx, dy, dz, (long long)n,
                               dvd.as<ViewDesc>(), perm, ox.as<int>(), oy.as<int>(), 0.f);
        else
            hipLaunchKernelGGL(project_kernel<kDivExact>, dim3(grid_for(n)), dim3(kBlock), 0, c->stream, dx, dy, dz, (long long)n,
                               dvd.as<ViewDesc>(), perm, ox.as<int>(), oy.as<int>(), 0.f);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(x_host, ox.p, sizeof(int) * n, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(y_host, oy.p, sizeof(int) * n, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) rc = fail(c, GSX_E_HIP, "project_all: %s", hipGetErrorString(e));
    dvd.release();
    ox.release();
    oy.release();
    return rc;
}

int vote_begin(Ctx* c, int n_classes, int first_view, int total_views) {
    if (n_classes < 1 || n_classes > 255)
        return fail(c, GSX_E_UNSUPPORTED, "vote_begin: n_classes=%d outside [1,255] (u8 seg maps)", n_classes);
    if (first_view < 0 || total_views < 1 || total_views > 65535 || first_view >= total_views)
        return fail(c, GSX_E_INVALID, "vote_begin: first_view=%d total_views=%d (need 0 <= first < total <= 65535)",
                    first_view, total_views);
    GSX_HIP(c, hipSetDevice(c->device));
    c->n_classes = n_classes;
    c->bins = n_classes + 1;
    c->first_view = first_view;
    c->total_views = total_views;
    c->local_codes = c->opt_local_codes != 0;
    c->wide = total_views > 255 && !c->local_codes;
    c->slabs = c->opt_slabs > 0 ? c->opt_slabs : 1;
    c->sn = ((c->n + c->slabs - 1) / c->slabs + 255) / 256 * 256;
    if (c->sn == 0) c->sn = 256;
    c->n_pad = c->sn * c->slabs;
    c->views.clear();
    c->pend_count = 0;  // packed maps of an abandoned run that never went up: forgotten with it
    c->compact_bytes = 0;
    c->views_dirty = true;
    c->pool_base = nullptr;
    c->seg_used = 0;
    c->n_flushed = 0;
    c->planes_valid = false;
    c->planes_zero = false;
    c->planes_stale = false;
    c->labels_valid = false;
    {
        const int rcj = early_join(c);  // an abandoned run's early stage may still be running: this run reuses its buffers
        if (rcj) return rcj;
    }
    c->early_state = 0;
    c->early_done = 0;
    c->early_batches = 0;
    GSX_HIP(c, c->errflag.ensure(sizeof(int)));
    GSX_HIP(c, hipMemsetAsync(c->errflag.p, 0x7f, sizeof(int), c->stream));  // kNoBadView
    c->vote_begun = true;
    return GSX_OK;
}

static int pool_reserve(Ctx* c, size_t need) {
    if (need <= c->segpool.cap) return GSX_OK;
    {
        const int rcf = vote_flush_pending(c);  // the pool is about to move: maps still waiting in the pinned ring go up first
        if (rcf) return rcf;
    }
    if (c->early_state == 1 || c->early_batches > 0) {  // the early stage reads the pool that is about to be freed
        GSX_HIP(c, hipStreamSynchronize(c->stream2));
        c->early_inflight = false;
        c->early_state = -1;
        c->early_batches = 0;
    }
    size_t cap = c->segpool.cap ? c->segpool.cap : ((size_t)64 << 20);
    while (cap < need) cap *= 2;
    void* np = nullptr;
    GSX_HIP(c, hipMalloc(&np, cap));
    if (c->seg_used) {
        hipError_t e = hipMemcpyAsync(np, c->segpool.p, c->seg_used, hipMemcpyDeviceToDevice, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) {
            (void)hipFree(np);
            return fail(c, GSX_E_HIP, "seg pool growth copy failed: %s", hipGetErrorString(e));
        }
    }
    c->segpool.release();
    c->segpool.p = np;
    c->segpool.cap = cap;
    return GSX_OK;
}

// ---- seg-map hand-over ------------------------------------------------------------------------------------------
// Common front part of gsx_vote_view / gsx_vote_views_device: argument checks, the map's layout, room in the pool.
static int view_prologue(Ctx* c, const char* who, const void* cams, const void* seg, int n, int seg_dtype, int seg_w, int seg_h,
                         int img_w, int img_h, MapLayout& L) {
    if (!c->vote_begun) return fail(c, GSX_E_STATE, "%s before vote_begin", who);
    if (c->pool_base) return fail(c, GSX_E_STATE, "%s after gsx_vote_import: the run's views are final; start the next one with gsx_vote_begin", who);
    if (!cams || !seg) return fail(c, GSX_E_INVALID, "%s: NULL argument", who);
    if (seg_w < 1 || seg_h < 1 || img_w < 1 || img_h < 1)
        return fail(c, GSX_E_INVALID, "%s: sizes must be positive (seg %dx%d, image %dx%d)", who, seg_w, seg_h, img_w, img_h);
    if (seg_w > 65535 || seg_h > 65535)  // in-map byte offsets are 32-bit, row products use 24-bit multiplies
        return fail(c, GSX_E_RANGE, "%s: segmentation map %dx%d exceeds 65535 pixels a side", who, seg_w, seg_h);
    if (seg_dtype != GSX_SEG_I32 && seg_dtype != GSX_SEG_I64 && seg_dtype != GSX_SEG_U8 && seg_dtype != GSX_SEG_U8_LABELS)
        return fail(c, GSX_E_INVALID, "%s: unknown seg_dtype %d", who, seg_dtype);
    if (c->first_view + (int)c->views.size() + n > c->total_views)
        return fail(c, GSX_E_RANGE, "%s: more views than total_views=%d announced at vote_begin", who, c->total_views);
    if (c->local_codes && (int)c->views.size() + n > kMaxBatch)
        return fail(c, GSX_E_RANGE, "%s: the all-to-all exchange keeps 8-bit per-rank counters: at most %d views per rank", who,
                    kMaxBatch);
    GSX_HIP(c, hipSetDevice(c->device));
    // bin 255 must be free to mean "mixed" in the coarse level
    L = map_layout(seg_w, seg_h, c->opt_seg_tiled != 0, c->opt_seg_coarse && c->bins <= 255);
    if (L.map_bytes > 0xffffffffull) return fail(c, GSX_E_RANGE, "%s: segmentation map %dx%d exceeds 4 GiB", who, seg_w, seg_h);
    const size_t stride = (L.map_bytes + 255) / 256 * 256;  // 256-B aligned maps
    const size_t off = (c->seg_used + 255) / 256 * 256;
    size_t need = off + stride * (size_t)n;
    if (c->views.empty()) {
        // first view of a run: room for all announced views of this geometry at once (capped), so that the pool is
        // not re-allocated and copied while maps stream in
        const size_t all = stride * (size_t)(c->total_views - c->first_view);
        need = std::max(need, std::min(all, (size_t)32 << 30));
    }
    return pool_reserve(c, need);
}

static void push_view(Ctx* c, const gsx_camera* cam, const MapLayout& L, size_t off, int img_w, int img_h) {
    ViewDesc vd;
    fill_view_desc(vd, cam, L.w, L.h, img_w, img_h);
    vd.seg_off = (long long)off;
    vd.seg_row_bytes = L.strip_bytes;
    vd.coarse_row_bytes = L.cstrip_bytes;
    vd.coarse_delta = (unsigned)L.coarse_off;
    c->views.push_back(vd);
    c->views_dirty = true;
    c->seg_used = off + L.map_bytes;
    c->labels_valid = false;
}

// near: a buffer the pool is about to read (the first map of a run): the workers are placed on its NUMA node
static void launch_pack(Ctx* c, const PackArgs& a, int jobs, int seg_dtype, bool vec, long long cells);

static Workers* host_workers(Ctx* c, const void* near = nullptr) {
    if (!c->workers) {
        try {
            c->workers = new Workers(c->opt_host_threads > 0 ? c->opt_host_threads : default_host_threads(), numa_node_of(near));
        } catch (...) {  // out of memory (a thread that cannot be started is handled inside: the pool then has one thread)
            c->workers = nullptr;
        }
    }
    return c->workers;  // nullptr: single-threaded packing on the calling thread
}

// Host map: the worker threads narrow it into the next slot of the pinned ring (u8 strips + coarse level, range
// checked on the way: an out-of-range label fails THIS call), one asynchronous DMA moves the packed map into the
// pool.  No kernel, no synchronisation; the call returns as soon as the caller's buffer has been read.
static int early_vote_stage(Ctx* c);

static int vote_view_host(Ctx* c, const gsx_camera* cam, const void* seg, int seg_dtype, int seg_w, int seg_h, int img_w, int img_h) {
    MapLayout L;
    int rc = view_prologue(c, "vote_view", cam, seg, 1, seg_dtype, seg_w, seg_h, img_w, img_h, L);
    if (rc) return rc;
    if (c->opt_host_pack) {
        const size_t stride = (L.map_bytes + 255) / 256 * 256;
        constexpr int kSlots = kDmaBatch * kDmaGroups;
        // compact transfer form (host_pack.hpp) whenever the pool form has both levels; else the pool form itself
        const bool compact = c->opt_host_compact && L.strip_bytes && L.cstrip_bytes;
        const CompactLayout CL = compact ? compact_layout(L) : CompactLayout{};
        const size_t slot_bytes = compact ? (CL.capacity + 255) / 256 * 256 : stride;
        if (c->hring_slot != slot_bytes || !c->hring || c->hring_compact != compact) {  // first map, or a new geometry: re-lay the ring
            if ((rc = vote_flush_pending(c))) return rc;
            for (int g = 0; g < kDmaGroups; ++g) {
                if (c->hring_busy[g]) GSX_HIP(c, hipEventSynchronize(c->hring_ev[g]));
                c->hring_busy[g] = false;
                if (!c->hring_ev[g]) GSX_HIP(c, hipEventCreateWithFlags(&c->hring_ev[g], hipEventDisableTiming));
            }
            if (c->hring_bytes < slot_bytes * kSlots) {
                if (c->hring) GSX_HIP(c, hipHostFree(c->hring));
                c->hring = nullptr;
                c->hring_bytes = 0;
                GSX_HIP(c, hipHostMalloc(&c->hring, slot_bytes * kSlots, hipHostMallocDefault));
                c->hring_bytes = slot_bytes * kSlots;
            }
            if (!compact) std::memset(c->hring, 0, slot_bytes * kSlots);  // the bytes between a map's end and its stride travel too
            c->hring_slot = slot_bytes;
            c->hring_compact = compact;
            c->hring_next = 0;
            c->cgrp = 0;
            c->grp_recs = 0;
        }
        const size_t off = (c->seg_used + 255) / 256 * 256;
        if (compact) {
            // records of a group lie back to back (a record is as long as its map has mixed cells): one DMA per group moves
            // exactly the bytes in use, and a group takes as many records as fit (at most kCompactBatch)
            const size_t gcap = (size_t)kDmaBatch * slot_bytes;
            if (c->grp_recs == 0) {
                if (c->hring_busy[c->cgrp]) {  // the group's previous DMA must have left the buffer
                    GSX_HIP(c, hipEventSynchronize(c->hring_ev[c->cgrp]));
                    c->hring_busy[c->cgrp] = false;
                }
                c->grp_used = 0;
            }
            if (c->h_scratch_cap < L.fine_bytes) {
                std::free(c->h_scratch);
                c->h_scratch = nullptr;
                c->h_scratch_cap = 0;
                const size_t cap = (L.fine_bytes + 4095) / 4096 * 4096;
                if (!(c->h_scratch = std::aligned_alloc(4096, cap))) return fail(c, GSX_E_HIP, "vote_view: out of host memory");
                c->h_scratch_cap = cap;
            }
            uint8_t* rec = static_cast<uint8_t*>(c->hring) + (size_t)c->cgrp * gcap + c->grp_used;
            size_t blocks = 0;
            if (!(c->opt_ablate & 2) &&  // timing experiment only: the DMA stream without the host pass; results invalid
                host_pack_map_compact(host_workers(c, seg), seg, seg_dtype, L, c->bins, static_cast<uint8_t*>(c->h_scratch), rec, &blocks))
                return fail(c, GSX_E_RANGE, "vote_view: segmentation map holds a label outside [-1, %d]", c->n_classes - 1);
            if (c->pend_count && (L.w != c->pend_L.w || L.h != c->pend_L.h)) {  // one expansion launch = one geometry
                if ((rc = vote_flush_pending(c))) return rc;
            }
            if (!c->pend_count) {
                c->pend_group = c->cgrp;
                c->pend_lo = c->grp_used;
                c->pend_L = L;
            }
            c->pend_rec[c->pend_count] = c->grp_used;
            c->pend_map[c->pend_count] = off;
            ++c->pend_count;
            ++c->grp_recs;
            c->grp_used += (CL.stream_off + blocks * 16 + 255) / 256 * 256;
            c->compact_bytes += CL.stream_off + blocks * 16;
            push_view(c, cam, L, off, img_w, img_h);
            const bool full = c->grp_recs == kCompactBatch || c->grp_used + slot_bytes > gcap;  // no room for a worst-case record
            // the last views of the run (as announced to vote_begin) go up one by one: a group waiting for more maps would put
            // its DMA between the last hand-over and the vote
            const bool tail = c->total_views - c->first_view - (int)c->views.size() < kDmaBatch;
            if (tail || full) rc = vote_flush_pending(c);
            if (full) {
                c->cgrp = (c->cgrp + 1) % kDmaGroups;
                c->grp_recs = 0;
            }
            return rc;
        }
        const int s = c->hring_next, g = s / kDmaBatch;
        if (s % kDmaBatch == 0 && c->hring_busy[g]) {  // the group's previous DMA must have left the buffer
            GSX_HIP(c, hipEventSynchronize(c->hring_ev[g]));
            c->hring_busy[g] = false;
        }
        uint8_t* slot = static_cast<uint8_t*>(c->hring) + (size_t)s * stride;
        if (!(c->opt_ablate & 2) && host_pack_map(host_workers(c, seg), seg, seg_dtype, L, c->bins, slot))
            return fail(c, GSX_E_RANGE, "vote_view: segmentation map holds a label outside [-1, %d]", c->n_classes - 1);
        if (c->pend_count && off != c->pend_dst + (size_t)c->pend_count * stride) {  // the pool moved on (device maps in between)
            if ((rc = vote_flush_pending(c))) return rc;
        }
        if (!c->pend_count) {
            c->pend_first = s;
            c->pend_dst = off;
        }
        ++c->pend_count;
        c->compact_bytes += stride;
        push_view(c, cam, L, off, img_w, img_h);
        c->hring_next = (s + 1) % kSlots;
        // the last views of the run (as announced to vote_begin) go up one by one: a whole group waiting for its fourth map
        // would put its 4-map DMA between the last hand-over and the vote
        const bool tail = c->total_views - c->first_view - (int)c->views.size() < kDmaBatch;
        if (tail || c->pend_count == kDmaBatch || c->hring_next % kDmaBatch == 0) return vote_flush_pending(c);
        return GSX_OK;
    }
    PinSlot& slot = c->ring[c->ring_next];
    if (slot.busy) {
        GSX_HIP(c, hipEventSynchronize(slot.ev));
        slot.busy = false;
    }
    if (slot.cap < L.map_bytes) {
        if (slot.p) GSX_HIP(c, hipHostFree(slot.p));
        slot.p = nullptr;
        slot.cap = 0;
        const size_t cap = (L.map_bytes + ((size_t)1 << 20) - 1) >> 20 << 20;
        GSX_HIP(c, hipHostMalloc(&slot.p, cap, hipHostMallocDefault));
        slot.cap = cap;
    }
    if (!slot.ev) GSX_HIP(c, hipEventCreateWithFlags(&slot.ev, hipEventDisableTiming));
    {
        // The alternative this library does NOT default to (kept for hosts short of cores, and as the A/B of DESIGN.md §3):
        // the raw map crosses PCIe (8.3 MB instead of 2.2 MB per 1080p int32 map) and the fused kernel packs it on the GPU.
        // The range check is then the device-side one (reported with the labels).
        const size_t esz = seg_dtype == GSX_SEG_I32 ? 4 : seg_dtype == GSX_SEG_I64 ? 8 : 1;
        const size_t raw = esz * (size_t)seg_w * (size_t)seg_h;
        if (slot.cap < raw) {
            GSX_HIP(c, hipHostFree(slot.p));
            slot.p = nullptr;
            slot.cap = 0;
            GSX_HIP(c, hipHostMalloc(&slot.p, raw, hipHostMallocDefault));
            slot.cap = raw;
        }
        DevBuf& land = c->dstage[c->ring_next];
        GSX_HIP(c, land.ensure(raw));
        host_copy(host_workers(c, seg), slot.p, seg, raw);
        GSX_HIP(c, hipMemcpyAsync(land.p, slot.p, raw, hipMemcpyHostToDevice, c->stream));
        const size_t off = (c->seg_used + 255) / 256 * 256;
        PackArgs a{};
        a.src[0] = land.p;
        a.dst[0] = c->segpool.as<uint8_t>() + off;
        a.view[0] = c->first_view + (int)c->views.size();
        a.w = L.w, a.h = L.h, a.strip_bytes = L.strip_bytes, a.cstrip_bytes = L.cstrip_bytes, a.cw = L.cw, a.ch = L.ch;
        a.bins = c->bins;
        a.coarse_off = (unsigned)L.coarse_off;
        a.err = c->errflag.as<int>();
        launch_pack(c, a, 1, seg_dtype, seg_w % 4 == 0, (long long)L.cw * L.ch);
        GSX_HIP(c, hipGetLastError());
        GSX_HIP(c, hipEventRecord(slot.ev, c->stream));  // after the kernel: it is the last reader of this slot's landing zone
        slot.busy = true;
        c->ring_next = (c->ring_next + 1) % kPinSlots;
        push_view(c, cam, L, off, img_w, img_h);
        return GSX_OK;
    }
    return GSX_OK;  // unreachable: both hand-over paths return above
}

int vote_view(Ctx* c, const gsx_camera* cam, const void* seg, int seg_dtype, int seg_w, int seg_h, int img_w, int img_h) {
    const int rc = vote_view_host(c, cam, seg, seg_dtype, seg_w, seg_h, img_w, img_h);
    return rc ? rc : early_vote_stage(c);  // enough of the run staged: its first views are voted while the rest is handed over
}

static void launch_pack(Ctx* c, const PackArgs& a, int jobs, int seg_dtype, bool vec, long long cells) {
    using K = void (*)(PackArgs);
    static const K table[4][2] = {{seg_pack_fused_kernel<int32_t, 1, false>, seg_pack_fused_kernel<int32_t, 1, true>},
                                  {seg_pack_fused_kernel<int64_t, 1, false>, seg_pack_fused_kernel<int64_t, 1, true>},
                                  {seg_pack_fused_kernel<uint8_t, 0, false>, seg_pack_fused_kernel<uint8_t, 0, true>},
                                  {seg_pack_fused_kernel<uint8_t, 1, false>, seg_pack_fused_kernel<uint8_t, 1, true>}};
    ProfScope ps(c, "seg_pack");
    hipLaunchKernelGGL(table[seg_dtype][vec ? 1 : 0], dim3(grid_for(cells), jobs), dim3(kBlock), 0, c->stream, a);
}

// Device maps (all of one geometry and dtype): kPackBatch maps per launch of the fused pack kernel.  Nothing is
// read back: a label out of range is recorded on the device and reported by the call that hands out the labels.
int vote_views_device(Ctx* c, int n, const gsx_camera* cams, const void* const* segs, int seg_dtype, int seg_w, int seg_h,
                      int img_w, int img_h) {
    if (n <= 0) return n == 0 ? GSX_OK : fail(c, GSX_E_INVALID, "vote_views_device: n < 0");
    MapLayout L;
    int rc = view_prologue(c, "vote_views_device", cams, segs, n, seg_dtype, seg_w, seg_h, img_w, img_h, L);
    if (rc) return rc;
    for (int i = 0; i < n; ++i)
        if (!segs[i]) return fail(c, GSX_E_INVALID, "vote_views_device: map %d is NULL", i);
    const size_t esz = seg_dtype == GSX_SEG_I32 ? 4 : seg_dtype == GSX_SEG_I64 ? 8 : 1;
    const long long cells = (long long)L.cw * L.ch;
    for (int i0 = 0; i0 < n; i0 += kPackBatch) {
        const int m = std::min(kPackBatch, n - i0);
        PackArgs a{};
        bool vec = seg_w % 4 == 0;
        size_t off = 0;
        for (int k = 0; k < m; ++k) {
            off = (c->seg_used + 255) / 256 * 256;
            a.src[k] = segs[i0 + k];
            a.dst[k] = c->segpool.as<uint8_t>() + off;
            a.view[k] = c->first_view + (int)c->views.size();
            vec = vec && reinterpret_cast<uintptr_t>(segs[i0 + k]) % (4 * esz) == 0;
            push_view(c, cams + i0 + k, L, off, img_w, img_h);
        }
        a.w = L.w, a.h = L.h, a.strip_bytes = L.strip_bytes, a.cstrip_bytes = L.cstrip_bytes, a.cw = L.cw, a.ch = L.ch;
        a.bins = c->bins;
        a.coarse_off = (unsigned)L.coarse_off;
        a.err = c->errflag.as<int>();
        launch_pack(c, a, m, seg_dtype, vec, cells);
        GSX_HIP(c, hipGetLastError());
    }
    return GSX_OK;
}

// One DMA for the packed maps that wait in the filling group of the pinned ring (up to kDmaBatch consecutive slots =
// consecutive pool offsets).  Called when the group is full, for each of the last maps of a run, and by everything that
// needs the maps on the device.
int vote_flush_pending(Ctx* c) {
    if (!c->pend_count) return GSX_OK;
    GSX_HIP(c, hipSetDevice(c->device));
    if (c->hring_compact) {
        const int g = c->pend_group;
        const size_t gcap = (size_t)kDmaBatch * c->hring_slot;
        GSX_HIP(c, c->cstage.ensure(gcap));
        const uint8_t* src = static_cast<uint8_t*>(c->hring) + (size_t)g * gcap;
        if (!(c->opt_ablate & 1))  // timing experiment only (tools/tail_probe.py): the host pass without its DMA; results invalid
            GSX_HIP(c, hipMemcpyAsync(c->cstage.as<uint8_t>() + c->pend_lo, src + c->pend_lo, c->grp_used - c->pend_lo, hipMemcpyHostToDevice,
                                      c->stream));
        GSX_HIP(c, hipEventRecord(c->hring_ev[g], c->stream));
        c->hring_busy[g] = true;
        const MapLayout& L = c->pend_L;
        const CompactLayout CL = compact_layout(L);
        ExpandArgs a{};
        for (int k = 0; k < c->pend_count; ++k) {
            a.rec[k] = c->cstage.as<uint8_t>() + c->pend_rec[k];
            a.map[k] = c->segpool.as<uint8_t>() + c->pend_map[k];
        }
        a.w = L.w, a.h = L.h, a.strip_bytes = L.strip_bytes, a.cstrip_bytes = L.cstrip_bytes, a.cw = L.cw, a.ch = L.ch;
        a.strips = (L.w + 15) / 16;
        a.table_bytes = (unsigned)CL.table_bytes, a.stream_off = (unsigned)CL.stream_off, a.coarse_off = (unsigned)L.coarse_off;
        a.fine_bytes = (unsigned)L.fine_bytes, a.map_bytes = (unsigned)L.map_bytes, a.stride = (unsigned)((L.map_bytes + 255) / 256 * 256);
        a.max_block = (unsigned)(L.cw - 1);
        if (!(c->opt_ablate & 3)) {  // (an ablation of the hand-over leaves no valid record to expand)
            ProfScope ps(c, "seg_expand");
            hipLaunchKernelGGL(seg_expand_kernel, dim3((unsigned)CL.bands, (unsigned)c->pend_count), dim3(kBlock), 0, c->stream, a);
            GSX_HIP(c, hipGetLastError());
        }
        c->pend_count = 0;
        return GSX_OK;
    }
    const int g = c->pend_first / kDmaBatch;
    const size_t bytes = (size_t)c->pend_count * c->hring_slot;
    if (!(c->opt_ablate & 1))
        GSX_HIP(c, hipMemcpyAsync(c->segpool.as<uint8_t>() + c->pend_dst, static_cast<uint8_t*>(c->hring) + (size_t)c->pend_first * c->hring_slot,
                                  bytes, hipMemcpyHostToDevice, c->stream));
    GSX_HIP(c, hipEventRecord(c->hring_ev[g], c->stream));
    c->hring_busy[g] = true;
    c->pend_count = 0;
    // a group that was flushed before it was full keeps filling: its later slots are not in flight, and the event (recorded
    // again by the next flush) always stands for the group's LAST copy, which is what the next lap waits for
    return GSX_OK;
}

int host_threads(Ctx* c) {
    Workers* w = host_workers(c);
    return w ? w->threads() : 1;
}

void vote_release_host(Ctx* c) {
    for (PinSlot& s : c->ring) {
        if (s.ev) (void)hipEventDestroy(s.ev);
        if (s.p) (void)hipHostFree(s.p);
        s = PinSlot{};
    }
    for (DevBuf& b : c->dstage) b.release();
    for (hipEvent_t& e : c->hring_ev) {
        if (e) (void)hipEventDestroy(e);
        e = nullptr;
    }
    if (c->hring) (void)hipHostFree(c->hring);
    c->hring = nullptr;
    c->hring_bytes = c->hring_slot = 0;
    std::free(c->h_scratch);
    c->h_scratch = nullptr;
    c->h_scratch_cap = 0;
    c->cstage.release();
    c->pend_count = 0;
    for (hipEvent_t& e : c->h_ev) {
        if (e) (void)hipEventDestroy(e);
        e = nullptr;
    }
    if (c->h_labels) (void)hipHostFree(c->h_labels);
    c->h_labels = nullptr;
    c->h_labels_cap = 0;
    if (c->h_views_ev) (void)hipEventDestroy(c->h_views_ev);
    c->h_views_ev = nullptr;
    if (c->h_views) (void)hipHostFree(c->h_views);
    c->h_views = nullptr;
    c->h_views_cap = 0;
    delete c->workers;
    c->workers = nullptr;
    if (c->stream2) {
        (void)hipStreamSynchronize(c->stream2);
        (void)hipStreamDestroy(c->stream2);
        c->stream2 = nullptr;
    }
    for (hipEvent_t* e : {&c->early_maps_ev, &c->early_done_ev, &c->early_up_ev}) {
        if (*e) (void)hipEventDestroy(*e);
        *e = nullptr;
    }
    if (c->h_early) (void)hipHostFree(c->h_early);
    c->h_early = nullptr;
    c->h_early_cap = 0;
    for (DevBuf* b : {&c->ecnt, &c->efv, &c->erec, &c->e_views, &c->e_cull}) b->release();
    c->early_state = -1;
}

// test hook, host only: the packed form of one map exactly as gsx_vote_view stages it
int debug_host_pack(const void* seg, int seg_dtype, int w, int h, int n_classes, int tiled, int coarse, int threads,
                    uint8_t* out, int64_t out_cap, int64_t* bytes, int64_t* coarse_off, int32_t* bad) {
    if (!seg || w < 1 || h < 1 || w > 65535 || h > 65535 || n_classes < 1 || n_classes > 255 || seg_dtype < 0 || seg_dtype > 3)
        return fail(nullptr, GSX_E_INVALID, "debug_host_pack: bad arguments");
    const MapLayout L = map_layout(w, h, tiled != 0, coarse != 0 && n_classes + 1 <= 255);
    if (bytes) *bytes = (int64_t)L.map_bytes;
    if (coarse_off) *coarse_off = L.cstrip_bytes ? (int64_t)L.coarse_off : -1;
    if (!out) return GSX_OK;
    if (out_cap < (int64_t)L.map_bytes) return fail(nullptr, GSX_E_INVALID, "debug_host_pack: output buffer too small");
    Workers pool(threads > 0 ? threads : 1);
    const int b = host_pack_map(&pool, seg, seg_dtype, L, n_classes + 1, out);
    if (bad) *bad = b;
    return GSX_OK;
}

// test hook, host only: the compact transfer form of one map (host_pack.hpp), as gsx_vote_view writes it into the pinned ring
int debug_host_pack_compact(const void* seg, int seg_dtype, int w, int h, int n_classes, int threads, uint8_t* out, int64_t out_cap,
                            int64_t* bytes, int64_t* table_bytes, int64_t* stream_off, int32_t* bad) {
    if (!seg || w < 1 || h < 1 || w > 65535 || h > 65535 || n_classes < 1 || n_classes > 254 || seg_dtype < 0 || seg_dtype > 3 || !bytes)
        return fail(nullptr, GSX_E_INVALID, "debug_host_pack_compact: bad arguments");
    const MapLayout L = map_layout(w, h, true, true);
    const CompactLayout CL = compact_layout(L);
    if (table_bytes) *table_bytes = (int64_t)CL.table_bytes;
    if (stream_off) *stream_off = (int64_t)CL.stream_off;
    *bytes = (int64_t)CL.capacity;
    if (!out) return GSX_OK;
    if (out_cap < (int64_t)CL.capacity) return fail(nullptr, GSX_E_INVALID, "debug_host_pack_compact: output buffer too small");
    void* scratch = std::aligned_alloc(4096, (L.fine_bytes + 4095) / 4096 * 4096);
    if (!scratch) return fail(nullptr, GSX_E_HIP, "debug_host_pack_compact: out of memory");
    Workers pool(threads > 0 ? threads : 1);
    size_t blocks = 0;
    const int b = host_pack_map_compact(&pool, seg, seg_dtype, L, n_classes + 1, static_cast<uint8_t*>(scratch), out, &blocks);
    std::free(scratch);
    if (bad) *bad = b;
    *bytes = (int64_t)(CL.stream_off + blocks * 16);
    return GSX_OK;
}

// Five world-space planes per view for wave_cull_masks().  With pc = R p + t (camera space) the reference rejects a
// Gaussian when pc_z <= 0 (dls.py:72) or when px = fx pc_x / pc_z + w/2 leaves [0, w) (:80), i.e. for pc_z > 0 when
// h(pc) >= 0 for one of
//      h = -pc_z,   fx pc_x - (w/2) pc_z,   -fx pc_x - (w/2) pc_z,   fy pc_y - (h/2) pc_z,   -fy pc_y - (h/2) pc_z .
// Each h = alpha . pc = (R^T alpha) . p + alpha . t is linear in the world position p; over a sphere (c, r) its minimum
// is a.c + b - r |a|.  The floating-point projection of a point p can deviate from real arithmetic by at most a few
// ulp of |alpha|_1 (|R|_F |p| + |t|), so the test demands h >= delta with delta = 1e-9 times that quantity (a million
// times the worst case) for the largest |p| of the sphere: stored are A = a/|a|, B = (b - 1e-9 |alpha|_1 |t|)/|a|
// and M = 1e-9 |alpha|_1 |R|_F / |a|, and the device tests  A.c + B > r + M (|c|_1 + r).
// Views with non-finite or extreme parameters (outside 1e-30 .. 1e30) get planes that never fire.
static constexpr int kCullStride = 5;  // doubles per plane
static void cull_planes(const ViewDesc& v, double out[kCullStride * kCullPlanes]) {
    for (int k = 0; k < kCullPlanes; ++k) {  // "never": 0.c - 1 > r + 0 is false for every r >= 0
        for (int j = 0; j < kCullStride; ++j) out[kCullStride * k + j] = 0.0;
        out[kCullStride * k + 3] = -1.0;
    }
    double rf = 0.0, tn = 0.0;
    for (int k = 0; k < 9; ++k) rf += v.R[k] * v.R[k];
    for (int k = 0; k < 3; ++k) tn += v.t[k] * v.t[k];
    rf = std::sqrt(rf);
    tn = std::sqrt(tn);
    auto sane = [](double x, double lo) { return std::isfinite(x) && std::fabs(x) >= lo && std::fabs(x) <= 1e30; };
    if (!(sane(rf, 1e-10) && sane(tn, 0.0) && sane(v.fx, 1e-30) && sane(v.fy, 1e-30) && sane(v.half_w, 1e-30) &&
          sane(v.half_h, 1e-30) && rf <= 1e10))
        return;
    const double alpha[kCullPlanes][3] = {{0.0, 0.0, -1.0},
                                          {v.fx, 0.0, -v.half_w},
                                          {-v.fx, 0.0, -v.half_w},
                                          {0.0, v.fy, -v.half_h},
                                          {0.0, -v.fy, -v.half_h}};
    for (int k = 0; k < kCullPlanes; ++k) {
        const double* al = alpha[k];
        double a[3], b = 0.0, l1 = 0.0;
        for (int j = 0; j < 3; ++j) {
            a[j] = v.R[0 + j] * al[0] + v.R[3 + j] * al[1] + v.R[6 + j] * al[2];  // (R^T alpha)_j
            b += al[j] * v.t[j];
            l1 += std::fabs(al[j]);
        }
        const double na = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
        if (!(std::isfinite(na) && na > 1e-200 && std::isfinite(b))) continue;
        double pl[kCullStride] = {a[0] / na, a[1] / na, a[2] / na, (b - 1e-9 * l1 * tn - 1e-280) / na, 1e-9 * l1 * rf / na};
        bool ok = true;
        for (int j = 0; j < kCullStride; ++j) ok = ok && std::isfinite(pl[j]);
        if (ok)
            for (int j = 0; j < kCullStride; ++j) out[kCullStride * k + j] = pl[j];
    }
}

static constexpr size_t kTallyBytes = sizeof(unsigned long long) * kTallySlots * kTallyStride;
static int ensure_tally(Ctx* c) {
    if (c->d_cull_tally.p) return GSX_OK;
    GSX_HIP(c, c->d_cull_tally.ensure(kTallyBytes));
    GSX_HIP(c, hipMemsetAsync(c->d_cull_tally.p, 0, kTallyBytes, c->stream));
    return GSX_OK;
}

int vote_culled(Ctx* c, int64_t* out, bool reset) {
    unsigned long long h = 0;
    if (c->d_cull_tally.p) {
        GSX_HIP(c, hipSetDevice(c->device));
        std::vector<unsigned long long> slots(kTallySlots * kTallyStride);
        GSX_HIP(c, hipMemcpyAsync(slots.data(), c->d_cull_tally.p, kTallyBytes, hipMemcpyDeviceToHost, c->stream));
        if (reset) GSX_HIP(c, hipMemsetAsync(c->d_cull_tally.p, 0, kTallyBytes, c->stream));
        GSX_HIP(c, hipStreamSynchronize(c->stream));
        for (unsigned k = 0; k < kTallySlots; ++k) h += slots[(size_t)k * kTallyStride];
    }
    *out = (int64_t)h;
    return GSX_OK;
}

// ---- what project_filtered()'s proof assumes about the hardware, measured on the device -------------------------------
// Exhaustive: v_rcp_f32 on EVERY float in [2^-41, 2^41] against the exact reciprocal (r * z in double is exact, so
// |r z - 1| is the relative error of r); and the special cases of v_fract_f32 / v_cvt_flr_i32_f32 the filter meets.
__global__ __launch_bounds__(kBlock) void filter_check_kernel(unsigned lo_bits, unsigned hi_bits, unsigned long long* worst) {
    double m = 0.0;
    for (unsigned long long b = (unsigned long long)lo_bits + (unsigned long long)blockIdx.x * kBlock + threadIdx.x; b <= hi_bits;
         b += (unsigned long long)gridDim.x * kBlock) {
        const float z = __builtin_bit_cast(float, (unsigned)b);
        const float r = __builtin_amdgcn_rcpf(z);
        m = fmax(m, fabs((double)r * (double)z - 1.0));
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) atomicMax(worst, (unsigned long long)__double_as_longlong(m));  // non-negative doubles order as integers
}
__global__ void filter_special_kernel(double* out) {
    const float inf = __builtin_inff(), nan = __builtin_nanf("");
    volatile float in[8] = {inf, -inf, -1e-10f, nan, 1920.5f, -0.25f, 3.0e38f, -3.0e38f};
    for (int k = 0; k < 8; ++k) {
        const float v = in[k];
        out[k] = (double)__builtin_amdgcn_fractf(v);
        out[8 + k] = (double)cvt_floor_i32(v);
    }
}

int filter_check(Ctx* c, double* out) {
    GSX_HIP(c, hipSetDevice(c->device));
    DevBuf buf;
    GSX_HIP(c, buf.ensure(8 + 16 * sizeof(double)));
    GSX_HIP(c, hipMemsetAsync(buf.p, 0, 8 + 16 * sizeof(double), c->stream));
    const unsigned lo = 0x2B000000u /* 2^-41 */, hi = 0x54000000u /* 2^41 */;
    hipLaunchKernelGGL(filter_check_kernel, dim3(4096), dim3(kBlock), 0, c->stream, lo, hi, buf.as<unsigned long long>());
    hipLaunchKernelGGL(filter_special_kernel, dim3(1), dim3(1), 0, c->stream, reinterpret_cast<double*>(buf.as<char>() + 8));
    GSX_HIP(c, hipGetLastError());
    unsigned long long w = 0;
    GSX_HIP(c, hipMemcpyAsync(&w, buf.p, 8, hipMemcpyDeviceToHost, c->stream));
    GSX_HIP(c, hipMemcpyAsync(out + 1, buf.as<char>() + 8, 16 * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    GSX_HIP(c, hipStreamSynchronize(c->stream));
    double m;
    std::memcpy(&m, &w, 8);
    out[0] = m / 5.9604644775390625e-08;  // in units of u = 2^-24
    buf.release();
    return GSX_OK;
}

void debug_cull_planes(const gsx_camera* cam, double* out) {
    ViewDesc vd;
    fill_view_desc(vd, cam, cam->width > 0 ? cam->width : 1, cam->height > 0 ? cam->height : 1, 1, 1);
    cull_planes(vd, out);
}

// Host descriptors -> device: the map's absolute address goes into the device copy (one scalar add less per view
// and wave), the culling planes are derived here.  Staged through pinned memory owned by the ctx and guarded by an
// event, so nothing waits for the stream: the maps that vote_view queued keep flowing while this is prepared.
static int sync_views(Ctx* c) {
    {
        const int rcf = vote_flush_pending(c);  // every consumer of the views comes through here
        if (rcf) return rcf;
    }
    if (!c->views_dirty) return GSX_OK;
    const size_t nv = c->views.size();
    const size_t bytes = sizeof(ViewDesc) * (nv ? nv : 1);
    GSX_HIP(c, c->d_views.ensure(bytes));
    c->views_simple = nv != 0;
    c->views_coarse = c->views_simple;
    if (nv) {
        const int pitch = (int)((nv + 63) / 64 * 64);
        const size_t plane_doubles = (size_t)kCullStride * kCullPlanes * pitch;
        const size_t need = bytes + sizeof(double) * plane_doubles;
        if (c->h_views_ev) GSX_HIP(c, hipEventSynchronize(c->h_views_ev));  // the previous upload has left the buffer
        else GSX_HIP(c, hipEventCreateWithFlags(&c->h_views_ev, hipEventDisableTiming));
        if (c->h_views_cap < need) {
            if (c->h_views) GSX_HIP(c, hipHostFree(c->h_views));
            c->h_views = nullptr;
            c->h_views_cap = 0;
            GSX_HIP(c, hipHostMalloc(&c->h_views, need * 2, hipHostMallocDefault));
            c->h_views_cap = need * 2;
        }
        ViewDesc* dev = static_cast<ViewDesc*>(c->h_views);
        double* planes = reinterpret_cast<double*>(static_cast<char*>(c->h_views) + bytes);
        std::memset(planes, 0, sizeof(double) * plane_doubles);
        const long long base = (long long)reinterpret_cast<uintptr_t>(c->pool_base ? c->pool_base : c->segpool.p);
        for (size_t i = 0; i < nv; ++i) {
            ViewDesc& v = dev[i];
            v = c->views[i];
            v.seg_off += base;
            c->views_simple = c->views_simple && v.unit_scale && v.seg_row_bytes;
            c->views_coarse = c->views_coarse && v.coarse_row_bytes;
            // culling planes, plane-component-major so that lane l reads view l with unit stride
            double pl[kCullStride * kCullPlanes];
            cull_planes(v, pl);
            for (int k = 0; k < kCullStride * kCullPlanes; ++k) planes[(size_t)k * pitch + i] = pl[k];
        }
        GSX_HIP(c, c->d_cull.ensure(sizeof(double) * plane_doubles));
        GSX_HIP(c, hipMemcpyAsync(c->d_views.p, dev, bytes, hipMemcpyHostToDevice, c->stream));
        GSX_HIP(c, hipMemcpyAsync(c->d_cull.p, planes, sizeof(double) * plane_doubles, hipMemcpyHostToDevice, c->stream));
        GSX_HIP(c, hipEventRecord(c->h_views_ev, c->stream));
        c->cull_pitch = pitch;
        {
            const int rct = ensure_tally(c);
            if (rct) return rct;
        }
    }
    c->views_dirty = false;
    return GSX_OK;
}

static inline int odd_dwords(int bytes) {
    int dw = (bytes + 3) / 4;
    return dw | 1;
}

static int ensure_planes(Ctx* c) {
    const size_t esz = c->wide ? 2 : 1;
    const size_t bytes = (size_t)c->bins * (size_t)c->n_pad * esz;
    const bool grew = bytes > c->cnt.cap || bytes > c->fv.cap;
    GSX_HIP(c, c->cnt.ensure(bytes ? bytes : 4));
    GSX_HIP(c, c->fv.ensure(bytes ? bytes : 4));
    if (grew || !(c->planes_valid || c->planes_zero || c->planes_stale)) {
        GSX_HIP(c, hipMemsetAsync(c->cnt.p, 0, bytes, c->stream));
        GSX_HIP(c, hipMemsetAsync(c->fv.p, 0, bytes, c->stream));
        c->planes_zero = true;
        c->planes_valid = false;
        c->planes_stale = false;
    }
    return GSX_OK;
}

// After a rewind the planes still hold the previous run's votes ("stale"): the next flush overwrites every
// element of every Gaussian with its first batch, so nothing is cleared up front (2 x 453 MB of memset per
// step at C3).  Anything that reads the planes before such a flush clears them here.
static int clear_stale_planes(Ctx* c) {
    if (!c->planes_stale) return GSX_OK;
    const size_t bytes = (size_t)c->bins * (size_t)c->n_pad * (c->wide ? 2 : 1);
    GSX_HIP(c, hipMemsetAsync(c->cnt.p, 0, bytes, c->stream));
    GSX_HIP(c, hipMemsetAsync(c->fv.p, 0, bytes, c->stream));
    c->planes_stale = false;
    c->planes_zero = true;
    return GSX_OK;
}

int vote_rewind(Ctx* c) {
    if (!c->vote_begun) return fail(c, GSX_E_STATE, "vote_rewind before vote_begin");
    GSX_HIP(c, hipSetDevice(c->device));
    c->n_flushed = 0;
    c->labels_valid = false;
    c->early_state = -1;  // a rewound run is voted in one piece (its views are all there)
    c->early_batches = 0;
    {
        const int rcj = early_join(c);  // ... behind whatever the early stage is still doing to bcnt / ecnt / erec
        if (rcj) return rcj;
    }
    if (c->planes_valid) {
        c->planes_valid = false;
        c->planes_zero = false;
        c->planes_stale = true;  // cleared lazily, see clear_stale_planes
    }
    return GSX_OK;
}

template <class K>
static int set_lds(Ctx* c, K kernel, size_t bytes) {
    if (bytes > 160 * 1024) return fail(c, GSX_E_UNSUPPORTED, "vote: %zu B of LDS per workgroup exceeds 160 KiB", bytes);
    GSX_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)bytes));
    return GSX_OK;
}

static constexpr int kUnroll = 8;

// ---- early vote ---------------------------------------------------------------------------------------------------
FusedParams fused_params(Ctx* c, int stride_bytes_per_bin);
// A run is the hand-over of its maps (the host pass over them, ~37 us per 1080p map) followed by the vote over all of
// them (~1.2 ms for 3 M Gaussians x 200 views) with the GPU idle during the first and the host during the second.
// When enough of the announced views are staged (`early_vote_at` permille, or a split chosen from the hand-over rate), their
// vote starts on a second stream (vote_fused_planes_kernel into ecnt / efv: counts and first-view codes per bin, plus the
// record of every vote) while the host goes on packing; what vote_finalize still has to do behind the last map is the walk
// over the remaining views plus one pass over the two planes (vote_fused_final_kernel).  Same labels, bit for bit: the
// planes keep everything the tie rule needs.  (Option early_replay: record only + replay, see vote_record_kernel.)
// More than 255 announced views: early_batch_stage starts the batches' count kernels one by one.
// Only where it pays and is simple: one rank holds all views of the run, they arrive one by one through gsx_vote_view, the
// branchless projection is on.  Anything else (rewind, flush, exchange protocols, a pool that had to move) ignores the
// early results and votes all views in one piece as before.
static constexpr int kEarlyPitch = 256;                    // views per block of culling planes (>= kMaxBatch)
static constexpr long long kEarlyMinGaussians = 1 << 18;   // below this the vote is too short to be worth a second stream
static constexpr int kEarlyMinViews = 32;
static constexpr double kEarlyUsPerGaussianView = 4.4e-6;  // early stage on MI355X: 1.6 ms for 3 M Gaussians x 140 views, + 15 %
static constexpr double kRecordUsPerGaussianView = 1.7e-6;  // record-only stage (early_replay): 0.62 ms for the same, + 15 %
static constexpr size_t kEarlySlack = 64 * 260;           // the last stage reads whole groups of four rows, up to row 4 * kRegRows - 1, whatever the bin count
static constexpr size_t kEarlyCullDoubles = (size_t)kCullStride * kCullPlanes * kEarlyPitch;

static bool early_common(const Ctx* c) {
    return c->opt_early_vote && c->first_view == 0 && !c->local_codes && c->slabs == 1 && c->n_flushed == 0 && !c->pool_base &&
           c->opt_flat_project && c->n > 0 && c->bins <= 255 &&
           (c->opt_early_vote == 2 || (c->n >= kEarlyMinGaussians && c->total_views >= kEarlyMinViews));
}
static bool early_possible(const Ctx* c) { return early_common(c) && c->total_views <= kMaxBatch && !c->wide; }
// more than 255 announced views: the batches of labels_batched() (cut by the ANNOUNCED number of views) start their count
// kernels on the second stream as soon as their views are staged; only the last batch, the totals and the tie walk are
// left behind the last map
static bool early_batched_possible(const Ctx* c) { return early_common(c) && c->total_views > kMaxBatch && c->opt_batched_counts; }
static int early_batch_stage(Ctx* c);
// The early batches: a short LAST batch (its count kernel is what stays behind the last map) after balanced ones of <= 255
// views.  Any cut of the view order into consecutive batches gives the same labels (the earliest batch wins a tie).
static int early_batch_count(int total) {
    const int tail = std::min(std::max(total / 16, 16), 64), body = total - tail;
    return (body + kMaxBatch - 1) / kMaxBatch + 1;
}
static void early_batch_bounds(int total, int s, int& lo, int& hi) {
    const int tail = std::min(std::max(total / 16, 16), 64), body = total - tail;
    const int Sb = (body + kMaxBatch - 1) / kMaxBatch;
    if (s >= Sb) {
        lo = body;
        hi = total;
    } else {
        lo = (int)((long long)body * s / Sb);
        hi = (int)((long long)body * (s + 1) / Sb);
    }
}

// Descriptors and culling planes of the views [lo, hi) -> e_views[lo..hi), culling block `block` of e_cull, on stream st.
// dm: the projection variant these views allow.
static int early_upload(Ctx* c, int lo, int hi, int block, int blocks, hipStream_t st, int* dm) {
    const size_t vbytes = sizeof(ViewDesc) * kEarlyPitch, cbytes = sizeof(double) * kEarlyCullDoubles;
    if (!c->h_early) {
        GSX_HIP(c, hipHostMalloc(&c->h_early, vbytes + cbytes, hipHostMallocDefault));
        c->h_early_cap = vbytes + cbytes;
    }
    // sized for the whole run at its first stage: a later stage must find the earlier ones' descriptors and planes in place
    GSX_HIP(c, c->e_views.ensure(sizeof(ViewDesc) * (size_t)std::max(kEarlyPitch, c->total_views)));
    GSX_HIP(c, c->e_cull.ensure(cbytes * (size_t)std::max(2, blocks)));
    if (!c->early_up_ev) GSX_HIP(c, hipEventCreateWithFlags(&c->early_up_ev, hipEventDisableTiming));
    else GSX_HIP(c, hipEventSynchronize(c->early_up_ev));  // the previous upload has left the staging buffer
    ViewDesc* hv = static_cast<ViewDesc*>(c->h_early);  // the stage's views, at most kEarlyPitch
    double* planes = reinterpret_cast<double*>(static_cast<char*>(c->h_early) + vbytes);
    std::memset(planes, 0, cbytes);
    const long long base = (long long)reinterpret_cast<uintptr_t>(c->segpool.p);
    bool simple = true, coarse = true;
    for (int i = lo; i < hi; ++i) {
        ViewDesc& v = hv[i - lo];
        v = c->views[i];
        v.seg_off += base;
        simple = simple && v.unit_scale && v.seg_row_bytes;
        coarse = coarse && v.coarse_row_bytes;
        double pl[kCullStride * kCullPlanes];
        cull_planes(v, pl);
        for (int k = 0; k < kCullStride * kCullPlanes; ++k) planes[(size_t)k * kEarlyPitch + (i - lo)] = pl[k];
    }
    *dm = !simple ? kDivFlat : (coarse && c->opt_seg_coarse ? (c->opt_filter_project ? kDivFiltCoarse : kDivFlatCoarse) : kDivFlatSimple);
    if (hi > lo) GSX_HIP(c, hipMemcpyAsync(c->e_views.as<ViewDesc>() + lo, hv, sizeof(ViewDesc) * (size_t)(hi - lo), hipMemcpyHostToDevice, st));
    GSX_HIP(c, hipMemcpyAsync(c->e_cull.as<double>() + (size_t)block * kEarlyCullDoubles, planes, cbytes, hipMemcpyHostToDevice, st));
    GSX_HIP(c, hipEventRecord(c->early_up_ev, st));
    return GSX_OK;
}

static FusedParams early_params(Ctx* c, int lo, int hi, int block, int stride_bytes_per_bin) {
    FusedParams p = fused_params(c, stride_bytes_per_bin);
    p.views = c->e_views.as<ViewDesc>() + lo;
    p.nviews = hi - lo;
    p.cull = c->opt_wave_cull ? c->e_cull.as<double>() + (size_t)block * kEarlyCullDoubles : nullptr;
    p.cull_pitch = kEarlyPitch;
    return p;
}

// Orders c->stream behind whatever the second stream was last asked to do.  Needed wherever an early stage's buffers (ecnt,
// efv, erec, bcnt) or the pool it reads are touched again: by the last stage that uses its result, and just as much by every
// path that DROPS the result (fewer views than announced, rewind, a new run) and is about to overwrite those buffers.
int early_join(Ctx* c) {
    if (!c->early_inflight) return GSX_OK;
    GSX_HIP(c, hipStreamWaitEvent(c->stream, c->early_done_ev, 0));
    c->early_inflight = false;
    return GSX_OK;
}

// The context's SECOND stream: the early vote's, and the stream of the first extra frame of gsx_render_views (render.hip borrows
// it: a context never labels and renders at the same time, and a process should not hold more streams than the hardware has
// queues - two streams that share a queue serialise even when the GPU has room).
int second_stream(Ctx* c) {
    if (c->stream2) return GSX_OK;
    // lowest priority: the kernels that expand the maps still arriving (c->stream) get the CUs the stage's waves free first
    int least = 0, greatest = 0;
    GSX_HIP(c, hipDeviceGetStreamPriorityRange(&least, &greatest));
    GSX_HIP(c, hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, least));
    return GSX_OK;
}

static int early_streams(Ctx* c) {
    const int rc = second_stream(c);
    if (rc) return rc;
    if (!c->early_maps_ev) GSX_HIP(c, hipEventCreateWithFlags(&c->early_maps_ev, hipEventDisableTiming));
    if (!c->early_done_ev) GSX_HIP(c, hipEventCreateWithFlags(&c->early_done_ev, hipEventDisableTiming));
    return GSX_OK;
}

static int early_vote_stage(Ctx* c) {
    if (c->early_state == 0 && early_batched_possible(c)) return early_batch_stage(c);
    if (c->early_state != 0 || !early_possible(c)) return GSX_OK;
    const int nv = (int)c->views.size();
    int at;
    if (c->opt_early_at > 0) {
        at = std::max(1, (int)(((long long)c->total_views * c->opt_early_at + 999) / 1000));
        if (nv != at) return GSX_OK;
    } else {
        // The split that lets the stage finish just as the last map arrives: it walks a view in about kEarlyUsPerGaussianView * n
        // microseconds, the caller hands over a map every t_map (measured on this run's own calls) - so E / V = t_map / (t_map +
        // that).  Kept between one half and 88 % of the announced views; the stage starts at the first call past it.
        const auto now = std::chrono::steady_clock::now();
        if (nv == 1) c->early_t0 = now;
        if (nv < std::max(8, c->total_views / 4)) return GSX_OK;
        const double t_map = std::chrono::duration<double, std::micro>(now - c->early_t0).count() / (double)(nv - 1);
        const double t_view = (c->opt_early_replay ? kRecordUsPerGaussianView : kEarlyUsPerGaussianView) * (double)c->n;
        const int lo = (c->total_views + 1) / 2, hi = (int)(((long long)c->total_views * 880) / 1000);
        const int want = std::min(std::max((int)std::ceil(c->total_views * t_map / (t_map + t_view)), lo), std::max(lo, hi));
        if (nv < want) return GSX_OK;
        at = nv;
    }
    if (at >= c->total_views) return GSX_OK;
    int rc = vote_flush_pending(c);  // the maps of these views are on their way on c->stream
    if (rc) return rc;
    if ((rc = early_streams(c))) return rc;
    const size_t plane = (size_t)c->bins * (size_t)c->n_pad;
    if (!c->opt_early_replay) {
        GSX_HIP(c, c->ecnt.ensure(plane + kEarlySlack));
        GSX_HIP(c, c->efv.ensure(plane + kEarlySlack));
    }
    GSX_HIP(c, c->erec.ensure((size_t)c->n_pad * (size_t)at));
    if ((rc = ensure_tally(c))) return rc;
    GSX_HIP(c, hipEventRecord(c->early_maps_ev, c->stream));
    GSX_HIP(c, hipStreamWaitEvent(c->stream2, c->early_maps_ev, 0));
    int dm = kDivFlat;
    if ((rc = early_upload(c, 0, at, 0, 2, c->stream2, &dm))) return rc;
    if (c->opt_early_replay) {  // record only: the last stage replays it
        FusedParams p = early_params(c, 0, at, 0, 1);
        auto kr = dm == kDivFiltCoarse ? vote_record_kernel<kUnroll, kDivFiltCoarse>
                 : dm == kDivFlatCoarse ? vote_record_kernel<kUnroll, kDivFlatCoarse>
                  : dm == kDivFlatSimple ? vote_record_kernel<kUnroll, kDivFlatSimple>
                                         : vote_record_kernel<kUnroll, kDivFlat>;
        {
            ProfScope ps(c, "vote_early_record", c->stream2);
            hipLaunchKernelGGL(kr, dim3(grid_for(c->n)), dim3(kBlock), 0, c->stream2, p, p.views, c->erec.as<uint8_t>());
            GSX_HIP(c, hipGetLastError());
        }
        GSX_HIP(c, hipEventRecord(c->early_done_ev, c->stream2));
        c->early_inflight = true;
        c->early_done = at;
        c->early_state = 1;
        c->early_replayed = true;
        return GSX_OK;
    }
    c->early_replayed = false;
    FusedParams p = early_params(c, 0, at, 0, 2);
    const size_t lds = (size_t)kBlock * p.stride_dw * 4;
    auto k = dm == kDivFiltCoarse ? vote_fused_planes_kernel<kUnroll, uint8_t, kDivFiltCoarse>
                 : dm == kDivFlatCoarse ? vote_fused_planes_kernel<kUnroll, uint8_t, kDivFlatCoarse>
             : dm == kDivFlatSimple ? vote_fused_planes_kernel<kUnroll, uint8_t, kDivFlatSimple>
                                    : vote_fused_planes_kernel<kUnroll, uint8_t, kDivFlat>;
    if ((rc = set_lds(c, k, lds))) return rc;
    {
        ProfScope ps(c, "vote_early_planes", c->stream2);
        // fresh planes, one "slab" per wave (sn = 64: the wave-major layout the last stage streams), global view codes 255 - v
        hipLaunchKernelGGL(k, dim3(grid_for(c->n)), dim3(kBlock), lds, c->stream2, p, p.views, c->ecnt.as<uint8_t>(), c->efv.as<uint8_t>(),
                           64LL, 0, 1, 0, c->erec.as<uint8_t>());
        GSX_HIP(c, hipGetLastError());
    }
    GSX_HIP(c, hipEventRecord(c->early_done_ev, c->stream2));
    c->early_inflight = true;
    c->early_done = at;
    c->early_state = 1;
    return GSX_OK;
}

static int early_batch_stage(Ctx* c) {
    const int nv = (int)c->views.size();
    const int S = early_batch_count(c->total_views);
    const int s = c->early_batches;
    if (s >= S - 1) return GSX_OK;  // the last batch ends with the last view: vote_finalize launches it
    int lo, hi;
    early_batch_bounds(c->total_views, s, lo, hi);
    if (nv != hi) return GSX_OK;
    int rc = vote_flush_pending(c);
    if (rc) return rc;
    if ((rc = early_streams(c))) return rc;
    const size_t plane = (size_t)c->bins * (size_t)c->n_pad;
    GSX_HIP(c, c->bcnt.ensure(plane * S));  // all batches at the first stage: a later one must find the earlier planes in place
    if ((rc = ensure_tally(c))) return rc;
    GSX_HIP(c, hipEventRecord(c->early_maps_ev, c->stream));
    GSX_HIP(c, hipStreamWaitEvent(c->stream2, c->early_maps_ev, 0));
    int dm = kDivFlat;
    if ((rc = early_upload(c, lo, hi, s, S, c->stream2, &dm))) return rc;
    FusedParams p = early_params(c, lo, hi, s, 1);
    const size_t lds = (size_t)kBlock * p.stride_dw * 4;
    auto k = dm == kDivFiltCoarse ? vote_fused_counts_kernel<kUnroll, kDivFiltCoarse>
                 : dm == kDivFlatCoarse ? vote_fused_counts_kernel<kUnroll, kDivFlatCoarse>
             : dm == kDivFlatSimple ? vote_fused_counts_kernel<kUnroll, kDivFlatSimple>
                                    : vote_fused_counts_kernel<kUnroll, kDivFlat>;
    if ((rc = set_lds(c, k, lds))) return rc;
    {
        ProfScope ps(c, "vote_early_counts", c->stream2);
        hipLaunchKernelGGL(k, dim3(grid_for(c->n)), dim3(kBlock), lds, c->stream2, p, p.views, c->bcnt.as<uint8_t>() + plane * s, (long long)c->n_pad);
        GSX_HIP(c, hipGetLastError());
    }
    GSX_HIP(c, hipEventRecord(c->early_done_ev, c->stream2));
    c->early_inflight = true;
    c->early_batches = s + 1;
    c->early_done = hi;
    return GSX_OK;
}

// vote_finalize behind an early stage: the views [early_done, nv) on top of the early planes -> c->labels
static int early_vote_finish(Ctx* c) {
    const int nv = (int)c->views.size();
    int rc = vote_flush_pending(c);
    if (rc) return rc;
    int dm = kDivFlat;
    if ((rc = early_upload(c, c->early_done, nv, 1, 2, c->stream, &dm))) return rc;
    GSX_HIP(c, c->labels.ensure(sizeof(int) * (size_t)(c->n_pad ? c->n_pad : 1)));
    if (c->early_replayed) {
        FusedParams pr = early_params(c, c->early_done, nv, 1, 1);
        const size_t ldsr = (size_t)kBlock * pr.stride_dw * 4;
        auto kr = dm == kDivFiltCoarse ? vote_fused_replay_kernel<kUnroll, kDivFiltCoarse>
                 : dm == kDivFlatCoarse ? vote_fused_replay_kernel<kUnroll, kDivFlatCoarse>
                  : dm == kDivFlatSimple ? vote_fused_replay_kernel<kUnroll, kDivFlatSimple>
                                         : vote_fused_replay_kernel<kUnroll, kDivFlat>;
        if ((rc = set_lds(c, kr, ldsr))) return rc;
        if ((rc = early_join(c))) return rc;
        if ((c->opt_ablate >> 4) & 2) pr.nviews = 0;  // timing experiment (tools/early_probe.py)
        ProfScope ps(c, "vote_fused_replay");
        hipLaunchKernelGGL(kr, dim3(grid_for(c->n)), dim3(kBlock), ldsr, c->stream, pr, pr.views, c->erec.as<uint8_t>(), c->early_done,
                           c->labels.as<int>());
        GSX_HIP(c, hipGetLastError());
        return GSX_OK;
    }
    FusedParams p = early_params(c, c->early_done, nv, 1, 1);
    const size_t lds = (size_t)(kBlock / 64) * ((c->bins + 3) / 4) * 256;  // per wave: [bin][64] bytes, rows padded to a multiple of four
    constexpr int kRegRows = 38;  // bins <= 152 (the 150 ADE20K classes + unlabelled): the first-view rows wait in registers
    const bool regs = c->bins <= 4 * kRegRows;
    auto k = dm == kDivFiltCoarse ? (regs ? vote_fused_final_kernel<kUnroll, kDivFiltCoarse, kRegRows> : vote_fused_final_kernel<kUnroll, kDivFiltCoarse, 0>)
                 : dm == kDivFlatCoarse ? (regs ? vote_fused_final_kernel<kUnroll, kDivFlatCoarse, kRegRows> : vote_fused_final_kernel<kUnroll, kDivFlatCoarse, 0>)
             : dm == kDivFlatSimple ? (regs ? vote_fused_final_kernel<kUnroll, kDivFlatSimple, kRegRows> : vote_fused_final_kernel<kUnroll, kDivFlatSimple, 0>)
                                    : (regs ? vote_fused_final_kernel<kUnroll, kDivFlat, kRegRows> : vote_fused_final_kernel<kUnroll, kDivFlat, 0>);
    if ((rc = set_lds(c, k, lds))) return rc;
    if ((rc = early_join(c))) return rc;
    // timing experiments (results invalid), option "ablate": 16 every wave reads the planes of wave 0; 32 no views behind the
    // early ones; 64 no pass over the first-view rows; 128 no look-up in the vote record; 256 labels stored in Morton order;
    // 512 no wave-level culling
    const int ablate = (c->opt_ablate >> 4) & 63;
    if (ablate & 2) p.nviews = 0;
    ProfScope ps(c, "vote_fused_final");
    hipLaunchKernelGGL(k, dim3(grid_for(c->n)), dim3(kBlock), lds, c->stream, p, p.views, c->ecnt.as<uint8_t>(), c->efv.as<uint8_t>(),
                       c->erec.as<uint8_t>(), c->early_done, c->labels.as<int>(), ablate);
    GSX_HIP(c, hipGetLastError());
    return GSX_OK;
}

// merge views [n_flushed, size) into the planes, kMaxBatch at a time
int vote_flush(Ctx* c) {
    if (!c->vote_begun) return fail(c, GSX_E_STATE, "vote_flush before vote_begin");
    GSX_HIP(c, hipSetDevice(c->device));
    int rc = early_join(c);
    if (rc) return rc;
    if ((rc = sync_views(c))) return rc;
    rc = ensure_planes(c);
    if (rc) return rc;
    const int nv = (int)c->views.size();
    if (c->n <= 0 || c->n_flushed >= nv) {
        c->n_flushed = nv;
        return clear_stale_planes(c);  // nothing will overwrite them
    }
    FusedParams p{};
    p.x = c->x.as<float>();
    p.y = c->y.as<float>();
    p.z = c->z.as<float>();
    p.n = c->n;
    p.pool = c->segpool.as<uint8_t>();
    p.bins = c->bins;
    p.xcd_swizzle = c->opt_xcd_swizzle;
    p.perm = c->sorted ? c->perm.as<uint32_t>() : nullptr;
    p.stride_dw = odd_dwords(c->bins * 2);
    p.filt_h = filter_half_width(c);
    const size_t lds = (size_t)kBlock * p.stride_dw * 4;
    while (c->n_flushed < nv) {
        const int batch = std::min(kMaxBatch, nv - c->n_flushed);
        p.views = c->d_views.as<ViewDesc>() + c->n_flushed;
        p.nviews = batch;
        p.cull = c->opt_wave_cull ? c->d_cull.as<double>() + c->n_flushed : nullptr;
        p.cull_pitch = c->cull_pitch;
        p.cull_tally = c->d_cull_tally.as<unsigned long long>();
        const int view_base = c->first_view + c->n_flushed;
        const int fresh = (c->planes_zero || c->planes_stale) ? 1 : 0;
        ProfScope ps(c, "vote_fused_planes");
        if (c->wide) {
            const int dm = div_mode(c);
            auto k = dm == kDivFiltCoarse ? vote_fused_planes_kernel<kUnroll, uint16_t, kDivFiltCoarse>
                 : dm == kDivFlatCoarse ? vote_fused_planes_kernel<kUnroll, uint16_t, kDivFlatCoarse>
                     : dm == kDivFlatSimple ? vote_fused_planes_kernel<kUnroll, uint16_t, kDivFlatSimple>
                     : dm == kDivFlat     ? vote_fused_planes_kernel<kUnroll, uint16_t, kDivFlat>
                     : dm == kDivCertified ? vote_fused_planes_kernel<kUnroll, uint16_t, kDivCertified>
                                           : vote_fused_planes_kernel<kUnroll, uint16_t, kDivExact>;
            if ((rc = set_lds(c, k, lds))) return rc;
            hipLaunchKernelGGL(k, dim3(grid_for(c->n)), dim3(kBlock), lds, c->stream, p, p.views, c->cnt.as<uint16_t>(),
                               c->fv.as<uint16_t>(), (long long)c->sn, view_base, fresh, 0, (uint8_t*)nullptr);
        } else {
            const int dm = div_mode(c);
            auto k = dm == kDivFiltCoarse ? vote_fused_planes_kernel<kUnroll, uint8_t, kDivFiltCoarse>
                 : dm == kDivFlatCoarse ? vote_fused_planes_kernel<kUnroll, uint8_t, kDivFlatCoarse>
                     : dm == kDivFlatSimple ? vote_fused_planes_kernel<kUnroll, uint8_t, kDivFlatSimple>
                     : dm == kDivFlat     ? vote_fused_planes_kernel<kUnroll, uint8_t, kDivFlat>
                     : dm == kDivCertified ? vote_fused_planes_kernel<kUnroll, uint8_t, kDivCertified>
                                           : vote_fused_planes_kernel<kUnroll, uint8_t, kDivExact>;
            if ((rc = set_lds(c, k, lds))) return rc;
            hipLaunchKernelGGL(k, dim3(grid_for(c->n)), dim3(kBlock), lds, c->stream, p, p.views, c->cnt.as<uint8_t>(),
                               c->fv.as<uint8_t>(), (long long)c->sn, view_base, fresh, c->local_codes ? 1 : 0, (uint8_t*)nullptr);
        }
        GSX_HIP(c, hipGetLastError());
        c->planes_zero = false;
        c->planes_stale = false;
        c->planes_valid = true;
        c->n_flushed += batch;
    }
    return GSX_OK;
}

int vote_tiebreak_keys(Ctx* c) {
    if (!c->vote_begun) return fail(c, GSX_E_STATE, "vote_tiebreak_keys before vote_begin");
    GSX_HIP(c, hipSetDevice(c->device));
    int rc = ensure_planes(c);
    if (rc) return rc;
    if ((rc = clear_stale_planes(c))) return rc;
    GSX_HIP(c, c->keys.ensure(sizeof(int) * (size_t)(c->n_pad ? c->n_pad : 1)));
    GSX_HIP(c, hipMemsetAsync(c->keys.p, 0, sizeof(int) * (size_t)c->n_pad, c->stream));
    if (c->n <= 0) return GSX_OK;
    ProfScope ps(c, "vote_keys");
    if (c->wide)
        hipLaunchKernelGGL(vote_keys_kernel<uint16_t>, dim3(grid_for(c->n)), dim3(kBlock), 0, c->stream,
                           c->cnt.as<uint16_t>(), c->fv.as<uint16_t>(), (long long)c->n, (long long)c->sn, c->bins,
                           c->keys.as<int>());
    else
        hipLaunchKernelGGL(vote_keys_kernel<uint8_t>, dim3(grid_for(c->n)), dim3(kBlock), 0, c->stream,
                           c->cnt.as<uint8_t>(), c->fv.as<uint8_t>(), (long long)c->n, (long long)c->sn, c->bins,
                           c->keys.as<int>());
    GSX_HIP(c, hipGetLastError());
    return GSX_OK;
}

// Last step of every protocol: the labels (and the device-side range flag) to the host.  A label is -1 .. 254, so it
// crosses PCIe as ONE byte (bin = label + 1; labels_narrow_kernel, 3 MB instead of 12 MB for 3 M Gaussians: the D2H is
// on the critical path of every run) into pinned memory, in kLabelChunks pieces; the worker threads widen piece i into
// the caller's (pageable) int32 array while piece i+1 is still on the link.  A map that gsx_vote_view_device packed
// with a label out of range fails the call here.
static constexpr int kBadLabelWord = -2;  // errflag value: c->labels held something that is not a label (never a view index)
__global__ __launch_bounds__(kBlock) void labels_narrow_kernel(const int* __restrict__ labels, long long n, uint8_t* __restrict__ out,
                                                               int* __restrict__ errflag) {
    const long long i4 = ((long long)blockIdx.x * kBlock + threadIdx.x) * 4;
    if (i4 >= n) return;
    unsigned b[4];
    if (i4 + 4 <= n) {
        const int4 v = *reinterpret_cast<const int4*>(labels + i4);
        b[0] = (unsigned)v.x + 1u, b[1] = (unsigned)v.y + 1u, b[2] = (unsigned)v.z + 1u, b[3] = (unsigned)v.w + 1u;
        if ((b[0] | b[1] | b[2] | b[3]) > 255u) atomicMin(errflag, kBadLabelWord);
        *reinterpret_cast<uint32_t*>(out + i4) = b[0] | (b[1] << 8) | (b[2] << 16) | (b[3] << 24);
    } else {
        for (long long i = i4; i < n; ++i) {
            const unsigned bb = (unsigned)labels[i] + 1u;
            if (bb > 255u) atomicMin(errflag, kBadLabelWord);
            out[i] = (uint8_t)bb;
        }
    }
}

static int labels_to_host(Ctx* c, int32_t* labels_out) {
    c->labels_valid = true;
    const size_t n = labels_out && c->n > 0 ? (size_t)c->n : 0;
    const size_t esz = c->opt_labels_u8 ? 1 : sizeof(int);  // bytes per label on the link ("labels_u8" = 0: the A/B of DESIGN.md)
    const size_t nbytes = n * esz;
    const size_t flag_off = (nbytes + 63) / 64 * 64;
    const size_t need = flag_off + 64;
    if (c->h_labels_cap < need) {
        if (c->h_labels) GSX_HIP(c, hipHostFree(c->h_labels));
        c->h_labels = nullptr;
        c->h_labels_cap = 0;
        GSX_HIP(c, hipHostMalloc(&c->h_labels, need + need / 8, hipHostMallocDefault));
        c->h_labels_cap = need + need / 8;
    }
    char* land = static_cast<char*>(c->h_labels);
    int* flag = reinterpret_cast<int*>(land + flag_off);
    const char* src = c->labels.as<char>();
    if (n && c->opt_labels_u8) {
        GSX_HIP(c, c->labels8.ensure((n + 255) / 256 * 256));
        ProfScope ps(c, "labels_narrow");
        hipLaunchKernelGGL(labels_narrow_kernel, dim3(grid_for((long long)(n + 3) / 4)), dim3(kBlock), 0, c->stream, c->labels.as<int>(),
                           (long long)n, c->labels8.as<uint8_t>(), c->errflag.as<int>());
        GSX_HIP(c, hipGetLastError());
        src = c->labels8.as<char>();
    }
    GSX_HIP(c, hipMemcpyAsync(flag, c->errflag.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    if (n) {
        const int chunks = nbytes >= ((size_t)1 << 20) ? kLabelChunks : 1;
        const size_t per = ((n + chunks - 1) / chunks + 4095) / 4096 * 4096;  // labels per chunk
        for (int k = 0; k < chunks; ++k) {
            const size_t lo = std::min(n, per * k), hi = std::min(n, per * (k + 1));
            if (lo < hi) GSX_HIP(c, hipMemcpyAsync(land + lo * esz, src + lo * esz, (hi - lo) * esz, hipMemcpyDeviceToHost, c->stream));
            if (!c->h_ev[k]) GSX_HIP(c, hipEventCreateWithFlags(&c->h_ev[k], hipEventDisableTiming));
            GSX_HIP(c, hipEventRecord(c->h_ev[k], c->stream));
        }
        Workers* w = host_workers(c);
        for (int k = 0; k < chunks; ++k) {
            const size_t lo = std::min(n, per * k), hi = std::min(n, per * (k + 1));
            GSX_HIP(c, hipEventSynchronize(c->h_ev[k]));
            if (lo >= hi) continue;
            if (c->opt_labels_u8) host_widen_labels(w, labels_out + lo, reinterpret_cast<const uint8_t*>(land) + lo, hi - lo);
            else host_copy(w, labels_out + lo, land + lo * esz, (hi - lo) * esz);
        }
    }
    GSX_HIP(c, hipStreamSynchronize(c->stream));
    if (*flag != kNoBadView) {
        const int view = *flag;
        GSX_HIP(c, hipMemsetAsync(c->errflag.p, 0x7f, sizeof(int), c->stream));
        if (view == kBadLabelWord)
            return fail(c, GSX_E_STATE, "the label buffer holds a value outside [-1, 254] (a corrupt exchange buffer?)");
        return fail(c, GSX_E_RANGE, "the segmentation map of view %d (gsx_vote_view_device) holds a label outside [-1, %d]", view,
                    c->n_classes - 1);
    }
    return GSX_OK;
}

int vote_labels_from_keys(Ctx* c, int32_t* labels_out) {
    if (!c->vote_begun) return fail(c, GSX_E_STATE, "vote_labels_from_keys before vote_begin");
    if (c->keys.cap < sizeof(int) * (size_t)c->n_pad) return fail(c, GSX_E_STATE, "vote_labels_from_keys before vote_tiebreak_keys");
    GSX_HIP(c, hipSetDevice(c->device));
    GSX_HIP(c, c->labels.ensure(sizeof(int) * (size_t)(c->n_pad ? c->n_pad : 1)));
    if (c->n > 0) {
        ProfScope ps(c, "vote_labels");
        hipLaunchKernelGGL(vote_labels_kernel, dim3(grid_for(c->n)), dim3(kBlock), 0, c->stream, c->keys.as<int>(),
                           (long long)c->n, c->sorted ? c->perm.as<uint32_t>() : nullptr, c->labels.as<int>());
        GSX_HIP(c, hipGetLastError());
    }
    return labels_to_host(c, labels_out);
}

FusedParams fused_params(Ctx* c, int stride_bytes_per_bin);

// A contiguous range of the Gaussians in upload (Morton) order: the whole scene for a single GPU, one rank's slab in
// exchange protocol v4.
struct VoteRange {
    long long i0, n;
};
static FusedParams range_params(Ctx* c, const VoteRange& r, int stride_bytes_per_bin) {
    FusedParams p = fused_params(c, stride_bytes_per_bin);
    p.x += r.i0;
    p.y += r.i0;
    p.z += r.i0;
    p.n = r.n;
    p.perm = nullptr;
    return p;
}

// <= 255 staged views: labels straight out of the fused kernel.  perm == nullptr: out[i] = label of Gaussian i0 + i
// (Morton order); otherwise out[perm[i0 + i]] (the caller's order).
static int labels_one_batch(Ctx* c, const VoteRange& r, const uint32_t* perm, int* out) {
    if (r.n <= 0) return GSX_OK;
    FusedParams p = range_params(c, r, 1);
    p.perm = perm ? perm + r.i0 : nullptr;
    const size_t lds = (size_t)kBlock * p.stride_dw * 4;
    // kernel variants: unroll U in {2,4,8} x division mode x batched LDS reads
    using K = void (*)(FusedParams, const ViewDesc*, int*);
    const int ui = c->opt_vote_unroll == 2 ? 0 : c->opt_vote_unroll == 4 ? 1 : 2;
#define GSX_ROW(U_) \
    {{vote_fused_labels_kernel<U_, kDivExact, false>, vote_fused_labels_kernel<U_, kDivExact, true>},         \
     {vote_fused_labels_kernel<U_, kDivCertified, false>, vote_fused_labels_kernel<U_, kDivCertified, true>}, \
     {vote_fused_labels_kernel<U_, kDivFlat, false>, vote_fused_labels_kernel<U_, kDivFlat, true>},           \
     {vote_fused_labels_kernel<U_, kDivFlatSimple, false>, vote_fused_labels_kernel<U_, kDivFlatSimple, true>}, \
     {vote_fused_labels_kernel<U_, kDivFlatCoarse, false>, vote_fused_labels_kernel<U_, kDivFlatCoarse, true>},   \
     {vote_fused_labels_kernel<U_, kDivFiltCoarse, false>, vote_fused_labels_kernel<U_, kDivFiltCoarse, true>}}
    static const K table[3][kDivModes][2] = {GSX_ROW(2), GSX_ROW(4), GSX_ROW(8)};
#undef GSX_ROW
    K k = table[ui][div_mode(c)][c->opt_lds_batch ? 1 : 0];
    int rc = set_lds(c, k, lds);
    if (rc) return rc;
    ProfScope ps(c, "vote_fused_labels");
    hipLaunchKernelGGL(k, dim3(grid_for(r.n)), dim3(kBlock), lds, c->stream, p, p.views, out);
    GSX_HIP(c, hipGetLastError());
    return GSX_OK;
}

// More than 255 views on one GPU (the reference's own cameras.json holds 311): the views are cut into S balanced
// batches of <= 255 that play the ranks of exchange protocol v3 on a single device.  Every batch is one launch of the
// fast u8-histogram walk (vote_fused_counts_kernel) into its own count plane; vote_slab_totals_kernel sums the S
// planes per bin (unique maximum -> label, else a candidate mask); vote_tie_kernel walks each batch forward for the
// tied Gaussians and vote_tie_resolve_kernel takes the earliest batch that voted a candidate = the earliest view,
// which is the reference's first-inserted rule (dls.py:303).  Replaces the 16-bit count + first-view planes of
// vote_flush() for this case (1.9 ms per 200 views at C3 sizes against 1.1 ms).
// Leaves the range's labels in Morton order at c->keys[0 .. r.n).
static int labels_batched(Ctx* c, const VoteRange& r) {
    const int nv = (int)c->views.size();
    // batches whose count kernels already ran on the second stream while the run's later maps were handed over (early_batch_stage:
    // same batches - the run brought exactly the announced views -, same planes, the whole scene)
    const bool early = c->early_batches > 0 && c->early_state == 0 && early_batched_possible(c) && nv == c->total_views && r.i0 == 0 &&
                       r.n == c->n;
    const int S = early ? early_batch_count(nv) : (nv + kMaxBatch - 1) / kMaxBatch;
    const long long npad = (r.n + 255) / 256 * 256;
    const size_t plane = (size_t)c->bins * (size_t)npad;
    GSX_HIP(c, c->bcnt.ensure(plane * S));
    GSX_HIP(c, c->cand.ensure(sizeof(uint32_t) * kCandWords * (size_t)npad));
    GSX_HIP(c, c->bcodes.ensure(sizeof(uint16_t) * (size_t)npad * S));
    if (r.n <= 0) return GSX_OK;
    const int dm = div_mode(c);
    auto k = dm == kDivFiltCoarse ? vote_fused_counts_kernel<kUnroll, kDivFiltCoarse>
                 : dm == kDivFlatCoarse ? vote_fused_counts_kernel<kUnroll, kDivFlatCoarse>
             : dm == kDivFlatSimple ? vote_fused_counts_kernel<kUnroll, kDivFlatSimple>
             : dm == kDivFlat     ? vote_fused_counts_kernel<kUnroll, kDivFlat>
             : dm == kDivCertified ? vote_fused_counts_kernel<kUnroll, kDivCertified>
                                   : vote_fused_counts_kernel<kUnroll, kDivExact>;
    FusedParams base = range_params(c, r, 1);
    const size_t lds = (size_t)kBlock * base.stride_dw * 4;
    int rc = set_lds(c, k, lds);
    if (rc) return rc;
    auto batch = [&](int s, FusedParams& p) {  // views [lo, hi) of batch s: balanced (sizes differ by at most one), or the early cut
        int lo = (int)((long long)nv * s / S), hi = (int)((long long)nv * (s + 1) / S);
        if (early) early_batch_bounds(nv, s, lo, hi);
        p = base;
        p.views = base.views + lo;
        p.nviews = hi - lo;
        if (p.cull) p.cull = base.cull + lo;
    };
    const int s0 = early ? c->early_batches : 0;
    for (int s = s0; s < S; ++s) {
        FusedParams p;
        batch(s, p);
        ProfScope ps(c, "vote_fused_counts");
        hipLaunchKernelGGL(k, dim3(grid_for(r.n)), dim3(kBlock), lds, c->stream, p, p.views, c->bcnt.as<uint8_t>() + plane * s, npad);
    }
    GSX_HIP(c, hipGetLastError());
    {
        ProfScope ps(c, "vote_slab_totals");
        hipLaunchKernelGGL(vote_slab_totals_kernel, dim3(grid_for(npad / 4)), dim3(kBlock), 0, c->stream, c->bcnt.as<uint8_t>(), S,
                           c->bins, npad, c->keys.as<int>(), c->cand.as<uint32_t>());
    }
    for (int s = 0; s < S; ++s) {
        FusedParams p;
        batch(s, p);
        ProfScope ps(c, "vote_tie");
        hipLaunchKernelGGL(vote_tie_kernel, dim3(grid_for(r.n)), dim3(kBlock), 0, c->stream, p, p.views, c->cand.as<uint32_t>(), npad,
                           c->bcodes.as<uint16_t>() + (size_t)npad * s, c->bcodes.as<uint16_t>(), s);
    }
    {
        ProfScope ps(c, "vote_tie_resolve");
        hipLaunchKernelGGL(vote_tie_resolve_kernel, dim3(grid_for(npad)), dim3(kBlock), 0, c->stream, c->bcodes.as<uint16_t>(), S, npad,
                           c->keys.as<int>());
    }
    GSX_HIP(c, hipGetLastError());
    return GSX_OK;
}

// labels of a range of Gaussians over ALL staged views, whatever their number.  to_sorted: leave them in Morton order
// at c->keys[0 .. r.n) (protocol v4); otherwise write c->labels in the caller's order (needs r = the whole scene).
static int labels_for_range(Ctx* c, const VoteRange& r, bool to_sorted) {
    // early or not: an early batch may still be writing its plane of bcnt on the second stream (a run that stopped short of the
    // announced views, a rewind) - the buffers are (re)sized and the kernels launched behind it
    int rc = early_join(c);
    if (rc) return rc;
    if ((rc = sync_views(c))) return rc;
    GSX_HIP(c, c->keys.ensure(sizeof(int) * (size_t)(c->n_pad ? c->n_pad : 1)));
    GSX_HIP(c, c->labels.ensure(sizeof(int) * (size_t)(c->n_pad ? c->n_pad : 1)));
    const uint32_t* perm = c->sorted ? c->perm.as<uint32_t>() : nullptr;
    if ((int)c->views.size() <= kMaxBatch)
        return labels_one_batch(c, r, to_sorted ? nullptr : perm, to_sorted ? c->keys.as<int>() : c->labels.as<int>());
    if ((rc = labels_batched(c, r))) return rc;
    if (!to_sorted && r.n > 0) {
        ProfScope ps(c, "vote_labels");
        hipLaunchKernelGGL(unpermute_labels_kernel, dim3(grid_for(r.n)), dim3(kBlock), 0, c->stream, c->keys.as<int>(), r.n, perm,
                           c->labels.as<int>());
        GSX_HIP(c, hipGetLastError());
    }
    return GSX_OK;
}

int vote_finalize(Ctx* c, int32_t* labels_out) {
    if (!c->vote_begun) return fail(c, GSX_E_STATE, "vote_finalize before vote_begin");
    GSX_HIP(c, hipSetDevice(c->device));
    const int nv = (int)c->views.size();
    const bool batched_ok = !c->local_codes && c->opt_batched_counts;
    if (c->early_state == 1 && early_possible(c) && nv >= c->early_done && nv <= kMaxBatch) {
        // the first views were voted while the rest was handed over: only the views behind them are left
        int rc = early_vote_finish(c);
        if (rc) return rc;
        return labels_to_host(c, labels_out);
    }
    if (c->n_flushed == 0 && (nv <= kMaxBatch || batched_ok)) {
        // nothing in the planes yet: labels come straight out of the fused kernel(s)
        int rc = labels_for_range(c, VoteRange{0, c->n}, false);
        if (rc) return rc;
        return labels_to_host(c, labels_out);
    }
    int rc = vote_flush(c);
    if (rc) return rc;
    if ((rc = vote_tiebreak_keys(c))) return rc;
    return vote_labels_from_keys(c, labels_out);
}

// ---- exchange v4 host side: views sharded for the hand-over, Gaussians sharded for the vote ------------------------
// The packed maps of 200 1080p views are 0.44 GB; the dense vote histogram of 3 M Gaussians is 0.45 GB PER RANK.  So
// the maps travel (one all-gather), not the votes: afterwards every rank holds every view, votes its own slab of
// the Gaussians with the single-GPU kernel, and only 4 bytes per Gaussian (the labels) are gathered.  No histogram,
// no tie-break exchange: a slab sees all views in order, the tie rule is the single-GPU one.
int vote_export(Ctx* c, int64_t reserve_bytes, void* blobs_out, void** pool_dev, int64_t* pool_bytes) {
    if (!c->vote_begun) return fail(c, GSX_E_STATE, "vote_export before vote_begin");
    if (c->pool_base) return fail(c, GSX_E_STATE, "vote_export after vote_import");
    GSX_HIP(c, hipSetDevice(c->device));
    if (reserve_bytes < 0) return fail(c, GSX_E_INVALID, "vote_export: negative size");
    {
        const int rcf = vote_flush_pending(c);  // the caller is about to read the pool (all-gather)
        if (rcf) return rcf;
    }
    const size_t used = (c->seg_used + 255) / 256 * 256;
    int rc = pool_reserve(c, std::max<size_t>(std::max<size_t>((size_t)reserve_bytes, used), 256));
    if (rc) return rc;
    c->views_dirty = true;  // the pool may have moved
    static_assert(sizeof(ViewDesc) == GSX_VIEW_BLOB_BYTES, "view blob size is part of the ABI");
    if (blobs_out && !c->views.empty()) std::memcpy(blobs_out, c->views.data(), sizeof(ViewDesc) * c->views.size());
    if (pool_dev) *pool_dev = c->segpool.p;
    if (pool_bytes) *pool_bytes = (int64_t)used;
    return GSX_OK;
}

// The imported views replace the rank's own; those are kept aside so that gsx_vote_import_undo can put them back (a protocol
// that imports optimistically and then learns that a peer's pool was not what the schedule assumed falls back to the plain
// exchange, which starts from the rank's own views again).
static void import_commit(Ctx* c, std::vector<ViewDesc>& all, const void* pool_all_dev) {
    if (!c->pool_base) {  // (a second import on top of an import keeps the first one's saved state)
        c->own_views.swap(c->views);
        c->own_first_view = c->first_view;
    }
    c->views.swap(all);
    c->views_dirty = true;
    c->pool_base = pool_all_dev;
    c->first_view = 0;
    c->n_flushed = 0;
    c->labels_valid = false;
}

int vote_import_undo(Ctx* c) {
    if (!c->vote_begun || !c->pool_base) return fail(c, GSX_E_STATE, "vote_import_undo without a gsx_vote_import");
    c->views.swap(c->own_views);
    c->own_views.clear();
    c->first_view = c->own_first_view;
    c->pool_base = nullptr;
    c->views_dirty = true;
    c->n_flushed = 0;
    c->labels_valid = false;
    return GSX_OK;
}

int vote_import(Ctx* c, int n_parts, const int32_t* part_views, const int64_t* part_offsets, const void* blobs,
                const void* pool_all_dev, int64_t pool_all_bytes) {
    if (!c->vote_begun) return fail(c, GSX_E_STATE, "vote_import before vote_begin");
    if (n_parts < 1 || !part_views || !part_offsets || !pool_all_dev || pool_all_bytes < 0)
        return fail(c, GSX_E_INVALID, "vote_import: bad arguments");
    long long total = 0;
    for (int r = 0; r < n_parts; ++r) {
        if (part_views[r] < 0 || part_offsets[r] < 0) return fail(c, GSX_E_INVALID, "vote_import: negative count or offset");
        total += part_views[r];
    }
    if (total > 65535) return fail(c, GSX_E_RANGE, "vote_import: %lld views exceed 65535", total);
    if (total > 0 && !blobs) return fail(c, GSX_E_INVALID, "vote_import: blobs is NULL");
    std::vector<ViewDesc> all((size_t)total);
    if (total) std::memcpy(all.data(), blobs, sizeof(ViewDesc) * (size_t)total);
    size_t k = 0;
    for (int r = 0; r < n_parts; ++r)
        for (int v = 0; v < part_views[r]; ++v, ++k) {
            ViewDesc& d = all[k];
            // the blob comes from another process: check what the kernels rely on before any address is formed from it
            const MapLayout L = map_layout(d.seg_w, d.seg_h, d.seg_row_bytes != 0, d.coarse_row_bytes != 0);
            const bool ok = d.seg_w >= 1 && d.seg_h >= 1 && d.seg_w <= 65535 && d.seg_h <= 65535 && d.seg_off >= 0 &&
                            d.seg_row_bytes == L.strip_bytes && d.coarse_row_bytes == L.cstrip_bytes &&
                            (!d.coarse_row_bytes || d.coarse_delta == (unsigned)L.coarse_off) &&
                            (unsigned long long)d.seg_off + (unsigned long long)part_offsets[r] + L.map_bytes <= (unsigned long long)pool_all_bytes;
            if (!ok) return fail(c, GSX_E_INVALID, "vote_import: descriptor %zu (part %d) is inconsistent with the gathered pool", k, r);
            d.seg_off += part_offsets[r];
            d.hw32 = (float)d.half_w;  // (never trusted from the blob: the filter's proof needs exactly these)
            d.hh32 = (float)d.half_h;
        }
    import_commit(c, all, pool_all_dev);
    return GSX_OK;
}

// gsx_vote_import without the blobs: every rank of a run holds the whole camera list (cameras.json) and, when all maps of the
// run share ONE geometry, knows where view k of part r lies in the gathered buffer - part_offsets[r] + k * stride, the stride
// gsx_vote_view gives a map of that geometry under this context's options (all ranks run the same build with the same
// options).  So the descriptors are derived here, the same way gsx_vote_view derives them (fill_view_desc + map_layout),
// and the header exchange of the blobs and its host wait disappear from the protocol.
int vote_import_uniform(Ctx* c, int n_parts, const int32_t* part_views, const int64_t* part_offsets, const gsx_camera* cams,
                        int seg_w, int seg_h, int img_w, int img_h, const void* pool_all_dev, int64_t pool_all_bytes) {
    if (!c->vote_begun) return fail(c, GSX_E_STATE, "vote_import_uniform before vote_begin");
    if (n_parts < 1 || !part_views || !part_offsets || !pool_all_dev || pool_all_bytes < 0)
        return fail(c, GSX_E_INVALID, "vote_import_uniform: bad arguments");
    if (seg_w < 1 || seg_h < 1 || img_w < 1 || img_h < 1 || seg_w > 65535 || seg_h > 65535)
        return fail(c, GSX_E_INVALID, "vote_import_uniform: map %dx%d, image %dx%d", seg_w, seg_h, img_w, img_h);
    long long total = 0;
    for (int r = 0; r < n_parts; ++r) {
        if (part_views[r] < 0 || part_offsets[r] < 0) return fail(c, GSX_E_INVALID, "vote_import_uniform: negative count or offset");
        total += part_views[r];
    }
    if (total > 65535) return fail(c, GSX_E_RANGE, "vote_import_uniform: %lld views exceed 65535", total);
    if (total > 0 && !cams) return fail(c, GSX_E_INVALID, "vote_import_uniform: cams is NULL");
    const MapLayout L = map_layout(seg_w, seg_h, c->opt_seg_tiled != 0, c->opt_seg_coarse && c->bins <= 255);  // as view_prologue
    const size_t stride = (L.map_bytes + 255) / 256 * 256;
    std::vector<ViewDesc> all((size_t)total);
    size_t k = 0;
    for (int r = 0; r < n_parts; ++r)
        for (int v = 0; v < part_views[r]; ++v, ++k) {
            const unsigned long long off = (unsigned long long)part_offsets[r] + (unsigned long long)v * stride;
            if (off + L.map_bytes > (unsigned long long)pool_all_bytes)
                return fail(c, GSX_E_INVALID, "vote_import_uniform: view %zu (part %d) lies outside the gathered pool", k, r);
            ViewDesc& d = all[k];
            fill_view_desc(d, &cams[k], L.w, L.h, img_w, img_h);
            d.seg_off = (long long)off;
            d.seg_row_bytes = L.strip_bytes;
            d.coarse_row_bytes = L.cstrip_bytes;
            d.coarse_delta = (unsigned)L.coarse_off;
        }
    import_commit(c, all, pool_all_dev);
    return GSX_OK;
}

int vote_slab_labels(Ctx* c, int slab, int slabs, int64_t* slab_size) {
    if (!c->vote_begun) return fail(c, GSX_E_STATE, "vote_slab_labels before vote_begin");
    if (slabs < 1 || slab < 0 || slab >= slabs) return fail(c, GSX_E_INVALID, "vote_slab_labels: slab %d of %d", slab, slabs);
    GSX_HIP(c, hipSetDevice(c->device));
    long long sn = ((c->n + slabs - 1) / slabs + 255) / 256 * 256;
    if (sn == 0) sn = 256;
    if (slab_size) *slab_size = sn;
    const long long i0 = std::min<long long>(c->n, sn * slab);
    const long long cnt = std::min<long long>(c->n - i0, sn);
    if ((long long)sn > c->n_pad) {  // tiny scenes: the key buffer must hold one whole slab for the all-gather
        GSX_HIP(c, c->keys.ensure(sizeof(int) * (size_t)sn));
    }
    return labels_for_range(c, VoteRange{i0, cnt}, true);
}

// ---- exchange v3 host side ------------------------------------------------------------------------------------
FusedParams fused_params(Ctx* c, int stride_bytes_per_bin) {
    FusedParams p{};
    p.x = c->x.as<float>();
    p.y = c->y.as<float>();
    p.z = c->z.as<float>();
    p.n = c->n;
    p.views = c->d_views.as<ViewDesc>();
    p.nviews = (int)c->views.size();
    p.pool = c->segpool.as<uint8_t>();
    p.bins = c->bins;
    p.xcd_swizzle = c->opt_xcd_swizzle;
    p.perm = c->sorted ? c->perm.as<uint32_t>() : nullptr;
    p.stride_dw = odd_dwords(c->bins * stride_bytes_per_bin);
    p.cull = c->opt_wave_cull ? c->d_cull.as<double>() : nullptr;
    p.cull_pitch = c->cull_pitch;
    p.cull_tally = c->d_cull_tally.as<unsigned long long>();
    p.filt_h = filter_half_width(c);
    return p;
}

int vote_flush_counts(Ctx* c) {
    if (!c->vote_begun || !c->local_codes) return fail(c, GSX_E_STATE, "vote_flush_counts needs vote_begin with the 'exchange_local' option");
    GSX_HIP(c, hipSetDevice(c->device));
    int rc = sync_views(c);
    if (rc) return rc;
    const size_t bytes = (size_t)c->bins * (size_t)c->n_pad;
    if (bytes > c->cnt.cap) {
        GSX_HIP(c, c->cnt.ensure(bytes));
        GSX_HIP(c, hipMemsetAsync(c->cnt.p, 0, bytes, c->stream));  // pad Gaussians stay zero for ever
    }
    if (c->n > 0) {
        FusedParams p = fused_params(c, 1);
        const size_t lds = (size_t)kBlock * p.stride_dw * 4;
        const int dm = div_mode(c);
        auto k = dm == kDivFiltCoarse ? vote_fused_counts_kernel<kUnroll, kDivFiltCoarse>
                 : dm == kDivFlatCoarse ? vote_fused_counts_kernel<kUnroll, kDivFlatCoarse>
                 : dm == kDivFlatSimple ? vote_fused_counts_kernel<kUnroll, kDivFlatSimple>
                 : dm == kDivFlat     ? vote_fused_counts_kernel<kUnroll, kDivFlat>
                 : dm == kDivCertified ? vote_fused_counts_kernel<kUnroll, kDivCertified>
                                       : vote_fused_counts_kernel<kUnroll, kDivExact>;
        if ((rc = set_lds(c, k, lds))) return rc;
        ProfScope ps(c, "vote_fused_counts");
        hipLaunchKernelGGL(k, dim3(grid_for(c->n)), dim3(kBlock), lds, c->stream, p, p.views, c->cnt.as<uint8_t>(),
                           (long long)c->sn);
        GSX_HIP(c, hipGetLastError());
    }
    c->n_flushed = (int)c->views.size();
    return GSX_OK;
}

int vote_slab_totals(Ctx* c, const void* recv_cnt) {
    if (!c->vote_begun || !c->local_codes) return fail(c, GSX_E_STATE, "vote_slab_totals needs the 'exchange_local' option");
    if (!recv_cnt) return fail(c, GSX_E_INVALID, "vote_slab_totals: NULL argument");
    GSX_HIP(c, hipSetDevice(c->device));
    GSX_HIP(c, c->keys.ensure(sizeof(int) * (size_t)c->n_pad));
    GSX_HIP(c, c->cand.ensure(sizeof(uint32_t) * kCandWords * (size_t)c->sn));
    ProfScope ps(c, "vote_slab_totals");
    hipLaunchKernelGGL(vote_slab_totals_kernel, dim3(grid_for(c->sn / 4)), dim3(kBlock), 0, c->stream, (const uint8_t*)recv_cnt,
                       c->slabs, c->bins, (long long)c->sn, c->keys.as<int>(), c->cand.as<uint32_t>());
    GSX_HIP(c, hipGetLastError());
    return GSX_OK;  // ordered by the ctx stream; nothing waits on the host
}

int vote_tie_codes(Ctx* c, const void* cand_all) {
    if (!c->vote_begun || !c->local_codes) return fail(c, GSX_E_STATE, "vote_tie_codes needs the 'exchange_local' option");
    if (!cand_all) return fail(c, GSX_E_INVALID, "vote_tie_codes: NULL argument");
    GSX_HIP(c, hipSetDevice(c->device));
    const size_t bytes = sizeof(uint16_t) * (size_t)c->n_pad;
    if (bytes > c->codes.cap) {
        GSX_HIP(c, c->codes.ensure(bytes));
        GSX_HIP(c, hipMemsetAsync(c->codes.p, 0, bytes, c->stream));
    }
    if (c->n > 0) {
        FusedParams p = fused_params(c, 1);
        ProfScope ps(c, "vote_tie");
        hipLaunchKernelGGL(vote_tie_kernel, dim3(grid_for(c->n)), dim3(kBlock), 0, c->stream, p, p.views,
                           (const uint32_t*)cand_all, (long long)c->sn, c->codes.as<uint16_t>(), (const uint16_t*)nullptr, 0);
        GSX_HIP(c, hipGetLastError());
    }
    return GSX_OK;  // ordered by the ctx stream; nothing waits on the host
}

int vote_tie_resolve(Ctx* c, const void* recv_codes) {
    if (!c->vote_begun || !c->local_codes) return fail(c, GSX_E_STATE, "vote_tie_resolve needs the 'exchange_local' option");
    if (!recv_codes) return fail(c, GSX_E_INVALID, "vote_tie_resolve: NULL argument");
    GSX_HIP(c, hipSetDevice(c->device));
    ProfScope ps(c, "vote_tie_resolve");
    hipLaunchKernelGGL(vote_tie_resolve_kernel, dim3(grid_for(c->sn)), dim3(kBlock), 0, c->stream, (const uint16_t*)recv_codes,
                       c->slabs, (long long)c->sn, c->keys.as<int>());
    GSX_HIP(c, hipGetLastError());
    return GSX_OK;  // ordered by the ctx stream; nothing waits on the host
}

int vote_slab_reduce(Ctx* c, const void* recv_cnt, const void* recv_fv) {
    if (!c->vote_begun || !c->local_codes) return fail(c, GSX_E_STATE, "vote_slab_reduce needs vote_begin with the 'exchange_local' option");
    if (!recv_cnt || !recv_fv) return fail(c, GSX_E_INVALID, "vote_slab_reduce: NULL argument");
    GSX_HIP(c, hipSetDevice(c->device));
    GSX_HIP(c, c->keys.ensure(sizeof(int) * (size_t)c->n_pad));
    ProfScope ps(c, "vote_slab_reduce");
    hipLaunchKernelGGL(vote_slab_reduce_kernel, dim3(grid_for(c->sn)), dim3(kBlock), 0, c->stream, (const uint8_t*)recv_cnt,
                       (const uint8_t*)recv_fv, c->slabs, c->bins, (long long)c->sn, c->keys.as<int>());
    GSX_HIP(c, hipGetLastError());
    return GSX_OK;  // ordered by the ctx stream; nothing waits on the host
}

int vote_labels_from_sorted(Ctx* c, const void* sorted_labels_dev, int32_t* labels_out) {
    if (!c->vote_begun) return fail(c, GSX_E_STATE, "vote_labels_from_sorted before vote_begin");
    if (!sorted_labels_dev) return fail(c, GSX_E_INVALID, "vote_labels_from_sorted: NULL argument");
    GSX_HIP(c, hipSetDevice(c->device));
    GSX_HIP(c, c->labels.ensure(sizeof(int) * (size_t)(c->n_pad ? c->n_pad : 1)));
    if (c->n > 0) {
        ProfScope ps(c, "vote_labels");
        hipLaunchKernelGGL(unpermute_labels_kernel, dim3(grid_for(c->n)), dim3(kBlock), 0, c->stream,
                           (const int*)sorted_labels_dev, (long long)c->n, c->sorted ? c->perm.as<uint32_t>() : nullptr,
                           c->labels.as<int>());
        GSX_HIP(c, hipGetLastError());
    }
    return labels_to_host(c, labels_out);
}

int vote_debug_planes(Ctx* c, uint16_t* counts_out, uint16_t* first_out) {
    if (!c->vote_begun || !(c->planes_valid || c->planes_zero || c->planes_stale)) return fail(c, GSX_E_STATE, "vote_debug_planes: no planes");
    {
        const int rc0 = clear_stale_planes(c);
        if (rc0) return rc0;
    }
    if (!counts_out || !first_out) return fail(c, GSX_E_INVALID, "vote_debug_planes: NULL argument");
    GSX_HIP(c, hipSetDevice(c->device));
    const size_t esz = c->wide ? 2 : 1;
    const size_t elems = (size_t)c->bins * (size_t)c->n_pad;
    std::vector<uint8_t> tmp(elems * esz);
    std::vector<uint32_t> perm;
    if (c->sorted && c->n > 0) {
        perm.resize((size_t)c->n);
        GSX_HIP(c, hipMemcpy(perm.data(), c->perm.p, sizeof(uint32_t) * (size_t)c->n, hipMemcpyDeviceToHost));
    }
    for (int which = 0; which < 2; ++which) {
        GSX_HIP(c, hipMemcpyAsync(tmp.data(), which ? c->fv.p : c->cnt.p, elems * esz, hipMemcpyDeviceToHost, c->stream));
        GSX_HIP(c, hipStreamSynchronize(c->stream));
        uint16_t* out = which ? first_out : counts_out;
        for (int b = 0; b < c->bins; ++b)
            for (int64_t i = 0; i < c->n; ++i) {
                const size_t slab = (size_t)i / (size_t)c->sn;
                const size_t at = (slab * c->bins + b) * (size_t)c->sn + ((size_t)i - slab * (size_t)c->sn);
                out[(size_t)b * c->n + (perm.empty() ? (size_t)i : (size_t)perm[i])] = c->wide ? reinterpret_cast<uint16_t*>(tmp.data())[at] : tmp[at];
            }
    }
    return GSX_OK;
}

}  // namespace gsx
