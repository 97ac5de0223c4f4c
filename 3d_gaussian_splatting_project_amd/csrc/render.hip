// render.hip — tile-based forward rasterizer for gfx950, bit-faithful to the reference viewer's data path.
//
// Restates Web_Viewer_Gaussians_Selection/gaussians_selection.js ("gs.js"):
//   splat_importance / splat_pack   processPlyBuffer   gs.js:513-582  (fp64 like JS numbers)
//                                   generateTexture    gs.js:301-354  (4*Sigma as truncated fp16)
//   pre_kernel (depth + vertex + SH) runSort           gs.js:432-447  (depth int, 16-bit bucket)
//                                   vertex shader      gs.js:696-750  (fp32, no contraction)
//   bin / sort / ranges             the global stable counting sort (gs.js:450-457) becomes two stable
//                                   radix sorts: splats by bucket, then (tile, splat) pairs by tile
//   blend.hip                       fragment shader + blend unit gs.js:782-799, 1036-1038
// This TU is compiled with -ffp-contract=off: JS never fuses, and the vertex stage must produce
// bit-identical axes to the oracle so that the fragment `discard` (A < -4) decides identically.
//
// HBM layout
//   tex        uint4[2n]   the viewer's RGBA32UI texel pairs: [x y z label][h01 h23 h45 rgba8], importance order
//   buffer     u8[32n]     the viewer's .splat rows (pos, exp(scale), rgba8, quat8)
//   rec        3 x float4 per splat per view, ONE 48-byte record: (cx, cy, g0x, g0y) (g1x, g1y, r, g) (b, a, -, -).  (Rounds 1-2 and
//              most of round 3 kept three arrays: a blend gather then touched three 64-byte segments per record instead of 1.5.)
//   keys/vals  u32[P]      (tile, splat) pairs emitted in depth order, P = sum of tiles touched
//   ranges     int2[tiles] [start, end) into the sorted pairs
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <new>
#include <thread>

#include "gsx_ctx.hpp"
#include "tile_test.hpp"

namespace gsx {

static constexpr int kRB = 256;

// ---- JavaScript number semantics -----------------------------------------------------------------
__device__ __forceinline__ unsigned js_u8clamp(double x) {  // Uint8ClampedArray store
    if (!(x > 0.0)) return 0u;
    if (x >= 255.0) return 255u;
    const double f = floor(x);
    if (f + 0.5 < x) return (unsigned)(f + 1.0);
    if (x < f + 0.5) return (unsigned)f;
    return (unsigned)(fmod(f, 2.0) == 0.0 ? f : f + 1.0);
}

__device__ __forceinline__ int js_toint32(double x) {  // `x | 0`
    if (!(fabs(x) <= 1.7976931348623157e308)) return 0;  // NaN, +-Infinity
    const double t = trunc(x);
    if (t >= -2147483648.0 && t <= 2147483647.0) return (int)t;
    double m = fmod(t, 4294967296.0);
    if (m < 0) m += 4294967296.0;
    return (int)(unsigned)m;
}

__device__ __forceinline__ unsigned js_float_to_half(double v) {  // gs.js:248-275
    const float fl = (float)v;
    const int f = __float_as_int(fl);
    const int sign = (f >> 31) & 0x0001;
    const int exp = (f >> 23) & 0x00ff;
    int frac = f & 0x007fffff;
    int newExp;
    if (exp == 0) {
        newExp = 0;
    } else if (exp < 113) {
        newExp = 0;
        frac |= 0x00800000;
        frac = frac >> ((113 - exp) & 31);  // JS shift counts are taken mod 32
        if (frac & 0x01000000) {
            newExp = 1;
            frac = 0;
        }
    } else if (exp < 142) {
        newExp = exp - 112;
    } else {
        newExp = 31;
        frac = 0;
    }
    return (unsigned)((sign << 15) | (newExp << 10) | (frac >> 13));
}

__device__ __forceinline__ float half_to_float(unsigned h) {  // unpackHalf2x16: exact
    const unsigned s = (h >> 15) & 1u, e = (h >> 10) & 31u, m = h & 1023u;
    float v;
    if (e == 0) v = (float)m * 5.9604644775390625e-08f;  // 2^-24
    else if (e == 31) v = m ? __int_as_float(0x7fc00000) : __int_as_float(0x7f800000);
    else v = __int_as_float((int)(((e + 112u) << 23) | (m << 13)));
    return s ? -v : v;
}

// ---- pack: processPlyBuffer + generateTexture --------------------------------------------------------
// importance (gs.js:520-522) as a sort key: ascending ~bits == descending float (values are >= 0)
__global__ __launch_bounds__(kRB) void splat_importance_kernel(const float* __restrict__ scale,
                                                                const float* __restrict__ opacity, long long n,
                                                                uint32_t* __restrict__ key, uint32_t* __restrict__ idx) {
    const long long r = (long long)blockIdx.x * kRB + threadIdx.x;
    if (r >= n) return;
    const double size = exp((double)scale[3 * r]) * exp((double)scale[3 * r + 1]) * exp((double)scale[3 * r + 2]);
    const double op = 1.0 / (1.0 + exp(-(double)opacity[r]));
    const float imp = (float)(size * op);  // Float32Array store
    key[r] = ~__float_as_uint(imp);
    idx[r] = (uint32_t)r;
}

__global__ __launch_bounds__(kRB) void iota_kernel(uint32_t* __restrict__ idx, long long n) {
    const long long r = (long long)blockIdx.x * kRB + threadIdx.x;
    if (r < n) idx[r] = (uint32_t)r;
}

__global__ __launch_bounds__(kRB) void splat_pack_kernel(const uint32_t* __restrict__ order, long long n,
                                                          const float* __restrict__ xyz, const float* __restrict__ scale,
                                                          const float* __restrict__ rot, const float* __restrict__ opacity,
                                                          const float* __restrict__ f_dc, const int* __restrict__ labels,
                                                          uint4* __restrict__ buffer /*2 per splat*/,
                                                          uint4* __restrict__ tex /*2 per splat*/) {
    const long long j = (long long)blockIdx.x * kRB + threadIdx.x;
    if (j >= n) return;
    const long long r = order[j];
    const float px = xyz[3 * r], py = xyz[3 * r + 1], pz = xyz[3 * r + 2];
    float sc[3];
    unsigned q[4];
    if (scale) {
        const double r0 = rot[4 * r], r1 = rot[4 * r + 1], r2 = rot[4 * r + 2], r3 = rot[4 * r + 3];
        const double qlen = sqrt(r0 * r0 + r1 * r1 + r2 * r2 + r3 * r3);  // gs.js:549
        q[0] = js_u8clamp((r0 / qlen) * 128.0 + 128.0);
        q[1] = js_u8clamp((r1 / qlen) * 128.0 + 128.0);
        q[2] = js_u8clamp((r2 / qlen) * 128.0 + 128.0);
        q[3] = js_u8clamp((r3 / qlen) * 128.0 + 128.0);
        sc[0] = (float)exp((double)scale[3 * r]);
        sc[1] = (float)exp((double)scale[3 * r + 1]);
        sc[2] = (float)exp((double)scale[3 * r + 2]);
    } else {  // gs.js:559-563
        sc[0] = sc[1] = sc[2] = (float)0.01;
        q[0] = 255u;
        q[1] = q[2] = q[3] = 0u;
    }
    const double SH_C0 = 0.28209479177387814;
    unsigned c[4];
    for (int k = 0; k < 3; ++k) c[k] = js_u8clamp((0.5 + SH_C0 * (double)f_dc[3 * r + k]) * 255.0);  // gs.js:567-569
    c[3] = opacity ? js_u8clamp((1.0 / (1.0 + exp(-(double)opacity[r]))) * 255.0) : 255u;           // gs.js:576
    const unsigned rgba = c[0] | (c[1] << 8) | (c[2] << 16) | (c[3] << 24);
    const unsigned quat = q[0] | (q[1] << 8) | (q[2] << 16) | (q[3] << 24);
    buffer[2 * j] = make_uint4(__float_as_uint(px), __float_as_uint(py), __float_as_uint(pz), __float_as_uint(sc[0]));
    buffer[2 * j + 1] = make_uint4(__float_as_uint(sc[1]), __float_as_uint(sc[2]), rgba, quat);

    // generateTexture, gs.js:322-353: covariance from the QUANTISED quaternion and the f32 scales, in fp64
    double rt[4];
    for (int k = 0; k < 4; ++k) rt[k] = ((double)q[k] - 128.0) / 128.0;
    double M[9] = {
        1.0 - 2.0 * (rt[2] * rt[2] + rt[3] * rt[3]), 2.0 * (rt[1] * rt[2] + rt[0] * rt[3]), 2.0 * (rt[1] * rt[3] - rt[0] * rt[2]),
        2.0 * (rt[1] * rt[2] - rt[0] * rt[3]), 1.0 - 2.0 * (rt[1] * rt[1] + rt[3] * rt[3]), 2.0 * (rt[2] * rt[3] + rt[0] * rt[1]),
        2.0 * (rt[1] * rt[3] + rt[0] * rt[2]), 2.0 * (rt[2] * rt[3] - rt[0] * rt[1]), 1.0 - 2.0 * (rt[1] * rt[1] + rt[2] * rt[2]),
    };
#pragma unroll
    for (int k = 0; k < 9; ++k) M[k] = M[k] * (double)sc[k / 3];
    const double s0 = M[0] * M[0] + M[3] * M[3] + M[6] * M[6];
    const double s1 = M[0] * M[1] + M[3] * M[4] + M[6] * M[7];
    const double s2 = M[0] * M[2] + M[3] * M[5] + M[6] * M[8];
    const double s3 = M[1] * M[1] + M[4] * M[4] + M[7] * M[7];
    const double s4 = M[1] * M[2] + M[4] * M[5] + M[7] * M[8];
    const double s5 = M[2] * M[2] + M[5] * M[5] + M[8] * M[8];
    const unsigned h01 = js_float_to_half(4 * s0) | (js_float_to_half(4 * s1) << 16);
    const unsigned h23 = js_float_to_half(4 * s2) | (js_float_to_half(4 * s3) << 16);
    const unsigned h45 = js_float_to_half(4 * s4) | (js_float_to_half(4 * s5) << 16);
    const float lab = (float)(labels ? labels[r] : -999999);  // NO_SELECTION, gs.js:6,579; texdata_f[..+3] = label
    tex[2 * j] = make_uint4(__float_as_uint(px), __float_as_uint(py), __float_as_uint(pz), __float_as_uint(lab));
    tex[2 * j + 1] = make_uint4(h01, h23, h45, rgba);
}

// ---- spherical-harmonics colour (degrees 1..3; not in the reference, see oracle/render_oracle.c) ----------
// coefficients re-laid at upload as coef[k][c][i] (k = 0 is f_dc), importance order: per view every thread reads
// its 3K coefficients from 3K planes, each load a fully used 256-B run per wave (as [k][i][c] the 12-byte stride
// made every load instruction straddle six lines).
__global__ __launch_bounds__(kRB) void sh_pack_kernel(const uint32_t* __restrict__ order, long long n,
                                                       const float* __restrict__ f_dc, const float* __restrict__ f_rest,
                                                       int K, float* __restrict__ coef) {
    const long long j = (long long)blockIdx.x * kRB + threadIdx.x;
    if (j >= n) return;
    const long long r = order[j];
    const int K1 = K - 1;
    for (int c = 0; c < 3; ++c) {
        coef[((size_t)0 * 3 + c) * n + j] = f_dc[3 * r + c];
        for (int k = 1; k < K; ++k) coef[((size_t)k * 3 + c) * n + j] = f_rest[(size_t)r * 3 * K1 + (size_t)c * K1 + (k - 1)];
    }
}

// colour of one splat seen from the camera position: clamp(0.5 + SH(dir), 0, 1) per channel.
// Coef: where coefficient (k, c) of the splat comes from - CoefMem reads the planes (one view at a time), CoefRegs holds the
// splat's coefficients in registers (pre_multi_kernel: read once, used by every view of the group).  Same operations in the
// same order either way.
struct CoefMem {
    const float* __restrict__ coef;
    long long n, i;
    __device__ __forceinline__ float get(int k, int c) const { return coef[((size_t)k * 3 + c) * (size_t)n + (size_t)i]; }
};
struct CoefRegs {
    float v[48];
    __device__ __forceinline__ float get(int k, int c) const { return v[3 * k + c]; }
};

// real SH basis function k of the unit direction (x, y, z): ONE set of expressions for every caller, so that the colour of a
// splat does not depend on which kernel evaluated it (no contraction in this file; k is a compile-time constant wherever
// this is called, the switch folds away)
__device__ __forceinline__ float sh_basis(int k, float x, float y, float z) {
    const float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
    const float C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f, 0.5462742152960396f};
    const float C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                         -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    switch (k) {
        case 0: return C0;
        case 1: return -C1 * y;
        case 2: return C1 * z;
        case 3: return -C1 * x;
        case 4: return C2[0] * xy;
        case 5: return C2[1] * yz;
        case 6: return C2[2] * (2.0f * zz - xx - yy);
        case 7: return C2[3] * xz;
        case 8: return C2[4] * (xx - yy);
        case 9: return C3[0] * y * (3.0f * xx - yy);
        case 10: return C3[1] * xy * z;
        case 11: return C3[2] * y * (4.0f * zz - xx - yy);
        case 12: return C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
        case 13: return C3[4] * x * (4.0f * zz - xx - yy);
        case 14: return C3[5] * z * (xx - yy);
        default: return C3[6] * x * (xx - 3.0f * yy);
    }
}

// unit vector from the camera position to the splat
__device__ __forceinline__ void sh_dir(float px, float py, float pz, float cpx, float cpy, float cpz, float& x, float& y, float& z) {
    const float dx = px - cpx, dy = py - cpy, dz = pz - cpz;
    const float len = sqrtf(dx * dx + dy * dy + dz * dz);
    x = dx / len, y = dy / len, z = dz / len;
}

__device__ __forceinline__ float sh_clamp(float acc) {
    const float v = acc + 0.5f;
    return v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
}

template <class Coef>
__device__ __forceinline__ void sh_color(const Coef& cf, int deg, float px, float py, float pz, float cpx, float cpy, float cpz,
                                         float rgb[3]) {
    float x, y, z;
    sh_dir(px, py, pz, cpx, cpy, cpz, x, y, z);
    const int K = (deg + 1) * (deg + 1);
    float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        if (k < K) {
            const float b = sh_basis(k, x, y, z);
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[c] = k == 0 ? b * cf.get(0, c) : acc[c] + b * cf.get(k, c);
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) rgb[c] = sh_clamp(acc[c]);
}

// ---- per view ------------------------------------------------------------------------------------------
struct ViewUniforms {
    double vp2, vp6, vp10;  // row 2 of proj*view (gs.js:437)
    float view[16];         // uniforms as the GPU sees them: f32
    float proj[16];
    float fx, fy, W, H;
    int tiles_x, tiles_y;
};

// (edge_min_q / tile_touches - the exact "does the ellipse reach this tile?" test - live in tile_test.hpp: the blend kernel uses
// them too)
static constexpr uint32_t kEmptyRect = 1u;  // tx0 = 1 > tx1 = 0: covers no tile

// One splat in one view, everything but the colour: depth key (gs.js:436-441: ((vp2 x + vp6 y + vp10 z) * 4096) | 0, fp64),
// vertex shader (gs.js:696-750, fp32 in the oracle's operation order) and the tile rectangle of the ellipse's bounding box.
// colour: the splat reaches a pixel of this view (r0, g1 and fade are valid, a colour is wanted).
struct PreGeom {
    int depth;
    uint32_t rect;
    float4 r0;       // (cx, cy, g0x, g0y)
    float g1x, g1y;  // second axis: the first half of r1
    float fade;
    bool colour;
};

__device__ __forceinline__ PreGeom pre_geom(const uint4 t0, const uint4 t1, const ViewUniforms& u) {
    PreGeom o;
    const float cx_ = __uint_as_float(t0.x), cy_ = __uint_as_float(t0.y), cz_ = __uint_as_float(t0.z);
    o.depth = js_toint32((u.vp2 * (double)cx_ + u.vp6 * (double)cy_ + u.vp10 * (double)cz_) * 4096.0);
    o.rect = kEmptyRect;
    o.r0 = make_float4(0.f, 0.f, 0.f, 0.f);
    o.g1x = o.g1y = o.fade = 0.f;
    o.colour = false;
    // ---- vertex shader, fp32 ----
    float cam[4], p2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) cam[k] = u.view[k] * cx_ + u.view[4 + k] * cy_ + u.view[8 + k] * cz_ + u.view[12 + k] * 1.0f;
#pragma unroll
    for (int k = 0; k < 4; ++k) p2[k] = u.proj[k] * cam[0] + u.proj[4 + k] * cam[1] + u.proj[8 + k] * cam[2] + u.proj[12 + k] * cam[3];
    const float clip = 1.2f * p2[3];
    bool drawn = !(p2[2] < -clip || p2[0] < -clip || p2[0] > clip || p2[1] < -clip || p2[1] > clip);
    if (drawn) {
        const float u1x = half_to_float(t1.x & 0xffffu), u1y = half_to_float(t1.x >> 16);
        const float u2x = half_to_float(t1.y & 0xffffu), u2y = half_to_float(t1.y >> 16);
        const float u3x = half_to_float(t1.z & 0xffffu), u3y = half_to_float(t1.z >> 16);
        const float V[3][3] = {{u1x, u1y, u2x}, {u1y, u2y, u3x}, {u2x, u3x, u3y}};
        const float ja = u.fx / cam[2], jb = -(u.fx * cam[0]) / (cam[2] * cam[2]);
        const float jc = -u.fy / cam[2], jd = (u.fy * cam[1]) / (cam[2] * cam[2]);
        float ta[3], tb[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            ta[k] = u.view[4 * k + 0] * ja + u.view[4 * k + 1] * 0.0f + u.view[4 * k + 2] * jb;
            tb[k] = u.view[4 * k + 0] * 0.0f + u.view[4 * k + 1] * jc + u.view[4 * k + 2] * jd;
        }
        float a0[3], a1[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            a0[k] = ta[0] * V[k][0] + ta[1] * V[k][1] + ta[2] * V[k][2];
            a1[k] = tb[0] * V[k][0] + tb[1] * V[k][1] + tb[2] * V[k][2];
        }
        const float c00 = a0[0] * ta[0] + a0[1] * ta[1] + a0[2] * ta[2];
        const float c01 = a1[0] * ta[0] + a1[1] * ta[1] + a1[2] * ta[2];
        const float c11 = a1[0] * tb[0] + a1[1] * tb[1] + a1[2] * tb[2];
        const float mid = (c00 + c11) / 2.0f;
        const float hx = (c00 - c11) / 2.0f;
        const float radius = sqrtf(hx * hx + c01 * c01);
        const float l1 = mid + radius, l2 = mid - radius;
        if (l2 < 0.0f) drawn = false;  // gs.js:736
        const float dx = c01, dy = l1 - c00;
        const float dl = sqrtf(dx * dx + dy * dy);
        const float ux = dx / dl, uy = dy / dl;
        const float s1 = fminf(sqrtf(2.0f * l1), 1024.0f), s2 = fminf(sqrtf(2.0f * l2), 1024.0f);
        const float mx = s1 * ux, my = s1 * uy;    // majorAxis
        const float nx = s2 * uy, ny = s2 * -ux;   // minorAxis
        float fade = p2[2] / p2[3] + 1.0f;
        fade = fade < 0.0f ? 0.0f : (fade > 1.0f ? 1.0f : fade);
        const float ndcx = p2[0] / p2[3], ndcy = p2[1] / p2[3];
        const float wcx = (ndcx + 1.0f) * 0.5f * u.W;  // GL window coordinates, y up
        const float wcy = (ndcy + 1.0f) * 0.5f * u.H;
        const float m2 = mx * mx + my * my, n2 = nx * nx + ny * ny;
        const float big = 3.0e38f;
        if (!(m2 > 0.0f) || !(n2 > 0.0f) || !(m2 < big) || !(n2 < big) || !(fabsf(wcx) < big) || !(fabsf(wcy) < big)) drawn = false;
        if (drawn) {
            // pixels whose CENTRE can satisfy |vPosition| <= 2: the bounding box of the ellipse with the orthogonal half-axes m
            // and n (vPosition = (2 d.m/|m|^2, 2 d.n/|n|^2)), i.e. |x + 0.5 - cx| <= sqrt(mx^2 + nx^2), + 1/64 px for the fp32
            // rounding of the per-pixel evaluation (which is good to ~1e-6 of the extent; cx itself to 1.2e-4 px at 1920).
            // (Rounds 1-2 took a whole pixel of slack and floor() on both sides: 1.5 px too wide on every side, i.e. one more
            // tile column or row for every sixth splat - 12 % more (tile, splat) pairs to bin, sort and stage.)
            const float ex = sqrtf(mx * mx + nx * nx) + 0.015625f, ey = sqrtf(my * my + ny * ny) + 0.015625f;
            const float top = u.H - wcy;  // image row coordinate of the centre
            int x0 = (int)ceilf(fminf(fmaxf(wcx - ex - 0.5f, -1.0f), u.W)), x1 = (int)floorf(fminf(fmaxf(wcx + ex - 0.5f, -1.0f), u.W));
            int y0 = (int)ceilf(fminf(fmaxf(top - ey - 0.5f, -1.0f), u.H)), y1 = (int)floorf(fminf(fmaxf(top + ey - 0.5f, -1.0f), u.H));
            x0 = max(x0, 0);
            y0 = max(y0, 0);
            x1 = min(x1, (int)u.W - 1);
            y1 = min(y1, (int)u.H - 1);
            if (x1 >= x0 && y1 >= y0) {
                const uint32_t tx0 = x0 >> 4, tx1 = x1 >> 4, ty0 = y0 >> 4, ty1 = y1 >> 4;
                o.rect = tx0 | (tx1 << 8) | (ty0 << 16) | (ty1 << 24);
                o.r0 = make_float4(wcx, wcy, 2.0f * mx / m2, 2.0f * my / m2);
                o.g1x = 2.0f * nx / n2;
                o.g1y = 2.0f * ny / n2;
                o.fade = fade;
                o.colour = true;
            }
        }
    }
    return o;
}

// the reference's colour: fade * rgba8 / 255 (gs.js:741-742); the SH colour replaces its first three channels
__device__ __forceinline__ float rgba8_channel(const uint4 t1, int k, float fade) { return fade * (float)((t1.w >> (8 * k)) & 0xffu) / 255.0f; }

// One splat in one view, complete (pre_kernel).  The colour is only needed for splats that reach a pixel: 192 B of SH
// coefficients per splat.  (Deferring it further, to the splats a depth phase really bins, was measured: those are visited in
// DEPTH order, the coefficient reads become gathers and cost 9x what the skipped splats save.)
struct PreOut {
    int depth;
    uint32_t rect;
    float4 r0, r1;
    float2 r2;
};

template <class Coef>
__device__ __forceinline__ PreOut pre_one(const uint4 t0, const uint4 t1, const ViewUniforms& u, bool sh_on, int sh_deg, float cpx,
                                          float cpy, float cpz, const Coef& cf) {
    const PreGeom g = pre_geom(t0, t1, u);
    PreOut o;
    o.depth = g.depth;
    o.rect = g.rect;
    o.r0 = g.r0;
    o.r1 = make_float4(0.f, 0.f, 0.f, 0.f);
    o.r2 = make_float2(0.f, 0.f);
    if (g.colour) {
        float col[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) col[k] = rgba8_channel(t1, k, g.fade);
        if (sh_on) {
            float rgb[3];
            sh_color(cf, sh_deg, __uint_as_float(t0.x), __uint_as_float(t0.y), __uint_as_float(t0.z), cpx, cpy, cpz, rgb);
#pragma unroll
            for (int k = 0; k < 3; ++k) col[k] = g.fade * rgb[k];
        }
        o.r1 = make_float4(g.g1x, g.g1y, col[0], col[1]);
        o.r2 = make_float2(col[2], col[3]);
    }
    return o;
}

// workgroup-wide min / max of the depth keys -> one atomic pair per workgroup (same-address atomics serialise in L2)
__device__ __forceinline__ void depth_range_to(int lo, int hi, int* __restrict__ pre, int* slo /*[4]*/, int* shi /*[4]*/) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = min(lo, __shfl_xor(lo, o));
        hi = max(hi, __shfl_xor(hi, o));
    }
    __syncthreads();  // (the previous view's values have been read)
    if ((threadIdx.x & 63) == 0) {
        slo[threadIdx.x >> 6] = lo;
        shi[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMin(&pre[0], min(min(slo[0], slo[1]), min(slo[2], slo[3])));
        atomicMax(&pre[1], max(max(shi[0], shi[1]), max(shi[2], shi[3])));
    }
}

// One pass over the texel pairs per view.  pre[4] = {min depth, max depth, splat 0's tile rectangle, -}: min / max over ALL
// splats (gs.js:436-441); the rectangle of splat 0 is kept apart because the blend's epilogue draws splat 0 again even when
// bucket_kernel drops it (and clears its entry of tile_rect).
__global__ __launch_bounds__(kRB) void pre_kernel(const uint4* __restrict__ tex, long long n, ViewUniforms u,
                                                   const float* __restrict__ sh_coef, int sh_deg, float cpx, float cpy, float cpz,
                                                   int* __restrict__ depth, int* __restrict__ pre,
                                                   float4* __restrict__ rec, uint32_t* __restrict__ tile_rect) {
    __shared__ int slo[4], shi[4];
    int lo = 2147483647, hi = -2147483647 - 1;
    // grid-stride: the launch is capped at 2048 workgroups, i.e. 4096 same-address atomics per frame (one pair per
    // workgroup of a 12 k-workgroup launch kept the L2 busy for 0.2 ms after the last wave had finished)
    for (long long i = (long long)blockIdx.x * kRB + threadIdx.x; i < n; i += (long long)gridDim.x * kRB) {
        const uint4 t0 = tex[2 * i], t1 = tex[2 * i + 1];
        const CoefMem cf{sh_coef, n, i};
        const PreOut o = pre_one(t0, t1, u, sh_coef != nullptr, sh_deg, cpx, cpy, cpz, cf);
        depth[i] = o.depth;
        lo = min(lo, o.depth);
        hi = max(hi, o.depth);
        if (o.rect != kEmptyRect || i == 0) {  // a record is read only through a list the splat was binned into - and splat 0's by the epilogue
            rec[3 * i] = o.r0;
            rec[3 * i + 1] = o.r1;
            rec[3 * i + 2] = make_float4(o.r2.x, o.r2.y, 0.f, 0.f);
        }
        tile_rect[i] = o.rect;
        if (i == 0) pre[2] = (int)o.rect;
    }
    depth_range_to(lo, hi, pre, slo, shi);
}

// The same for NV views (1 .. kPreViews) in ONE pass over the scene: a splat's texel pair and its SH coefficients (224 B at
// degree 3) are read ONCE - gsx_render_views keeps several frames in flight, and each of them used to stream the whole scene
// again (4 x 816 MB at 3 M splats).  The loop is turned inside out for that: first the geometry of every view (pre_geom, the
// code pre_kernel runs), then one pass over the coefficients in which every view that draws the splat accumulates its own
// colour - three accumulators and a direction per view, the basis function recomputed from the direction when its
// coefficient arrives (sh_basis: the expressions of sh_color).  Per view the operations are pre_kernel's in pre_kernel's
// order, on the same operands: the frames are bit-identical.  (Two earlier forms, both measured: the view loop around
// pre_one with the coefficients re-read through L2 - they had left the 4 MB L2 by the next view, 2.2 GB fetched per 4-view
// launch instead of 0.67 - and with the coefficients held in 48 registers across the views - spills.)
static constexpr int kPreViews = Ctx::kMaxFrames;
typedef float nt_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void nt_store4(float4* p, const float4 v) {
    nt_f4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<nt_f4*>(p));
}
struct PreMultiArgs {
    ViewUniforms u[kPreViews];
    float cam[kPreViews][3];
    int* depth[kPreViews];
    int* pre[kPreViews];
    float4* rec[kPreViews];
    uint32_t* rect[kPreViews];
    int nv;
};

template <int NV>
__global__ __launch_bounds__(kRB) void pre_multi_kernel(const uint4* __restrict__ tex, long long n, const float* __restrict__ sh_coef,
                                                         int sh_deg, const PreMultiArgs* __restrict__ ap) {
    // (the arguments live in device memory: the views' uniforms are scalar loads)
    const PreMultiArgs& a = *ap;
    __shared__ int slo[4], shi[4];
    int lo[NV], hi[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) lo[v] = 2147483647, hi[v] = -2147483647 - 1;
    const int K = (sh_deg + 1) * (sh_deg + 1);
    for (long long i = (long long)blockIdx.x * kRB + threadIdx.x; i < n; i += (long long)gridDim.x * kRB) {
        const uint4 t0 = tex[2 * i], t1 = tex[2 * i + 1];
        const float px = __uint_as_float(t0.x), py = __uint_as_float(t0.y), pz = __uint_as_float(t0.z);
        // ---- geometry of every view: depth key, rectangle and the first record go out at once; what the colour needs stays
        float g1x[NV], g1y[NV], fade[NV];
        unsigned want = 0;  // bit v: view v wants a colour for this splat
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            // (the pointer is laundered so that a view's 44 uniforms are scalar loads HERE, dead after its vertex shader: hoisted out
            // of the splat loop as invariants, the uniforms of four views are 375 spilled SGPRs)
            const PreMultiArgs* av = ap;
            asm volatile("" : "+s"(av));
            const PreGeom g = pre_geom(t0, t1, av->u[v]);
            __builtin_nontemporal_store(g.depth, &a.depth[v][i]);  // streaming stores: the records are not read again here
            lo[v] = min(lo[v], g.depth);
            hi[v] = max(hi[v], g.depth);
            // (ordinary stores: the three 16-byte pieces of a record meet in the L2 before they leave it; a splat without a
            // rectangle - two in five on the bench scene - is in no list and writes none: only splat 0's is read regardless, by the epilogue)
            if (g.rect != kEmptyRect || i == 0) a.rec[v][3 * i] = g.r0;
            __builtin_nontemporal_store(g.rect, &a.rect[v][i]);
            if (i == 0) a.pre[v][2] = (int)g.rect;
            g1x[v] = g.g1x, g1y[v] = g.g1y, fade[v] = g.fade;
            want |= g.colour ? 1u << v : 0u;
            __builtin_amdgcn_sched_barrier(0);  // one view after the other: interleaved, the six vertex shaders need 256 VGPRs
        }
        // ---- colour: the coefficients are read ONCE (if any view draws the splat) and every view accumulates its own sum, in
        // sh_color's order: acc = b_0 p_0, then acc += b_k p_k for k = 1 .. K-1, the basis recomputed from the view's direction
        float acc[NV][3], dx[NV], dy[NV], dz[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            acc[v][0] = acc[v][1] = acc[v][2] = 0.f;
            sh_dir(px, py, pz, a.cam[v][0], a.cam[v][1], a.cam[v][2], dx[v], dy[v], dz[v]);
        }
        if (sh_coef != nullptr && want != 0u) {
#pragma unroll
            for (int k = 0; k < 16; ++k) {
                if (k < K) {
                    float p[3];
#pragma unroll
                    for (int c = 0; c < 3; ++c) p[c] = sh_coef[((size_t)k * 3 + c) * (size_t)n + (size_t)i];
#pragma unroll
                    for (int v = 0; v < NV; ++v) {
                        const float b = sh_basis(k, dx[v], dy[v], dz[v]);
#pragma unroll
                        for (int c = 0; c < 3; ++c) acc[v][c] = k == 0 ? b * p[c] : acc[v][c] + b * p[c];
                    }
                }
            }
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            float4 r1 = make_float4(0.f, 0.f, 0.f, 0.f);
            float2 r2 = make_float2(0.f, 0.f);
            if ((want >> v) & 1u) {
                float col[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) col[k] = rgba8_channel(t1, k, fade[v]);
                if (sh_coef != nullptr) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) col[c] = fade[v] * sh_clamp(acc[v][c]);
                }
                r1 = make_float4(g1x[v], g1y[v], col[0], col[1]);
                r2 = make_float2(col[2], col[3]);
            }
            if (((want >> v) & 1u) || i == 0) {  // (want <=> the view gave the splat a rectangle)
                a.rec[v][3 * i + 1] = r1;
                a.rec[v][3 * i + 2] = make_float4(r2.x, r2.y, 0.f, 0.f);
            }
        }
    }
#pragma unroll
    for (int v = 0; v < NV; ++v) depth_range_to(lo[v], hi[v], a.pre[v], slo, shi);
}

// 16-bit depth bucket (gs.js:443-447) once the depth range is known; doubles as the (key, value) initialisation of the
// level-1 sort.  The JS's typed arrays silently drop a bucket outside [0, 65535] (the farthest splat always gets
// 65536): such a splat is never drawn, it sorts to the end and loses its tile rectangle here.
__global__ __launch_bounds__(kRB) void bucket_kernel(const int* __restrict__ depth, long long n, const int* __restrict__ minmax,
                                                      uint32_t* __restrict__ bucket, uint32_t* __restrict__ key,
                                                      uint32_t* __restrict__ idx, uint32_t* __restrict__ tile_rect,
                                                      int* __restrict__ dropped, int compact) {
    const long long i = (long long)blockIdx.x * kRB + threadIdx.x;
    bool drop = false;
    if (i < n) {
        const double minDepth = (double)minmax[0], maxDepth = (double)minmax[1];
        const double depthInv = (256.0 * 256.0) / (maxDepth - minDepth);
        const int b = js_toint32(((double)depth[i] - minDepth) * depthInv);
        const bool in_range = b >= 0 && b < 65536;
        const uint32_t v = in_range ? (uint32_t)b : 65536u;
        bucket[i] = v;
        // 16-bit sort key: a dropped splat covers no tile, where it sorts does not matter.  compact: a splat that covers no tile
        // (dropped, or without a rectangle in this view: behind the camera, off the frame, between the pixel centres) gets the
        // key the level-1 sort leaves out (radix_sort_pairs_drop) - the bin kernels then see the others only, in their order
        const bool seen = in_range && tile_rect[i] != kEmptyRect;
        key[i] = compact ? (seen ? v : 0xffffffffu) : (in_range ? v : 65535u);
        idx[i] = (uint32_t)i;
        drop = !in_range;
        if (drop) tile_rect[i] = kEmptyRect;
    }
    const unsigned long long m = __ballot(drop);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(dropped, (int)__popcll(m));
}

// ---- exclusive scan of u32 (2 launches: tile sums; downsweep, every workgroup adding up the sums before its own) -----
static constexpr int kScanItems = 16;
static constexpr int kScanTile = kRB * kScanItems;

__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* wsum /*[4]*/, uint32_t& total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(inc, o);
        if (lane >= o) inc += up;
    }
    if (lane == 63) wsum[wave] = inc;
    __syncthreads();
    uint32_t off = 0;
    for (int w = 0; w < wave; ++w) off += wsum[w];
    total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    return off + inc - v;
}

__global__ __launch_bounds__(kRB) void scan_sums_kernel(const uint32_t* __restrict__ in, long long n,
                                                         uint32_t* __restrict__ sums) {
    __shared__ uint32_t wsum[4];
    const long long base = (long long)blockIdx.x * kScanTile + (long long)threadIdx.x * kScanItems;
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k)
        if (base + k < n) s += in[base + k];
    uint32_t total;
    (void)block_exclusive_scan(s, wsum, total);
    if (threadIdx.x == 0) sums[blockIdx.x] = total;
}

// sums: one u32 per workgroup (a few hundred): each workgroup re-adds the ones before it instead of a third launch.
// The last workgroup also writes the grand total (64 bit: the pair count of a close-up view can pass 2^32).
__global__ __launch_bounds__(kRB) void scan_down_kernel(const uint32_t* __restrict__ in, long long n,
                                                         const uint32_t* __restrict__ sums, uint32_t* __restrict__ out,
                                                         unsigned long long* __restrict__ grand) {
    __shared__ uint32_t wsum[4];
    __shared__ unsigned long long wbase[4];
    unsigned long long before = 0;
    for (int t = threadIdx.x; t < (int)blockIdx.x; t += kRB) before += sums[t];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o);
    if ((threadIdx.x & 63) == 0) wbase[threadIdx.x >> 6] = before;
    __syncthreads();
    const unsigned long long block_base = wbase[0] + wbase[1] + wbase[2] + wbase[3];
    const long long base = (long long)blockIdx.x * kScanTile + (long long)threadIdx.x * kScanItems;
    uint32_t v[kScanItems];
    uint32_t s = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        v[k] = base + k < n ? in[base + k] : 0u;
        s += v[k];
    }
    uint32_t total;
    uint32_t run = (uint32_t)block_base + block_exclusive_scan(s, wsum, total);  // offsets are only used when the total fits 31 bits
#pragma unroll
    for (int k = 0; k < kScanItems; ++k) {
        if (base + k < n) out[base + k] = run;
        run += v[k];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) *grand = block_base + total;
}

// ---- binning ---------------------------------------------------------------------------------------------------------
// The reference orders ALL splats by (bucket, index) (stable counting sort, gs.js:450-457).  Level 1 sorts the n splats
// by bucket (16 bits, stable); level 2 emits their (tile, splat) pairs in that order and sorts the pairs by tile only
// (stable): inside a tile the pairs keep the (bucket, index) order.
//
// Wave-cooperative binning.  The 64 splats of a wave (consecutive in depth order) span a list of CANDIDATE tiles: the
// concatenation of their tile rectangles.  The lanes walk that list 64 candidates at a time: a lane finds the splat that
// owns its candidate by a binary search over the wave's candidate offsets (shuffles), derives the tile from the
// candidate's rank inside the rectangle and tests it -
//    * exact: the ellipse |vPosition| <= 2 really reaches the tile (tile_touches), not just its bounding box;
//    * skip:  the tile is not yet opaque (`sat`, written by the blend kernel of the previous depth phase) -
// and one ballot turns the 64 verdicts into a mask from which every splat counts its kept tiles (COUNT) or every kept
// candidate derives its slot in the pair arrays (EMIT: splat's offset + kept candidates of the same splat before it).
// No lane ever loops over a rectangle on its own: a splat covering 4000 tiles costs the wave 63 rounds, not one lane
// 4000, which is what made the exact test a net loss when every thread walked its own rectangle.
// COUNT and EMIT run the same tests on the same data (sat is not written in between), so the slots are exact.
//
// BIN32 (option render_bin32): the candidates are 32x32-pixel BINS (2x2 tiles) instead of tiles.  A pair's key is the bin,
// its value carries in bits 28-31 the mask of the bin's tiles (bit = 2 * (ty & 1) + (tx & 1)) that the splat's rectangle
// covers and that are not opaque yet; the blend kernel of a tile walks its bin's list and takes the entries whose mask names
// it - in list order, so a tile blends exactly the splats, in exactly the order, of the per-tile lists.  A splat's rectangle
// spans about half as many bins as tiles: half the pairs to emit, sort and range.  `tiles_x` is then the bins per row and
// `sat` holds four bytes per bin (bin * 4 + bit), read as one word.
template <bool EMIT, bool BIN32>
__global__ __launch_bounds__(kRB) void bin_kernel(const unsigned long long* __restrict__ nvis_dev, long long div0, long long div1,
                                                   long long m_cap, const uint32_t* __restrict__ by_depth,
                                                   const uint32_t* __restrict__ tile_rect, const float4* __restrict__ rec,
                                                   float H, int tiles_x, int exact,
                                                   const uint8_t* __restrict__ sat, uint32_t* __restrict__ count_out,
                                                   const uint32_t* __restrict__ offset, uint32_t* __restrict__ keys,
                                                   uint32_t* __restrict__ vals, const unsigned long long* __restrict__ total_dev,
                                                   unsigned long long cap, uint32_t* __restrict__ rect_seq) {
    if (EMIT && *total_dev > cap) return;  // the phase does not fit the pair buffers: the host redoes the frame with larger ones
    // the phase's splats: [nvis / div0, nvis / div1) of the depth order, nvis = the splats the level-1 sort kept (on the device:
    // the host never waits for it; the launch covers m_cap >= the phase's length, and COUNT writes every slot of it)
    const long long nvis = (long long)*nvis_dev;
    const long long j0 = div0 ? nvis / div0 : 0, j1 = nvis / div1;
    const long long rel = (long long)blockIdx.x * kRB + threadIdx.x;
    const long long j = j0 + rel;
    const int lane = threadIdx.x & 63;
    const bool live = j < j1;
    const bool slot = rel < m_cap;
    const uint32_t i = live ? by_depth[j] : 0u;
    // COUNT gathers the splat's rectangle (a 4-byte read that costs a whole line) and leaves it in the phase's own order for
    // EMIT (count_out / offset and rect_seq are per-phase arrays of m_cap slots)
    const uint32_t rect = live ? (EMIT ? rect_seq[rel] : tile_rect[i]) : kEmptyRect;
    if (!EMIT && slot) rect_seq[rel] = rect;
    const uint32_t tx0 = rect & 255u, tx1 = (rect >> 8) & 255u, ty0 = (rect >> 16) & 255u, ty1 = rect >> 24;
    constexpr uint32_t kSh = BIN32 ? 1u : 0u;  // candidate = bin: the rectangle in bin units
    const uint32_t w = tx1 >= tx0 ? (tx1 >> kSh) - (tx0 >> kSh) + 1u : 0u;
    const uint32_t area = w * (ty1 >= ty0 ? (ty1 >> kSh) - (ty0 >> kSh) + 1u : 0u);
    // candidate offsets inside the wave: exclusive scan of the areas
    uint32_t inc = area;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t up = __shfl_up(inc, o);
        if (lane >= o) inc += up;
    }
    const uint32_t total = __shfl(inc, 63);
    if (total == 0) {  // wave-uniform
        if (!EMIT && slot) count_out[rel] = 0u;
        return;
    }
    const uint32_t cstart = inc - area;
    if (!EMIT && !exact && !sat) {  // wave-uniform: nothing to test, every tile of the rectangle is kept
        if (slot) count_out[rel] = area;  // (a slot past the phase's end: area 0)
        return;
    }
    const bool test = exact && area > 1u;  // a one-tile rectangle holds the centre's tile or touches it by construction
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0;
    if (test) {
        r0 = rec[3 * (size_t)i];
        r1 = rec[3 * (size_t)i + 1];
    }
    const uint32_t my_off = EMIT && live ? offset[rel] : 0u;
    uint32_t kept_run = 0;  // kept candidates of MY splat in the rounds so far
    // every lane runs every round: the shuffles read registers of ALL lanes; only the stores are predicated
    for (uint32_t base = 0; base < total; base += 64u) {
        const uint32_t e = base + (uint32_t)lane;
        const bool valid = e < total;
        const uint32_t target = valid ? e : total - 1u;
        int lo = 0, hi = 63;  // the LAST lane whose candidates start at or before `target` (empty splats share their
                              // start with the next non-empty one, which is the later lane)
#pragma unroll
        for (int step = 0; step < 6; ++step) {
            const int mid = (lo + hi + 1) >> 1;
            const uint32_t om = __shfl(cstart, mid);
            if (om <= target) lo = mid;
            else hi = mid - 1;
        }
        const uint32_t s_start = __shfl(cstart, lo);
        const uint32_t s_rect = __shfl(rect, lo);
        const uint32_t s_tx0 = s_rect & 255u, s_tx1 = (s_rect >> 8) & 255u, s_ty0 = (s_rect >> 16) & 255u;
        const uint32_t s_w = (s_tx1 >> kSh) - (s_tx0 >> kSh) + 1u;
        const uint32_t local = target - s_start;
        const uint32_t ry = local / s_w, rx = local - ry * s_w;
        const uint32_t tx = (s_tx0 >> kSh) + rx, ty = (s_ty0 >> kSh) + ry;
        const uint32_t tile = ty * (uint32_t)tiles_x + tx;
        bool keep = valid;
        uint32_t tag = 0u;  // BIN32: the mask of the bin's tiles, in the value's top four bits
        if (BIN32) {
            const uint32_t s_ty1 = s_rect >> 24;
            const uint32_t cx = 2u * tx, cy = 2u * ty;
            const uint32_t xb = (cx >= s_tx0 && cx <= s_tx1 ? 1u : 0u) | (cx + 1u >= s_tx0 && cx + 1u <= s_tx1 ? 2u : 0u);
            uint32_t m = (cy >= s_ty0 && cy <= s_ty1 ? xb : 0u) | (cy + 1u >= s_ty0 && cy + 1u <= s_ty1 ? xb << 2 : 0u);
            if (sat && keep) {
                const uint32_t sw = reinterpret_cast<const uint32_t*>(sat)[tile];
                m &= ~((sw & 0xffu ? 1u : 0u) | (sw & 0xff00u ? 2u : 0u) | (sw & 0xff0000u ? 4u : 0u) | (sw & 0xff000000u ? 8u : 0u));
            }
            keep = keep && m != 0u;
            tag = m << 28;
        }
        if (exact) {  // wave-uniform
            const int s_test = __shfl((int)test, lo);
            const float a0 = __shfl(r0.x, lo), a1 = __shfl(r0.y, lo), a2 = __shfl(r0.z, lo), a3 = __shfl(r0.w, lo);
            const float b0 = __shfl(r1.x, lo), b1 = __shfl(r1.y, lo);
            if (keep && s_test) keep = tile_touches(a0, a1, a2, a3, b0, b1, H, tx, ty);
        }
        if (!BIN32 && sat && keep) keep = sat[tile] == 0;
        if (EMIT && !exact && !sat) {  // wave-uniform fast path: slot = the splat's offset + the candidate's rank in its rectangle
            const uint32_t pos = __shfl(my_off, lo) + local;  // shuffles outside the predicate: every lane must take part
            const uint32_t s_i = __shfl(i, lo);
            if (keep) {
                keys[pos] = tile;
                vals[pos] = s_i | tag;
            }
            continue;
        }
        const unsigned long long verdict = __ballot(keep);
        // bits of this round that belong to my splat: candidates [cstart, cstart + area) shifted by base
        const long long mlo = (long long)cstart - (long long)base, mhi = mlo + (long long)area;
        const int blo = (int)(mlo < 0 ? 0 : (mlo > 64 ? 64 : mlo)), bhi = (int)(mhi < 0 ? 0 : (mhi > 64 ? 64 : mhi));
        const unsigned long long below_hi = bhi >= 64 ? ~0ull : ((1ull << bhi) - 1ull);
        const unsigned long long below_lo = blo >= 64 ? ~0ull : ((1ull << blo) - 1ull);
        const uint32_t mine = (uint32_t)__popcll(verdict & below_hi & ~below_lo);
        if (EMIT) {
            const uint32_t s_prior = __shfl(kept_run, lo);   // kept candidates of the owner in earlier rounds
            const uint32_t s_off = __shfl(my_off, lo);
            const uint32_t s_i = __shfl(i, lo);
            const long long slo = (long long)s_start - (long long)base;
            const int sblo = (int)(slo < 0 ? 0 : slo);        // first bit of the owner in this round (<= lane)
            const unsigned long long upto = (1ull << lane) - 1ull;
            const unsigned long long from = sblo >= 64 ? ~0ull : ((1ull << sblo) - 1ull);
            const uint32_t rank = s_prior + (uint32_t)__popcll(verdict & upto & ~from);
            if (keep) {
                keys[s_off + rank] = tile;
                vals[s_off + rank] = s_i | tag;
            }
        }
        kept_run += mine;
    }
    if (!EMIT && slot) count_out[rel] = kept_run;
}

__global__ __launch_bounds__(kRB) void ranges_kernel(const uint32_t* __restrict__ keys, const unsigned long long* __restrict__ total,
                                                      unsigned long long cap, int ntiles, int2* __restrict__ ranges) {
    const unsigned long long tot = *total;
    const long long P = tot > cap ? 0 : (long long)tot;
    const long long p = (long long)blockIdx.x * kRB + threadIdx.x;
    if (p >= P) return;
    const uint32_t t = keys[p];
    if (t >= (uint32_t)ntiles) return;  // defensive: a key outside the frame must never index the table
    if (p == 0 || keys[p - 1] != t) ranges[t].x = (int)p;
    if (p == P - 1 || keys[p + 1] != t) ranges[t].y = (int)(p + 1);
}

// ---- host drivers -----------------------------------------------------------------------------------------
static inline unsigned grid_for(long long n) { return (unsigned)((n + kRB - 1) / kRB); }

static int exclusive_scan_u32(Ctx* c, const uint32_t* in, uint32_t* out, long long n, unsigned long long* grand_dev) {
    const int m = (int)((n + kScanTile - 1) / kScanTile);
    GSX_HIP(c, c->r_scan.ensure(sizeof(uint32_t) * (size_t)(m + 1)));
    uint32_t* sums = c->r_scan.as<uint32_t>();
    {
        ProfScope ps(c, "scan");
        hipLaunchKernelGGL(scan_sums_kernel, dim3(m), dim3(kRB), 0, c->stream, in, n, sums);
        hipLaunchKernelGGL(scan_down_kernel, dim3(m), dim3(kRB), 0, c->stream, in, n, sums, out, grand_dev);
    }
    GSX_HIP(c, hipGetLastError());
    return GSX_OK;
}

int upload_splats(Ctx* c, int64_t n, const float* xyz, const float* scale, const float* rot, const float* opacity,
                  const float* f_dc, const int32_t* labels) {
    GSX_HIP(c, hipSetDevice(c->device));
    c->rn = 0;
    if (n < 0 || (n > 0 && (!xyz || !f_dc || (scale && (!rot || !opacity)))))
        return fail(c, GSX_E_INVALID, "upload_splats: xyz and f_dc are required; scale needs rot and opacity");
    if (n > ((int64_t)1 << 30)) return fail(c, GSX_E_UNSUPPORTED, "upload_splats: n > 2^30");
    if (n == 0) return GSX_OK;
    DevBuf dxyz, dscale, drot, dop, dlab, key0, key1, idx0, idx1;
    DevBuf& ddc = c->r_fdc;  // kept (source order): gsx_upload_sh re-lays it with f_rest
    c->r_sh_deg = 0;
    auto up = [&](DevBuf& b, const void* src, size_t bytes) -> hipError_t {
        hipError_t e = b.ensure(bytes);
        if (e != hipSuccess) return e;
        return hipMemcpyAsync(b.p, src, bytes, hipMemcpyHostToDevice, c->stream);
    };
    GSX_HIP(c, up(dxyz, xyz, sizeof(float) * 3 * n));
    GSX_HIP(c, up(ddc, f_dc, sizeof(float) * 3 * n));
    if (scale) {
        GSX_HIP(c, up(dscale, scale, sizeof(float) * 3 * n));
        GSX_HIP(c, up(drot, rot, sizeof(float) * 4 * n));
    }
    if (opacity) GSX_HIP(c, up(dop, opacity, sizeof(float) * n));
    if (labels) GSX_HIP(c, up(dlab, labels, sizeof(int32_t) * n));
    GSX_HIP(c, key0.ensure(4 * n));
    GSX_HIP(c, key1.ensure(4 * n));
    GSX_HIP(c, idx0.ensure(4 * n));
    GSX_HIP(c, idx1.ensure(4 * n));
    GSX_HIP(c, c->r_order.ensure(4 * n));
    GSX_HIP(c, c->r_buffer.ensure(32 * n));
    GSX_HIP(c, c->r_tex.ensure(32 * n));
    int where = 0;
    if (scale) {
        hipLaunchKernelGGL(splat_importance_kernel, dim3(grid_for(n)), dim3(kRB), 0, c->stream, dscale.as<float>(),
                           dop.as<float>(), (long long)n, key0.as<uint32_t>(), idx0.as<uint32_t>());
        GSX_HIP(c, hipGetLastError());
        int rc = radix_sort_pairs(c, key0.as<uint32_t>(), idx0.as<uint32_t>(), key1.as<uint32_t>(), idx1.as<uint32_t>(), n, 32,
                                  &where);
        if (rc) return rc;
    } else {
        hipLaunchKernelGGL(iota_kernel, dim3(grid_for(n)), dim3(kRB), 0, c->stream, idx0.as<uint32_t>(), (long long)n);
    }
    GSX_HIP(c, hipMemcpyAsync(c->r_order.p, where ? idx1.p : idx0.p, 4 * n, hipMemcpyDeviceToDevice, c->stream));
    {
        ProfScope ps(c, "splat_pack");
        hipLaunchKernelGGL(splat_pack_kernel, dim3(grid_for(n)), dim3(kRB), 0, c->stream, c->r_order.as<uint32_t>(),
                           (long long)n, dxyz.as<float>(), scale ? dscale.as<float>() : nullptr,
                           scale ? drot.as<float>() : nullptr, opacity ? dop.as<float>() : nullptr, ddc.as<float>(),
                           labels ? dlab.as<int>() : nullptr, c->r_buffer.as<uint4>(), c->r_tex.as<uint4>());
    }
    GSX_HIP(c, hipGetLastError());
    GSX_HIP(c, hipStreamSynchronize(c->stream));
    c->rn = n;
    c->r_sh_on = false;
    return GSX_OK;
}

// gs.js:81-107 and 66-79, fp64 exactly as the JavaScript evaluates them
void js_view_matrix(const gsx_camera* cam, double out[16]) {
    const double* R = cam->R;
    const double* p = cam->p;
    for (int i = 0; i < 3; ++i) {
        out[4 * i + 0] = R[3 * i + 0];
        out[4 * i + 1] = R[3 * i + 1];
        out[4 * i + 2] = R[3 * i + 2];
        out[4 * i + 3] = 0.0;
    }
    for (int i = 0; i < 3; ++i) out[12 + i] = -p[0] * R[i] - p[1] * R[i + 3] - p[2] * R[i + 6];
    out[15] = 1.0;
}

void js_proj_matrix(double fx, double fy, double width, double height, double out[16]) {
    const double Z_FAR = 200.0, Z_NEAR = 0.2;
    const double zRange = Z_FAR - Z_NEAR;
    for (int k = 0; k < 16; ++k) out[k] = 0.0;
    out[0] = (2 * fx) / width;
    out[5] = -(2 * fy) / height;
    out[10] = Z_FAR / zRange;
    out[11] = 1.0;
    out[14] = -(Z_FAR * Z_NEAR) / zRange;
}

void js_multiply4(const double A[16], const double B[16], double out[16]) {
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            out[4 * i + j] = B[4 * i] * A[j] + B[4 * i + 1] * A[j + 4] + B[4 * i + 2] * A[j + 8] + B[4 * i + 3] * A[j + 12];
}

int upload_sh(Ctx* c, const float* f_rest, int deg) {
    GSX_HIP(c, hipSetDevice(c->device));
    if (deg < 0 || deg > 3) return fail(c, GSX_E_INVALID, "upload_sh: degree %d outside [0,3]", deg);
    if (c->rn == 0) return fail(c, GSX_E_STATE, "upload_sh before upload_splats");
    if (deg > 0 && !f_rest) return fail(c, GSX_E_INVALID, "upload_sh: f_rest is NULL");
    const long long n = c->rn;
    const int K = (deg + 1) * (deg + 1);
    DevBuf drest;
    if (K > 1) {
        GSX_HIP(c, drest.ensure(sizeof(float) * 3 * (size_t)(K - 1) * n));
        GSX_HIP(c, hipMemcpyAsync(drest.p, f_rest, sizeof(float) * 3 * (size_t)(K - 1) * n, hipMemcpyHostToDevice, c->stream));
    }
    GSX_HIP(c, c->r_shc.ensure(sizeof(float) * 3 * (size_t)K * n));
    hipLaunchKernelGGL(sh_pack_kernel, dim3(grid_for(n)), dim3(kRB), 0, c->stream, c->r_order.as<uint32_t>(), n,
                       c->r_fdc.as<float>(), K > 1 ? drest.as<float>() : nullptr, K, c->r_shc.as<float>());
    GSX_HIP(c, hipGetLastError());
    GSX_HIP(c, hipStreamSynchronize(c->stream));
    c->r_sh_deg = deg;
    c->r_sh_on = true;
    return GSX_OK;
}

int launch_blend(Ctx* c, const uint32_t* vals, int W, int H, int tiles_x, int tiles_y, const int* dropped_dev,
                 unsigned long long* consumed_dev, uint8_t* sat, int first_phase, int last_phase);  // blend.hip

static const int kPreInit[4] = {2147483647, -2147483647 - 1, (int)kEmptyRect, 0};  // depth min / max, splat 0's rectangle (static: outlives the copy)

static int ensure_pre_set(Ctx* c, Ctx::PreSet& ps, long long n) {
    const size_t n4 = 4 * (size_t)n;
    GSX_HIP(c, ps.depth.ensure(n4));
    GSX_HIP(c, ps.rect.ensure(n4));
    GSX_HIP(c, ps.rec.ensure(48 * (size_t)n));
    GSX_HIP(c, ps.pre.ensure(16));
    return GSX_OK;
}

static ViewUniforms view_uniforms(const gsx_camera* cam, int W, int H) {
    ViewUniforms u{};
    double view[16], proj[16], vp[16];
    js_view_matrix(cam, view);
    js_proj_matrix(cam->fx, cam->fy, (double)W, (double)H, proj);
    js_multiply4(proj, view, vp);
    u.vp2 = vp[2];
    u.vp6 = vp[6];
    u.vp10 = vp[10];
    for (int k = 0; k < 16; ++k) {  // gl.uniformMatrix4fv: JS numbers -> f32
        u.view[k] = (float)view[k];
        u.proj[k] = (float)proj[k];
    }
    u.fx = (float)cam->fx;
    u.fy = (float)cam->fy;
    u.W = (float)W;
    u.H = (float)H;
    u.tiles_x = (W + 15) / 16;
    u.tiles_y = (H + 15) / 16;
    return u;
}

// One frame.  Per view: pre_kernel (depth keys, vertex shader, colours, tile rectangles), bucket_kernel, the level-1
// sort of the splats by depth bucket; then the splats are rasterized FRONT TO BACK IN DEPTH PHASES (the nearest
// n/r^(K-1) splats, the next ones up to n/r^(K-2), .., the rest): each phase bins its splats into the tiles that are not
// opaque yet (bin_kernel COUNT -> scan -> EMIT), sorts its (tile, splat) pairs by tile, and blends them on top of the
// image, marking the tiles whose every pixel has reached 1 - alpha < 1e-5.  A dense scene sorts a few times the pairs
// the blend really consumes instead of all of them (3 M splats at 1080p: 22 M pairs bounded by the boxes, 2.2 M
// consumed); what is skipped could not have changed any channel by more than 1e-5 (the remaining transmittance bounds
// the sum of everything behind it).
// The host never waits inside a frame: the pair count of a phase stays on the device, the sort / ranges kernels are
// launched for the CAPACITY of the pair buffers and read the count themselves.  Only the end-of-frame read-back tells
// the host whether a phase overflowed the buffers; then they are enlarged and the frame is redone (first frame of a
// scene, or a much closer view).
int render_view(Ctx* c, const gsx_camera* cam, int W, int H, float* rgba_out) {
    GSX_HIP(c, hipSetDevice(c->device));
    if (!cam || W < 1 || H < 1) return fail(c, GSX_E_INVALID, "render_view: bad arguments");
    const int tiles_x = (W + 15) / 16, tiles_y = (H + 15) / 16;
    if (tiles_x > 256 || tiles_y > 256)
        return fail(c, GSX_E_UNSUPPORTED, "render_view: %dx%d exceeds 4096 pixels per side", W, H);
    const size_t img_bytes = sizeof(float) * 4 * (size_t)W * H;
    GSX_HIP(c, c->r_image.ensure(img_bytes));
    c->r_W = W;
    c->r_H = H;
    const int ntiles = tiles_x * tiles_y;
    constexpr int kMaxPhases = 8;
    const long long n = c->rn;
    // 32x32-pixel bins (see bin_kernel): the one-wave blend kernel reads the masks; a value's top four bits are the mask
    const bool bin32 = c->opt_render_bin32 && c->opt_blend_pk2 == 2 && !c->opt_exact_cull && n < (1ll << 28);
    c->r_bin32 = bin32;
    const int bins_x = (tiles_x + 1) / 2, bins_y = (tiles_y + 1) / 2;
    const int nlists = bin32 ? bins_x * bins_y : ntiles;          // lists the pairs are sorted into
    const int lists_x = bin32 ? bins_x : tiles_x;
    const auto count_k = bin32 ? bin_kernel<false, true> : bin_kernel<false, false>;
    const auto emit_k = bin32 ? bin_kernel<true, true> : bin_kernel<true, false>;
    const size_t sat_bytes = bin32 ? 4 * (size_t)nlists : (size_t)ntiles;
    GSX_HIP(c, c->r_ranges.ensure(sizeof(int2) * (size_t)nlists));
    GSX_HIP(c, c->r_sat.ensure(sat_bytes));
    constexpr size_t kSmallBytes = 256 + (size_t)kConsumedSlots * 128;
    GSX_HIP(c, c->r_small.ensure(kSmallBytes));
    // [2]=dropped, then u64: [3 + p]=pairs of phase p; from byte 256: pairs consumed, kConsumedSlots
    // counters 128 B apart (summed below)
    int* small = c->r_small.as<int>();
    unsigned long long* small64 = reinterpret_cast<unsigned long long*>(small);
    unsigned long long* consumed_dev = small64 + 32;
    unsigned long long* pairs_dev = small64 + 3;
    unsigned long long* nvis_dev = small64 + 12;  // splats the level-1 sort kept (those with a rectangle in this view)
    c->r_P = 0;
    c->r_consumed = 0;
    if (n <= 0) {  // no splats at all: the cleared canvas
        GSX_HIP(c, hipMemsetAsync(c->r_image.p, 0, img_bytes, c->stream));
        if (rgba_out) GSX_HIP(c, hipMemcpyAsync(rgba_out, c->r_image.p, img_bytes, hipMemcpyDeviceToHost, c->stream));
        GSX_HIP(c, hipStreamSynchronize(c->stream));
        return GSX_OK;
    }
    const ViewUniforms u = view_uniforms(cam, W, H);
    const size_t n4 = 4 * (size_t)n;
    const bool ext_pre = c->r_pre_ext >= 0;  // gsx_render_views ran this frame's pre pass (pre_multi_kernel) into one of the rotating sets
    {
        Ctx::PreSet& ps = c->r_sets[ext_pre ? c->r_pre_ext : 0];  // a frame on its own uses set 0
        int rcs = ensure_pre_set(c, ps, n);
        if (rcs) return rcs;
        c->r_depth.alias(ps.depth);
        c->r_rect.alias(ps.rect);
        c->r_rec.alias(ps.rec);
        c->r_pre.alias(ps.pre);
    }
    GSX_HIP(c, c->r_bucket.ensure(n4));
    GSX_HIP(c, c->r_count.ensure(n4));
    GSX_HIP(c, c->r_offset.ensure(n4));
    GSX_HIP(c, c->r_d0.ensure(n4));
    GSX_HIP(c, c->r_d1.ensure(n4));
    GSX_HIP(c, c->r_d2.ensure(n4));
    GSX_HIP(c, c->r_d3.ensure(n4));
    GSX_HIP(c, c->r_rects.ensure(n4));
    // depth phases: boundaries nvis / r^(K-1), nvis / r^(K-2), .., nvis / r, nvis of the depth order, nvis = the splats the
    // level-1 sort keeps (on the device); the host sizes a phase's launches for n instead
    const int K = std::max(1, std::min(c->opt_render_phases, kMaxPhases));
    const long long ratio = std::max(2, std::min(c->opt_render_phase_ratio, 64));
    long long divs[kMaxPhases + 1];  // phase p = [nvis / divs[p], nvis / divs[p + 1]); divs[0] = 0 stands for the start
    divs[0] = 0;
    for (int p = 1; p <= K; ++p) {
        long long div = 1;
        for (int q = p; q < K; ++q) div = std::min<long long>(div * ratio, (long long)1 << 40);
        divs[p] = div;
    }
    int tile_bits = 1;
    while ((1 << tile_bits) < nlists) ++tile_bits;
    // the pair sorts in one pass (2040 bins at 1080p; 4K: 8160, two passes) - for a frame on its own: 6 % less kernel time; with
    // several frames in flight its scattered stores cost the other streams more than the saved launches give (option value 2: always)
    const bool wide = tile_bits <= 11 && (c->opt_render_wide_sort == 2 || (c->opt_render_wide_sort == 1 && !c->r_in_flight));
    if (c->r_pair_cap == 0) c->r_pair_cap = std::max<size_t>((size_t)1 << 20, 2 * (size_t)n);

    for (int attempt = 0;; ++attempt) {
        const size_t cap = c->r_pair_cap;
        GSX_HIP(c, c->r_keys0.ensure(4 * cap));
        GSX_HIP(c, c->r_keys1.ensure(4 * cap));
        GSX_HIP(c, c->r_vals0.ensure(4 * cap));
        GSX_HIP(c, c->r_vals1.ensure(4 * cap));
        GSX_HIP(c, hipMemsetAsync(c->r_sat.p, 0, sat_bytes, c->stream));
        GSX_HIP(c, hipMemsetAsync(small, 0, kSmallBytes, c->stream));  // counters
        if (!ext_pre) {  // (a frame that is redone because a phase overflowed the pair buffers keeps the pre pass's records)
            GSX_HIP(c, hipMemcpyAsync(c->r_pre.p, kPreInit, sizeof kPreInit, hipMemcpyHostToDevice, c->stream));
            ProfScope ps(c, "render_pre");
            hipLaunchKernelGGL(pre_kernel, dim3(std::min<unsigned>(grid_for(n), 2048u)), dim3(kRB), 0, c->stream, c->r_tex.as<uint4>(), n, u,
                               c->r_sh_on ? c->r_shc.as<float>() : nullptr, c->r_sh_deg, (float)cam->p[0], (float)cam->p[1],
                               (float)cam->p[2], c->r_depth.as<int>(), c->r_pre.as<int>(), c->r_rec.as<float4>(),
                               c->r_rect.as<uint32_t>());
        }
        {
            ProfScope ps(c, "render_bucket");
            hipLaunchKernelGGL(bucket_kernel, dim3(grid_for(n)), dim3(kRB), 0, c->stream, c->r_depth.as<int>(), n, c->r_pre.as<int>(),
                               c->r_bucket.as<uint32_t>(), c->r_d0.as<uint32_t>(), c->r_d1.as<uint32_t>(), c->r_rect.as<uint32_t>(),
                               small + 2, c->opt_render_compact);
        }
        GSX_HIP(c, hipGetLastError());
        // level 1: splats by depth bucket (stable)
        int dwhere = 0;
        // (its first pass leaves out the splats bucket_kernel marked: nvis_dev = the ones that remain, sorted, in front)
        int rc = radix_sort_pairs_drop(c, c->r_d0.as<uint32_t>(), c->r_d1.as<uint32_t>(), c->r_d2.as<uint32_t>(), c->r_d3.as<uint32_t>(),
                                       n, nullptr, 16, &dwhere, nvis_dev);
        if (rc) return rc;
        const uint32_t* by_depth = dwhere ? c->r_d3.as<uint32_t>() : c->r_d1.as<uint32_t>();
        int first_phase = 1;
        for (int p = 0; p < K; ++p) {
            const long long m = n / divs[p + 1];  // >= the phase's length nvis / divs[p + 1] - nvis / divs[p]
            const bool last = p == K - 1;
            if (m <= 0 && !last) continue;
            const uint8_t* sat = first_phase ? nullptr : c->r_sat.as<uint8_t>();
            int where = 0;
            if (m > 0) {
                {
                    ProfScope ps(c, "render_bin_count");
                    hipLaunchKernelGGL(count_k, dim3(grid_for(m)), dim3(kRB), 0, c->stream,
                                       nvis_dev, divs[p], divs[p + 1], m, by_depth, c->r_rect.as<uint32_t>(), c->r_rec.as<float4>(), (float)H, lists_x,
                                       c->opt_exact_cull, sat, c->r_count.as<uint32_t>(), (const uint32_t*)nullptr,
                                       (uint32_t*)nullptr, (uint32_t*)nullptr, (const unsigned long long*)nullptr, 0ull, c->r_rects.as<uint32_t>());
                }
                GSX_HIP(c, hipGetLastError());
                rc = exclusive_scan_u32(c, c->r_count.as<uint32_t>(), c->r_offset.as<uint32_t>(), m, pairs_dev + p);
                if (rc) return rc;
                {
                    ProfScope ps(c, "render_bin_emit");
                    hipLaunchKernelGGL(emit_k, dim3(grid_for(m)), dim3(kRB), 0, c->stream,
                                       nvis_dev, divs[p], divs[p + 1], m, by_depth, c->r_rect.as<uint32_t>(), c->r_rec.as<float4>(), (float)H, lists_x,
                                       c->opt_exact_cull, sat, (uint32_t*)nullptr, c->r_offset.as<uint32_t>(),
                                       c->r_keys0.as<uint32_t>(), c->r_vals0.as<uint32_t>(), pairs_dev + p, (unsigned long long)cap, c->r_rects.as<uint32_t>());
                }
                GSX_HIP(c, hipGetLastError());
                if (wide) {  // one pass over the whole key; the digits' bases are the lists' ranges (sort.hip)
                    rc = radix_sort_values_wide(c, c->r_keys0.as<uint32_t>(), c->r_vals0.as<uint32_t>(), c->r_vals1.as<uint32_t>(),
                                                (long long)cap, pairs_dev + p, tile_bits, c->r_ranges.as<int2>(), nlists);
                    where = 1;
                } else {
                    rc = radix_sort_pairs_dev(c, c->r_keys0.as<uint32_t>(), c->r_vals0.as<uint32_t>(), c->r_keys1.as<uint32_t>(),
                                              c->r_vals1.as<uint32_t>(), (long long)cap, pairs_dev + p, tile_bits, &where);
                }
                if (rc) return rc;
            }
            if (!(wide && m > 0)) GSX_HIP(c, hipMemsetAsync(c->r_ranges.p, 0, sizeof(int2) * (size_t)nlists, c->stream));
            if (m > 0 && !wide) {
                ProfScope ps(c, "render_ranges");
                hipLaunchKernelGGL(ranges_kernel, dim3(grid_for((long long)cap)), dim3(kRB), 0, c->stream,
                                   where ? c->r_keys1.as<uint32_t>() : c->r_keys0.as<uint32_t>(), pairs_dev + p, (unsigned long long)cap,
                                   nlists, c->r_ranges.as<int2>());
                GSX_HIP(c, hipGetLastError());
            }
            c->r_sorted_in = where;
            rc = launch_blend(c, where ? c->r_vals1.as<uint32_t>() : c->r_vals0.as<uint32_t>(), W, H, tiles_x, tiles_y, small + 2,
                              consumed_dev, c->r_sat.as<uint8_t>(), first_phase, last ? 1 : 0);
            if (rc) return rc;
            first_phase = 0;
        }
        // end of frame: the only host wait.  Did every phase fit the pair buffers?
        unsigned long long stats[kSmallBytes / 8];
        GSX_HIP(c, hipMemcpyAsync(stats, small, sizeof stats, hipMemcpyDeviceToHost, c->stream));
        GSX_HIP(c, hipStreamSynchronize(c->stream));
        unsigned long long maxP = 0, sumP = 0;
        for (int p = 0; p < K; ++p) {
            maxP = std::max(maxP, stats[3 + p]);
            sumP += stats[3 + p];
        }
        if (maxP > 0x7fffffffull)  // offsets are 32 bit (and the scan's partial sums would have wrapped beyond 2^32)
            return fail(c, GSX_E_RANGE, "render_view: %llu (tile, splat) pairs in one depth phase exceed 2^31-1 "
                                        "(a close-up of a very large scene): raise the option \"render_phases\"", maxP);
        if (maxP > cap) {
            if (attempt >= 2) return fail(c, GSX_E_STATE, "render_view: pair buffers overflowed three times in a row");
            c->r_pair_cap = (size_t)(maxP + maxP / 4 + 4096);
            continue;
        }
        c->r_P = sumP;
        c->r_consumed = 0;
        for (int k = 0; k < kConsumedSlots; ++k) c->r_consumed += stats[32 + 16 * k];
        break;
    }
    if (rgba_out) {
        GSX_HIP(c, hipMemcpyAsync(rgba_out, c->r_image.p, img_bytes, hipMemcpyDeviceToHost, c->stream));
        GSX_HIP(c, hipStreamSynchronize(c->stream));
    }
    return GSX_OK;
}

// ---- several views: a few frames in flight ------------------------------------------------------------------------------
// A frame is a chain of dependent kernels: memory-bound per-splat passes and sorts first, then the VALU-bound blend with
// its tail of long tiles.  Frames on separate HIP streams fill each other's gaps (measured 916 -> 1221 views/s with two at
// 3 M splats / 1080p / SH 3).  The context therefore keeps TWINS: further streams with their own per-frame buffers that
// alias this context's scene (texel pairs, SH planes: nothing is duplicated), each driven by a host thread of its own; frame
// k is rendered by stream k % F (option "render_frames").
static int twin_sync_scene(Ctx* c, int k) {
    if (!c->twins[k]) {
        c->twins[k] = new (std::nothrow) Ctx();
        if (!c->twins[k]) return fail(c, GSX_E_INVALID, "render_views: out of host memory");
        c->twins[k]->device = c->device;
        hipError_t e = hipSuccess;
        if (k == 0 && c->opt_render_share_stream) {
            // the first extra frame runs on the context's second stream (the early vote's): with a stream of its own a process
            // that has labelled before holds six streams for four hardware queues, and frames that share a queue serialise
            // (bench.py's render leg: 1160-1200 views/s against 1335-1400 in a process that never labelled)
            if (second_stream(c) != GSX_OK) e = hipErrorUnknown;
            c->twins[k]->stream = c->stream2;
            c->twins[k]->stream_borrowed = true;
        } else {
            e = hipStreamCreateWithFlags(&c->twins[k]->stream, hipStreamNonBlocking);
        }
        if (e != hipSuccess) {
            c->twins[k]->stream = nullptr;
            delete c->twins[k];
            c->twins[k] = nullptr;
            return fail(c, GSX_E_HIP, "render_views: hipStreamCreate failed: %s", hipGetErrorString(e));
        }
    }
    Ctx* t = c->twins[k];
    t->r_tex.alias(c->r_tex);
    t->r_shc.alias(c->r_shc);
    t->rn = c->rn;
    t->r_sh_on = c->r_sh_on;
    t->r_sh_deg = c->r_sh_deg;
    t->opt_render_phases = c->opt_render_phases;
    t->opt_render_phase_ratio = c->opt_render_phase_ratio;
    t->opt_exact_cull = c->opt_exact_cull;
    t->opt_render_bin32 = c->opt_render_bin32;
    t->opt_render_compact = c->opt_render_compact;
    t->opt_render_wide_sort = c->opt_render_wide_sort;
    t->opt_blend_pk2 = c->opt_blend_pk2;
    t->opt_tile_lpt = c->opt_tile_lpt;
    if (t->r_pair_cap < c->r_pair_cap) t->r_pair_cap = c->r_pair_cap;
    return GSX_OK;
}

void render_release_twin(Ctx* c) {
    for (hipEvent_t& e : c->r_pre_ev)
        if (e) {
            (void)hipEventDestroy(e);
            e = nullptr;
        }
    for (Ctx*& t : c->twins) {
        if (!t) continue;
        (void)hipStreamSynchronize(t->stream);
        for (DevBuf* b : {&t->r_tex, &t->r_shc, &t->r_image, &t->r_ranges, &t->r_small, &t->r_scan, &t->r_depth, &t->r_bucket, &t->r_rect,
                          &t->r_count, &t->r_offset, &t->r_rec, &t->r_keys0, &t->r_keys1, &t->r_vals0, &t->r_vals1,
                          &t->r_tile_order, &t->r_sat, &t->r_d0, &t->r_d1, &t->r_d2, &t->r_d3, &t->r_rects, &t->sort_hist, &t->r_pre})
            b->release();
        for (Ctx::PreSet& ps : t->r_sets)
            for (DevBuf* b : {&ps.depth, &ps.rect, &ps.rec, &ps.pre}) b->release();
        if (!t->stream_borrowed) (void)hipStreamDestroy(t->stream);
        t->stream = nullptr;
        delete t;
        t = nullptr;
    }
}

int render_views(Ctx* c, int n, const gsx_camera* cams, int W, int H, float* const* rgba_out) {
    if (n < 0 || (n > 0 && !cams)) return fail(c, GSX_E_INVALID, "render_views: bad arguments");
    if (n == 0) return GSX_OK;
    const int F = std::max(1, std::min({c->opt_render_frames, (int)Ctx::kMaxFrames, n}));
    if (F == 1 || c->prof_on)  // the per-kernel events belong to one stream: profiled runs render one frame at a time
    {
        unsigned long long P = 0, used = 0;
        for (int k = 0; k < n; ++k) {
            const int rc = render_view(c, cams + k, W, H, rgba_out ? rgba_out[k] : nullptr);
            if (rc) return rc;
            P += c->r_P;
            used += c->r_consumed;
        }
        c->r_P = P;
        c->r_consumed = used;
        return GSX_OK;
    }
    int rc = GSX_OK;
    for (int f = 1; f < F; ++f)
        if ((rc = twin_sync_scene(c, f - 1))) return rc;
    // frame k is rendered by stream k % F: stream 0 is the context's own (this thread), the others have a host thread each.
    // Group g = the frames g * F .. g * F + F - 1.  With the option render_multi_pre the pre pass of a whole group is ONE launch
    // (pre_multi_kernel) on stream 0, issued by this thread ahead of its own frame of the previous group, into record set
    // g % 3 of every frame's context; the other frames wait for it through an event.  Host-side ordering (an event must have
    // been recorded before another thread may wait for it, a set must not be rewritten while a frame still reads it) goes
    // through two counters: pre_issued (groups whose pass has been launched) and done[f] (frames context f has completed -
    // render_view returns only after its end-of-frame synchronisation).
    int rcs[Ctx::kMaxFrames] = {};  // GSX_OK == 0
    unsigned long long P[Ctx::kMaxFrames] = {}, used[Ctx::kMaxFrames] = {};
    const bool multi = c->opt_render_multi_pre != 0 && c->rn > 0;
    const int groups = (n + F - 1) / F;
    std::atomic<int> pre_issued{0}, done[Ctx::kMaxFrames];
    std::atomic<bool> failed{false};
    for (auto& d : done) d.store(0);
    Ctx* ctxs[Ctx::kMaxFrames] = {c};
    for (int f = 1; f < F; ++f) ctxs[f] = c->twins[f - 1];
    if (multi) {
        for (int f = 0; f < F; ++f)
            for (Ctx::PreSet& ps : ctxs[f]->r_sets)
                if ((rc = ensure_pre_set(c, ps, c->rn))) return rc;
        GSX_HIP(c, c->r_pre_args.ensure(sizeof(PreMultiArgs) * Ctx::kPreSets));
        c->r_pre_args_host.resize((sizeof(PreMultiArgs) * Ctx::kPreSets + 7) / 8);
        for (int s = 0; s < Ctx::kPreSets; ++s)
            if (!c->r_pre_ev[s]) GSX_HIP(c, hipEventCreateWithFlags(&c->r_pre_ev[s], hipEventDisableTiming));
    }
    auto issue_pre = [&](int g) -> int {  // this thread only; stream 0
        // set g % 3 was last read by the frames of group g - 3: every context must have completed them
        for (int f = 0; f < F; ++f)
            while (done[f].load(std::memory_order_acquire) < std::min(g - 2, (n - 1 - f) / F + 1) && !failed.load()) std::this_thread::yield();
        if (failed.load()) return GSX_OK;
        const int set = g % Ctx::kPreSets;
        // the host copy of the arguments lives in the context, one slot per record set: the asynchronous copy below may read it
        // after this function has returned (a slot is rewritten three groups later)
        PreMultiArgs& a = reinterpret_cast<PreMultiArgs*>(c->r_pre_args_host.data())[set];
        a = PreMultiArgs{};
        a.nv = std::min(F, n - g * F);
        for (int v = 0; v < a.nv; ++v) {
            const gsx_camera* cam = cams + (g * F + v);
            Ctx::PreSet& ps = ctxs[v]->r_sets[set];
            a.u[v] = view_uniforms(cam, W, H);
            for (int k = 0; k < 3; ++k) a.cam[v][k] = (float)cam->p[k];
            a.depth[v] = ps.depth.as<int>();
            a.pre[v] = ps.pre.as<int>();
            a.rec[v] = ps.rec.as<float4>();
            a.rect[v] = ps.rect.as<uint32_t>();
            GSX_HIP(c, hipMemcpyAsync(ps.pre.p, kPreInit, sizeof kPreInit, hipMemcpyHostToDevice, c->stream));
        }
        PreMultiArgs* a_dev = c->r_pre_args.as<PreMultiArgs>() + set;
        GSX_HIP(c, hipMemcpyAsync(a_dev, &a, sizeof a, hipMemcpyHostToDevice, c->stream));
        using PK = void (*)(const uint4*, long long, const float*, int, const PreMultiArgs*);
        static const PK kernels[kPreViews] = {pre_multi_kernel<1>, pre_multi_kernel<2>, pre_multi_kernel<3>,
                                              pre_multi_kernel<4>, pre_multi_kernel<5>, pre_multi_kernel<6>};
        static_assert(kPreViews == 6, "one instantiation per group size");
        hipLaunchKernelGGL(kernels[a.nv - 1], dim3(std::min<unsigned>(grid_for(c->rn), 2048u)), dim3(kRB), 0, c->stream, c->r_tex.as<uint4>(),
                           (long long)c->rn, c->r_sh_on ? c->r_shc.as<float>() : nullptr, c->r_sh_deg, a_dev);
        GSX_HIP(c, hipGetLastError());
        GSX_HIP(c, hipEventRecord(c->r_pre_ev[set], c->stream));
        pre_issued.store(g + 1, std::memory_order_release);
        return GSX_OK;
    };
    auto frames_of = [&](Ctx* t, int f) {
        t->r_in_flight = true;  // (other frames run next to this context's: render_view picks the forms that share the GPU best)
        int g = 0;
        for (int k = f; k < n && rcs[f] == GSX_OK; k += F, ++g) {
            if (multi) {
                if (f == 0) {  // this thread: the pass of the NEXT group goes out before this group's frame (group 0's before the threads start)
                    if (g + 1 < groups && (rcs[f] = issue_pre(g + 1))) break;
                    if (failed.load()) break;  // (a pass that was not issued leaves no records to render from)
                } else {
                    while (pre_issued.load(std::memory_order_acquire) < g + 1 && !failed.load()) std::this_thread::yield();
                    if (failed.load()) break;
                    const hipError_t e = hipStreamWaitEvent(t->stream, c->r_pre_ev[g % Ctx::kPreSets], 0);
                    if (e != hipSuccess) {
                        rcs[f] = fail(t, GSX_E_HIP, "render_views: hipStreamWaitEvent: %s", hipGetErrorString(e));
                        break;
                    }
                }
                t->r_pre_ext = g % Ctx::kPreSets;
            }
            rcs[f] = render_view(t, cams + k, W, H, rgba_out ? rgba_out[k] : nullptr);
            t->r_pre_ext = -1;
            P[f] += t->r_P;
            used[f] += t->r_consumed;
            done[f].store(g + 1, std::memory_order_release);
        }
        t->r_pre_ext = -1;
        t->r_in_flight = false;
        if (rcs[f] != GSX_OK) failed.store(true);  // nobody waits for a pass or a frame that will not come
    };
    if (multi && (rc = issue_pre(0))) return rc;
    std::thread others[Ctx::kMaxFrames - 1];
    int started = 0;
    try {
        for (int f = 1; f < F; ++f) {
            Ctx* t = c->twins[f - 1];
            others[f - 1] = std::thread([&, t, f] {
                try {  // nothing may escape a thread's entry function
                    (void)hipSetDevice(t->device);
                    frames_of(t, f);
                } catch (...) {
                    rcs[f] = fail(t, GSX_E_INVALID, "render_views: a stream's host thread failed (out of host memory?)");
                    failed.store(true);
                }
            });
            ++started;
        }
        frames_of(c, 0);
    } catch (...) {
        failed.store(true);
        for (int f = 0; f < started; ++f) others[f].join();  // never leave a joinable thread behind
        throw;
    }
    for (int f = 0; f < started; ++f) others[f].join();
    for (int f = 0; f < F; ++f) ctxs[f]->r_in_flight = false;
    rc = rcs[0];
    for (int f = 1; f < F && rc == GSX_OK; ++f)
        if (rcs[f] != GSX_OK) {
            c->err = c->twins[f - 1]->err;
            rc = rcs[f];
        }
    c->r_P = 0;
    c->r_consumed = 0;
    for (int f = 0; f < F; ++f) {
        c->r_P += P[f];
        c->r_consumed += used[f];
    }
    return rc;
}

// ---- hit test: performHitTesting, gs.js:361-395 -------------------------------------------------------------
// An arg-min reduction over all splats of the key (dist, depth, index): the sequential JS loop keeps the
// first splat that is strictly nearer, or equally near and strictly less deep.  fp64, JS operation order.
struct HitKey {
    double dist, depth;
    long long idx;  // -1: none
};

__device__ __forceinline__ bool hit_better(const HitKey& a, const HitKey& b) {  // a replaces b?
    if (a.idx < 0) return false;
    if (b.idx < 0) return true;
    if (a.dist < b.dist) return true;
    if (a.dist > b.dist) return false;
    if (a.depth < b.depth) return true;
    if (a.depth > b.depth) return false;
    return a.idx < b.idx;
}

__device__ __forceinline__ double js_hypot2(double a, double b) {  // V8 Math.hypot for two finite-or-not doubles
    const double big = __longlong_as_double(0x7ff0000000000000LL);
    if (fabs(a) == big || fabs(b) == big) return big;
    if (a != a || b != b) return a + b;
    const double v0 = fabs(a), v1 = fabs(b);
    const double mx = v0 > v1 ? v0 : v1;
    if (mx == 0.0) return 0.0;
    double sum = 0.0, comp = 0.0;
    {
        const double q = v0 / mx;
        const double summand = (q * q) - comp;
        const double pre = sum + summand;
        comp = (pre - sum) - summand;
        sum = pre;
    }
    {
        const double q = v1 / mx;
        const double summand = (q * q) - comp;
        const double pre = sum + summand;
        comp = (pre - sum) - summand;
        sum = pre;
    }
    return sqrt(sum) * mx;
}

struct HitUniforms {
    double m[16];  // proj * view, gs.js:364
    double x, y, vw, vh;
};

__device__ __forceinline__ HitKey hit_reduce_block(HitKey k, HitKey* sh /*[4]*/) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        HitKey other;
        other.dist = __shfl_xor(k.dist, o);
        other.depth = __shfl_xor(k.depth, o);
        other.idx = __shfl_xor(k.idx, o);
        if (hit_better(other, k)) k = other;
    }
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = k;
    __syncthreads();
    HitKey r = sh[0];
    for (int w = 1; w < 4; ++w)
        if (hit_better(sh[w], r)) r = sh[w];
    return r;
}

__global__ __launch_bounds__(kRB) void hit_partial_kernel(const uint4* __restrict__ tex, long long n, HitUniforms u,
                                                           HitKey* __restrict__ partial) {
    __shared__ HitKey sh[4];
    HitKey best{0.0, 0.0, -1};
    for (long long i = (long long)blockIdx.x * kRB + threadIdx.x; i < n; i += (long long)gridDim.x * kRB) {
        const uint4 t = tex[2 * i];
        const double p0 = (double)__uint_as_float(t.x), p1 = (double)__uint_as_float(t.y), p2 = (double)__uint_as_float(t.z);
        double r[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) r[k] = p0 * u.m[k] + p1 * u.m[k + 4] + p2 * u.m[k + 8] + 1.0 * u.m[k + 12];
        if (r[3] <= 0.0) continue;  // gs.js:402
        const double sx = (r[0] / r[3] + 1.0) * 0.5 * u.vw;
        const double sy = (r[1] / r[3] + 1.0) * 0.5 * u.vh;
        const double depth = r[2] / r[3];
        const double dist = js_hypot2(sx - u.x, sy - u.y);
        if (dist < 10.0) {  // gs.js:387
            const HitKey k{dist, depth, i};
            if (hit_better(k, best)) best = k;
        }
    }
    best = hit_reduce_block(best, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = best;
}

__global__ __launch_bounds__(kRB) void hit_final_kernel(const HitKey* __restrict__ partial, int m,
                                                         const uint4* __restrict__ tex, long long* __restrict__ out) {
    __shared__ HitKey sh[4];
    HitKey best{0.0, 0.0, -1};
    for (int t = threadIdx.x; t < m; t += kRB)
        if (hit_better(partial[t], best)) best = partial[t];
    best = hit_reduce_block(best, sh);
    if (threadIdx.x == 0) {
        out[0] = best.idx;
        out[1] = best.idx >= 0 ? (long long)(int)__uint_as_float(tex[2 * best.idx].w) : -999999LL;  // labelData[i]
    }
}

int hit_test(Ctx* c, const gsx_camera* cam, int W, int H, double x, double y, int32_t* label_out, int64_t* index_out) {
    GSX_HIP(c, hipSetDevice(c->device));
    if (!cam || W < 1 || H < 1) return fail(c, GSX_E_INVALID, "hit_test: bad arguments");
    long long res[2] = {-1, -999999};
    if (c->rn > 0) {
        HitUniforms u{};
        double view[16], proj[16];
        js_view_matrix(cam, view);
        js_proj_matrix(cam->fx, cam->fy, (double)W, (double)H, proj);
        js_multiply4(proj, view, u.m);
        u.x = x;
        u.y = y;
        u.vw = (double)W;
        u.vh = (double)H;
        const int blocks = (int)std::min<long long>(1024, (c->rn + kRB - 1) / kRB);
        GSX_HIP(c, c->r_scan.ensure(sizeof(HitKey) * (size_t)blocks + 16));
        HitKey* partial = c->r_scan.as<HitKey>();
        long long* out = reinterpret_cast<long long*>(partial + blocks);
        {
            ProfScope ps(c, "hit_test");
            hipLaunchKernelGGL(hit_partial_kernel, dim3(blocks), dim3(kRB), 0, c->stream, c->r_tex.as<uint4>(), (long long)c->rn, u,
                               partial);
            hipLaunchKernelGGL(hit_final_kernel, dim3(1), dim3(kRB), 0, c->stream, partial, blocks, c->r_tex.as<uint4>(), out);
        }
        GSX_HIP(c, hipGetLastError());
        GSX_HIP(c, hipMemcpyAsync(res, out, sizeof res, hipMemcpyDeviceToHost, c->stream));
        GSX_HIP(c, hipStreamSynchronize(c->stream));
    }
    if (index_out) *index_out = res[0];
    if (label_out) *label_out = (int32_t)res[1];
    return GSX_OK;
}

int render_debug(Ctx* c, uint8_t* buffer_out, uint32_t* order_out, uint32_t* tex_out, uint32_t* bucket_out) {
    GSX_HIP(c, hipSetDevice(c->device));
    const size_t n = (size_t)c->rn;
    if (n == 0) return GSX_OK;
    if (buffer_out) GSX_HIP(c, hipMemcpyAsync(buffer_out, c->r_buffer.p, 32 * n, hipMemcpyDeviceToHost, c->stream));
    if (order_out) GSX_HIP(c, hipMemcpyAsync(order_out, c->r_order.p, 4 * n, hipMemcpyDeviceToHost, c->stream));
    if (tex_out) GSX_HIP(c, hipMemcpyAsync(tex_out, c->r_tex.p, 32 * n, hipMemcpyDeviceToHost, c->stream));
    if (bucket_out) {
        if (c->r_bucket.cap < 4 * n) return fail(c, GSX_E_STATE, "render_debug: no view has been rendered yet");
        GSX_HIP(c, hipMemcpyAsync(bucket_out, c->r_bucket.p, 4 * n, hipMemcpyDeviceToHost, c->stream));
    }
    GSX_HIP(c, hipStreamSynchronize(c->stream));
    return GSX_OK;
}

}  // namespace gsx
