/*
 * ORACLE (test infrastructure, NOT the product): CPU restatement of the reference viewer's
 * rendering path.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Restates /root/reference/Web_Viewer_Gaussians_Selection/gaussians_selection.js ("gs.js"):
 *   gsxo_pack_splats      processPlyBuffer          gs.js:513-582  (importance order, 32-byte rows)
 *   gsxo_texture          generateTexture           gs.js:286-357  (+ floatToHalf 248-275, packHalf2x16 277-279)
 *   gsxo_view_matrix      getViewMatrix             gs.js:81-107
 *   gsxo_proj_matrix      calculateProjectionMatrix gs.js:66-79    (Z_NEAR 0.2, Z_FAR 200: gs.js:10-11)
 *   gsxo_multiply4        multiply4                 gs.js:110-123
 *   gsxo_depth_order      runSort                   gs.js:432-457  (16-bit counting sort, incl. its
 *                                                    out-of-range bucket for the farthest splat)
 *   gsxo_vertex           vertex shader             gs.js:696-750
 *   gsxo_render_view      fragment shader + blend   gs.js:782-799, 1033-1038, 1608-1609
 *
 * Parity status.  The JavaScript half (pack / texture / matrices / depth order) is PINNED byte for
 * byte by tests/golden/render_js.npz, produced by running the reference's own worker under node
 * (tools/make_golden_js.js).  The GLSL half (gsxo_vertex / gsxo_render_view) is PINNED since round 3 by
 * tests/golden/render_gl_{cases,scenes}.npz: frames of the reference's own vertex + fragment shaders and blend state
 * executed by Mesa llvmpipe (the software OpenGL ES 3.2 of the image, reached through the DRI swrast interface:
 * tools/gl_reference/gl_frames.c, tests/golden/make_golden_gl.py) on the reference's own texture / depthIndex -
 * <= 1e-4 per channel on every pixel except fragments that sit on the discard threshold A = -4, which GL's
 * 1/256-pixel vertex snapping decides (1 pixel in ~50 000 over 128 random frames, profiles/r03/gl_reference_soak.txt).
 * Rounds 1-2 had known-answer tests only.  The SH colour (gsxo_sh_colors) stays "parity unpinned": gs.js reads f_dc only.
 * JavaScript numbers are fp64 and never fused: build with -ffp-contract=off.  GLSL highp float is
 * restated in fp32 without contraction.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * helpers with JavaScript semantics
 * ---------------------------------------------------------------------------------------------- */
/* Uint8ClampedArray store (ToUint8Clamp): NaN -> 0, clamp, round half to even */
static uint8_t js_u8clamp(double x) {
    if (!(x > 0.0)) return 0; /* NaN, negatives, zero */
    if (x >= 255.0) return 255;
    double f = floor(x);
    if (f + 0.5 < x) return (uint8_t)(f + 1.0);
    if (x < f + 0.5) return (uint8_t)f;
    return (uint8_t)(fmod(f, 2.0) == 0.0 ? f : f + 1.0);
}

/* ToInt32 of `x | 0` */
static int32_t js_toint32(double x) {
    if (!isfinite(x)) return 0;
    double t = trunc(x);
    double m = fmod(t, 4294967296.0);
    if (m < 0) m += 4294967296.0;
    return (int32_t)(uint32_t)m;
}

/* floatToHalf, gs.js:248-275: f64 -> f32 (round to nearest) -> half by TRUNCATION; JS shifts are mod 32 */
static uint32_t js_float_to_half(double v) {
    float fl = (float)v;
    int32_t f;
    memcpy(&f, &fl, 4);
    int32_t sign = (f >> 31) & 0x0001;
    int32_t exp = (f >> 23) & 0x00ff;
    int32_t frac = f & 0x007fffff;
    int32_t newExp;
    if (exp == 0) {
        newExp = 0;
    } else if (exp < 113) {
        newExp = 0;
        frac |= 0x00800000;
        frac = frac >> ((113 - exp) & 31);
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
    return (uint32_t)((sign << 15) | (newExp << 10) | (frac >> 13));
}

static uint32_t js_pack_half2(double x, double y) { return js_float_to_half(x) | (js_float_to_half(y) << 16); }

/* ------------------------------------------------------------------------------------------------
 * processPlyBuffer, gs.js:513-582
 * xyz n x 3, scale n x 3 (log), rot n x 4 (w first), opacity n (logit), f_dc n x 3; all f32.
 * scale == NULL -> the "no scale_0" fallbacks (gs.js:519, 559-563); f_dc == NULL -> rgb (u8 n x 3).
 * buffer: n x 32 bytes in importance order; order[j] = source row of packed row j.
 * ---------------------------------------------------------------------------------------------- */
static void merge_sort_desc(uint32_t* idx, uint32_t* tmp, const float* key, int64_t lo, int64_t hi) {
    if (hi - lo < 2) return;
    int64_t mid = lo + (hi - lo) / 2;
    merge_sort_desc(idx, tmp, key, lo, mid);
    merge_sort_desc(idx, tmp, key, mid, hi);
    int64_t a = lo, b = mid, o = lo;
    while (a < mid && b < hi) {
        /* comparator (b, a) => sizeList[a] - sizeList[b] (gs.js:527): stable, descending */
        if (key[idx[b]] > key[idx[a]]) tmp[o++] = idx[b++];
        else tmp[o++] = idx[a++];
    }
    while (a < mid) tmp[o++] = idx[a++];
    while (b < hi) tmp[o++] = idx[b++];
    memcpy(idx + lo, tmp + lo, sizeof(uint32_t) * (size_t)(hi - lo));
}

void gsxo_pack_splats(int64_t n, const float* xyz, const float* scale, const float* rot, const float* opacity,
                      const float* f_dc, const uint8_t* rgb, uint8_t* buffer, uint32_t* order) {
    float* size_list = (float*)calloc((size_t)(n > 0 ? n : 1), sizeof(float));
    uint32_t* tmp = (uint32_t*)malloc(sizeof(uint32_t) * (size_t)(n > 0 ? n : 1));
    for (int64_t r = 0; r < n; ++r) {
        order[r] = (uint32_t)r;
        if (!scale) continue; /* gs.js:519 */
        double size = exp((double)scale[3 * r]) * exp((double)scale[3 * r + 1]) * exp((double)scale[3 * r + 2]);
        double op = 1.0 / (1.0 + exp(-(double)opacity[r]));
        size_list[r] = (float)(size * op); /* Float32Array store */
    }
    merge_sort_desc(order, tmp, size_list, 0, n);
    const double SH_C0 = 0.28209479177387814;
    for (int64_t j = 0; j < n; ++j) {
        const int64_t r = order[j];
        uint8_t* row = buffer + 32 * j;
        float f[6];
        f[0] = xyz[3 * r];
        f[1] = xyz[3 * r + 1];
        f[2] = xyz[3 * r + 2];
        uint8_t* rgba = row + 24;
        uint8_t* q = row + 28;
        if (scale) {
            const double r0 = rot[4 * r], r1 = rot[4 * r + 1], r2 = rot[4 * r + 2], r3 = rot[4 * r + 3];
            const double qlen = sqrt(r0 * r0 + r1 * r1 + r2 * r2 + r3 * r3);
            q[0] = js_u8clamp((r0 / qlen) * 128.0 + 128.0);
            q[1] = js_u8clamp((r1 / qlen) * 128.0 + 128.0);
            q[2] = js_u8clamp((r2 / qlen) * 128.0 + 128.0);
            q[3] = js_u8clamp((r3 / qlen) * 128.0 + 128.0);
            f[3] = (float)exp((double)scale[3 * r]);
            f[4] = (float)exp((double)scale[3 * r + 1]);
            f[5] = (float)exp((double)scale[3 * r + 2]);
        } else {
            f[3] = f[4] = f[5] = (float)0.01;
            q[0] = 255;
            q[1] = q[2] = q[3] = 0;
        }
        memcpy(row, f, 24);
        if (f_dc) {
            rgba[0] = js_u8clamp((0.5 + SH_C0 * (double)f_dc[3 * r]) * 255.0);
            rgba[1] = js_u8clamp((0.5 + SH_C0 * (double)f_dc[3 * r + 1]) * 255.0);
            rgba[2] = js_u8clamp((0.5 + SH_C0 * (double)f_dc[3 * r + 2]) * 255.0);
        } else {
            rgba[0] = rgb[3 * r];
            rgba[1] = rgb[3 * r + 1];
            rgba[2] = rgb[3 * r + 2];
        }
        rgba[3] = opacity ? js_u8clamp((1.0 / (1.0 + exp(-(double)opacity[r]))) * 255.0) : 255;
    }
    free(size_list);
    free(tmp);
}

/* ------------------------------------------------------------------------------------------------
 * generateTexture, gs.js:301-354 (no displacement, everything visible).
 * labels: int32 per PACKED row (labelData), or NULL -> NO_SELECTION (-999999, gs.js:6,579).
 * texdata: 8 u32 per splat: [x, y, z, label-as-f32] [h01, h23, h45, rgba8]
 * ---------------------------------------------------------------------------------------------- */
void gsxo_texture(int64_t n, const uint8_t* buffer, const int32_t* labels, uint32_t* texdata) {
    for (int64_t i = 0; i < n; ++i) {
        const uint8_t* row = buffer + 32 * i;
        float f[6];
        memcpy(f, row, 24);
        uint32_t* t = texdata + 8 * i;
        memcpy(t, f, 12);
        float lab = (float)(labels ? labels[i] : -999999);
        memcpy(t + 3, &lab, 4);
        memcpy(t + 7, row + 24, 4);
        const double scale[3] = {f[3], f[4], f[5]};
        double rot[4];
        for (int k = 0; k < 4; ++k) rot[k] = ((double)row[28 + k] - 128.0) / 128.0;
        double M[9] = {
            1.0 - 2.0 * (rot[2] * rot[2] + rot[3] * rot[3]),
            2.0 * (rot[1] * rot[2] + rot[0] * rot[3]),
            2.0 * (rot[1] * rot[3] - rot[0] * rot[2]),
            2.0 * (rot[1] * rot[2] - rot[0] * rot[3]),
            1.0 - 2.0 * (rot[1] * rot[1] + rot[3] * rot[3]),
            2.0 * (rot[2] * rot[3] + rot[0] * rot[1]),
            2.0 * (rot[1] * rot[3] + rot[0] * rot[2]),
            2.0 * (rot[2] * rot[3] - rot[0] * rot[1]),
            1.0 - 2.0 * (rot[1] * rot[1] + rot[2] * rot[2]),
        };
        for (int k = 0; k < 9; ++k) M[k] = M[k] * scale[k / 3];
        const double sigma[6] = {
            M[0] * M[0] + M[3] * M[3] + M[6] * M[6], M[0] * M[1] + M[3] * M[4] + M[6] * M[7],
            M[0] * M[2] + M[3] * M[5] + M[6] * M[8], M[1] * M[1] + M[4] * M[4] + M[7] * M[7],
            M[1] * M[2] + M[4] * M[5] + M[7] * M[8], M[2] * M[2] + M[5] * M[5] + M[8] * M[8],
        };
        t[4] = js_pack_half2(4 * sigma[0], 4 * sigma[1]);
        t[5] = js_pack_half2(4 * sigma[2], 4 * sigma[3]);
        t[6] = js_pack_half2(4 * sigma[4], 4 * sigma[5]);
    }
}

/* ------------------------------------------------------------------------------------------------
 * camera matrices (column-major 4x4, fp64 like JS numbers)
 * ---------------------------------------------------------------------------------------------- */
void gsxo_view_matrix(const double R[9], const double p[3], double out[16]) {
    for (int i = 0; i < 3; ++i) {
        out[4 * i + 0] = R[3 * i + 0]; /* gs.js:94-97: rows of `rotation` become columns */
        out[4 * i + 1] = R[3 * i + 1];
        out[4 * i + 2] = R[3 * i + 2];
        out[4 * i + 3] = 0.0;
    }
    for (int i = 0; i < 3; ++i) out[12 + i] = -p[0] * R[i] - p[1] * R[i + 3] - p[2] * R[i + 6]; /* gs.js:86-91 */
    out[15] = 1.0;
}

void gsxo_proj_matrix(double fx, double fy, double width, double height, double out[16]) {
    const double Z_FAR = 200.0, Z_NEAR = 0.2;
    const double zRange = Z_FAR - Z_NEAR;
    memset(out, 0, sizeof(double) * 16);
    out[0] = (2 * fx) / width;
    out[5] = -(2 * fy) / height;
    out[10] = Z_FAR / zRange;
    out[11] = 1.0;
    out[14] = -(Z_FAR * Z_NEAR) / zRange;
}

void gsxo_multiply4(const double A[16], const double B[16], double out[16]) { /* gs.js:110-123 */
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            out[4 * i + j] = B[4 * i] * A[j] + B[4 * i + 1] * A[j + 4] + B[4 * i + 2] * A[j + 8] + B[4 * i + 3] * A[j + 12];
}

/* ------------------------------------------------------------------------------------------------
 * runSort, gs.js:432-457.  depth_index[n]: front-to-back order; slots the JS never writes stay 0.
 * Returns the number of splats the JS drops (bucket 65536: typed-array writes out of range are no-ops).
 * ---------------------------------------------------------------------------------------------- */
int64_t gsxo_depth_order(int64_t n, const uint8_t* buffer, const double viewproj[16], uint32_t* depth_index) {
    int32_t* size_list = (int32_t*)malloc(sizeof(int32_t) * (size_t)(n > 0 ? n : 1));
    uint32_t* counts = (uint32_t*)calloc(65536, sizeof(uint32_t));
    uint32_t* starts = (uint32_t*)calloc(65536, sizeof(uint32_t));
    double maxDepth = -INFINITY, minDepth = INFINITY;
    for (int64_t i = 0; i < n; ++i) {
        float f[3];
        memcpy(f, buffer + 32 * i, 12);
        int32_t depth = js_toint32((viewproj[2] * (double)f[0] + viewproj[6] * (double)f[1] + viewproj[10] * (double)f[2]) * 4096.0);
        size_list[i] = depth;
        if (depth > maxDepth) maxDepth = depth;
        if (depth < minDepth) minDepth = depth;
    }
    const double depthInv = (256.0 * 256.0) / (maxDepth - minDepth);
    for (int64_t i = 0; i < n; ++i) {
        size_list[i] = js_toint32(((double)size_list[i] - minDepth) * depthInv);
        if (size_list[i] >= 0 && size_list[i] < 65536) counts[size_list[i]]++;
    }
    for (int i = 1; i < 65536; ++i) starts[i] = starts[i - 1] + counts[i - 1];
    memset(depth_index, 0, sizeof(uint32_t) * (size_t)n);
    int64_t dropped = 0;
    for (int64_t i = 0; i < n; ++i) {
        if (size_list[i] >= 0 && size_list[i] < 65536) depth_index[starts[size_list[i]]++] = (uint32_t)i;
        else ++dropped;
    }
    free(size_list);
    free(counts);
    free(starts);
    return dropped;
}

/* ------------------------------------------------------------------------------------------------
 * vertex shader, gs.js:696-750 (fp32, no selection / displacement)
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t drawn;     /* 0: culled (gs.js:709-713), lambda2 < 0 (gs.js:736) or a degenerate/NaN quad */
    float cx, cy;      /* splat centre in GL window coordinates (pixels, y up) */
    float major[2];    /* gs.js:738, pixels, GL window axes */
    float minor[2];    /* gs.js:739 */
    float color[4];    /* vColor, gs.js:741-742 */
    float fade;        /* clamp(z/w + 1, 0, 1), the factor inside vColor */
    float g0[2];       /* 2*major/|major|^2 : vPosition.x = dot(d, g0) for a window offset d from the centre */
    float g1[2];       /* 2*minor/|minor|^2 : vPosition.y = dot(d, g1) */
} gsxo_vertex_out;

static float half_to_float(uint32_t h) {
    const uint32_t s = (h >> 15) & 1, e = (h >> 10) & 31, m = h & 1023;
    float v;
    if (e == 0) v = ldexpf((float)m, -24);
    else if (e == 31) v = m ? NAN : INFINITY;
    else v = ldexpf((float)(m | 1024), (int)e - 25);
    return s ? -v : v;
}

void gsxo_vertex(const uint32_t* texel8, const float view[16], const float proj[16], float fx, float fy, float W,
                 float H, gsxo_vertex_out* o) {
    memset(o, 0, sizeof *o);
    float c[3];
    memcpy(c, texel8, 12);
    float cam[4], p2[4];
    for (int i = 0; i < 4; ++i) cam[i] = view[i] * c[0] + view[4 + i] * c[1] + view[8 + i] * c[2] + view[12 + i] * 1.0f;
    for (int i = 0; i < 4; ++i) p2[i] = proj[i] * cam[0] + proj[4 + i] * cam[1] + proj[8 + i] * cam[2] + proj[12 + i] * cam[3];
    const float clip = 1.2f * p2[3];
    if (p2[2] < -clip || p2[0] < -clip || p2[0] > clip || p2[1] < -clip || p2[1] > clip) return;
    const float u1x = half_to_float(texel8[4] & 0xffff), u1y = half_to_float(texel8[4] >> 16);
    const float u2x = half_to_float(texel8[5] & 0xffff), u2y = half_to_float(texel8[5] >> 16);
    const float u3x = half_to_float(texel8[6] & 0xffff), u3y = half_to_float(texel8[6] >> 16);
    /* Vrk columns (symmetric): (u1x,u1y,u2x) (u1y,u2y,u3x) (u2x,u3x,u3y) */
    const float V[3][3] = {{u1x, u1y, u2x}, {u1y, u2y, u3x}, {u2x, u3x, u3y}};
    const float ja = fx / cam[2], jb = -(fx * cam[0]) / (cam[2] * cam[2]);
    const float jc = -fy / cam[2], jd = (fy * cam[1]) / (cam[2] * cam[2]);
    /* T = transpose(mat3(view)) * J : T[j][i] = sum_k view[4i+k] * J[j][k] */
    float t0[3], t1[3];
    for (int i = 0; i < 3; ++i) {
        t0[i] = view[4 * i + 0] * ja + view[4 * i + 1] * 0.0f + view[4 * i + 2] * jb;
        t1[i] = view[4 * i + 0] * 0.0f + view[4 * i + 1] * jc + view[4 * i + 2] * jd;
    }
    /* cov2d = transpose(T) * Vrk * T, evaluated left to right */
    float a0[3], a1[3]; /* rows of transpose(T)*Vrk */
    for (int k = 0; k < 3; ++k) {
        a0[k] = t0[0] * V[k][0] + t0[1] * V[k][1] + t0[2] * V[k][2];
        a1[k] = t1[0] * V[k][0] + t1[1] * V[k][1] + t1[2] * V[k][2];
    }
    const float c00 = a0[0] * t0[0] + a0[1] * t0[1] + a0[2] * t0[2];
    const float c01 = a1[0] * t0[0] + a1[1] * t0[1] + a1[2] * t0[2]; /* cov2d[0][1]: column 0, row 1 */
    const float c11 = a1[0] * t1[0] + a1[1] * t1[1] + a1[2] * t1[2];
    const float mid = (c00 + c11) / 2.0f;
    const float hx = (c00 - c11) / 2.0f;
    const float radius = sqrtf(hx * hx + c01 * c01);
    const float l1 = mid + radius, l2 = mid - radius;
    if (l2 < 0.0f) return;
    const float dx = c01, dy = l1 - c00;
    const float dl = sqrtf(dx * dx + dy * dy);
    const float ux = dx / dl, uy = dy / dl;
    const float s1 = fminf(sqrtf(2.0f * l1), 1024.0f), s2 = fminf(sqrtf(2.0f * l2), 1024.0f);
    o->major[0] = s1 * ux;
    o->major[1] = s1 * uy;
    o->minor[0] = s2 * uy;
    o->minor[1] = s2 * -ux;
    float fade = p2[2] / p2[3] + 1.0f;
    fade = fade < 0.0f ? 0.0f : (fade > 1.0f ? 1.0f : fade);
    const uint32_t rgba = texel8[7];
    o->fade = fade;
    for (int k = 0; k < 4; ++k) o->color[k] = fade * (float)((rgba >> (8 * k)) & 0xff) / 255.0f;
    const float ndcx = p2[0] / p2[3], ndcy = p2[1] / p2[3];
    o->cx = (ndcx + 1.0f) * 0.5f * W;
    o->cy = (ndcy + 1.0f) * 0.5f * H;
    const float m2 = o->major[0] * o->major[0] + o->major[1] * o->major[1];
    const float n2 = o->minor[0] * o->minor[0] + o->minor[1] * o->minor[1];
    /* a quad with a zero or non-finite axis has no area / no valid position: nothing is rasterised */
    if (!(m2 > 0.0f) || !(n2 > 0.0f) || !isfinite(m2) || !isfinite(n2) || !isfinite(o->cx) || !isfinite(o->cy)) return;
    /* The rasteriser interpolates `position` (corners +-2) linearly over the quad whose window-space
     * half-diagonals are major and minor (orthogonal): a pixel at offset d from the centre gets
     * vPosition = (2 d.major/|major|^2, 2 d.minor/|minor|^2). */
    o->g0[0] = 2.0f * o->major[0] / m2;
    o->g0[1] = 2.0f * o->major[1] / m2;
    o->g1[0] = 2.0f * o->minor[0] / n2;
    o->g1[1] = 2.0f * o->minor[1] / n2;
    o->drawn = 1;
}

/* ------------------------------------------------------------------------------------------------
 * fragment shader + blend over the whole frame, splats in depth_index order (ALL n slots, trailing
 * zeros included: drawArraysInstanced(..., vertexCount) with the index attribute, gs.js:1076-1077,1609).
 * rgba_out: H x W x 4 floats, row 0 = top image row, premultiplied colour, cleared to 0 (gs.js:1608).
 * override_color: NULL, or n x 4 floats whose rgb replace rgba8/255 per splat (SH colour path,
 * not in the reference: gs.js reads only f_dc); alpha always comes from the packed u8.
 * ---------------------------------------------------------------------------------------------- */
void gsxo_render_view(int64_t n, const uint32_t* texdata, const uint32_t* depth_index, const double view64[16],
                      const double proj64[16], double fx64, double fy64, int32_t W, int32_t H,
                      const float* override_color, float* rgba_out) {
    float view[16], proj[16];
    for (int k = 0; k < 16; ++k) { /* gl.uniformMatrix4fv / uniform2fv: JS numbers -> f32 */
        view[k] = (float)view64[k];
        proj[k] = (float)proj64[k];
    }
    const float fx = (float)fx64, fy = (float)fy64;
    memset(rgba_out, 0, sizeof(float) * 4 * (size_t)W * (size_t)H);
    for (int64_t k = 0; k < n; ++k) {
        const uint32_t i = depth_index[k];
        gsxo_vertex_out v;
        gsxo_vertex(texdata + 8 * (size_t)i, view, proj, fx, fy, (float)W, (float)H, &v);
        if (!v.drawn) continue;
        if (override_color) /* SH colour path: the depth fade multiplies whatever rgb the splat has */
            for (int q = 0; q < 3; ++q) v.color[q] = v.fade * override_color[4 * (size_t)i + q];
        const float ex = sqrtf(v.major[0] * v.major[0] + v.minor[0] * v.minor[0]) + 1.0f;
        const float ey = sqrtf(v.major[1] * v.major[1] + v.minor[1] * v.minor[1]) + 1.0f;
        /* window y is up; image row r has its centre at yw = H - (r + 0.5) */
        int x0 = (int)floorf(v.cx - ex), x1 = (int)ceilf(v.cx + ex);
        int r0 = (int)floorf((float)H - (v.cy + ey)), r1 = (int)ceilf((float)H - (v.cy - ey));
        if (x0 < 0) x0 = 0;
        if (r0 < 0) r0 = 0;
        if (x1 > W - 1) x1 = W - 1;
        if (r1 > H - 1) r1 = H - 1;
        for (int r = r0; r <= r1; ++r) {
            const float dy = ((float)H - ((float)r + 0.5f)) - v.cy;
            for (int x = x0; x <= x1; ++x) {
                const float dx = ((float)x + 0.5f) - v.cx;
                const float vx = dx * v.g0[0] + dy * v.g0[1]; /* interpolated vPosition */
                const float vy = dx * v.g1[0] + dy * v.g1[1];
                const float A = -(vx * vx + vy * vy);
                if (A < -4.0f) continue; /* discard, gs.js:784 */
                const float B = expf(A) * v.color[3];
                float* px = rgba_out + 4 * ((size_t)r * W + x);
                const float om = 1.0f - px[3]; /* ONE_MINUS_DST_ALPHA, ONE: gs.js:1037 */
                px[0] += om * (B * v.color[0]);
                px[1] += om * (B * v.color[1]);
                px[2] += om * (B * v.color[2]);
                px[3] += om * B;
            }
        }
    }
}

/* ------------------------------------------------------------------------------------------------
 * Spherical-harmonics colour, degrees 0..3 — NOT in the reference (gs.js reads only f_dc, 565-569);
 * BASELINE.json's config 3 asks for SH degree 3, so the standard 3DGS real-SH basis (Kerbl et al. 2023,
 * evaluated along dir = normalize(position - camera position)) is restated here.  Parity for this term
 * is UNPINNED by the reference.  rgb = clamp(0.5 + sum_k basis_k(dir) * coeff_k, 0, 1), fp32.
 * xyz: n x 3 positions; f_dc: n x 3; f_rest: n x 3*((deg+1)^2-1), channel-major as in a 3DGS PLY
 * (f_rest_[c*K1 + k-1]); rows are in the SAME order as xyz.  out: n x 4 (alpha slot unused = 0).
 * ---------------------------------------------------------------------------------------------- */
void gsxo_sh_colors(int64_t n, const float* xyz, const float* f_dc, const float* f_rest, int32_t deg,
                    const double campos64[3], float* out) {
    const float C0 = 0.28209479177387814f, C1 = 0.4886025119029199f;
    const float C2[5] = {1.0925484305920792f, -1.0925484305920792f, 0.31539156525252005f, -1.0925484305920792f, 0.5462742152960396f};
    const float C3[7] = {-0.5900435899266435f, 2.890611442640554f, -0.4570457994644658f, 0.3731763325901154f,
                         -0.4570457994644658f, 1.445305721320277f, -0.5900435899266435f};
    const int K1 = (deg + 1) * (deg + 1) - 1;
    const float cp[3] = {(float)campos64[0], (float)campos64[1], (float)campos64[2]};
    for (int64_t i = 0; i < n; ++i) {
        float dx = xyz[3 * i] - cp[0], dy = xyz[3 * i + 1] - cp[1], dz = xyz[3 * i + 2] - cp[2];
        const float len = sqrtf(dx * dx + dy * dy + dz * dz);
        const float x = dx / len, y = dy / len, z = dz / len;
        float b[16];
        b[0] = C0;
        if (deg > 0) {
            b[1] = -C1 * y;
            b[2] = C1 * z;
            b[3] = -C1 * x;
        }
        const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
        if (deg > 1) {
            b[4] = C2[0] * xy;
            b[5] = C2[1] * yz;
            b[6] = C2[2] * (2.0f * zz - xx - yy);
            b[7] = C2[3] * xz;
            b[8] = C2[4] * (xx - yy);
        }
        if (deg > 2) {
            b[9] = C3[0] * y * (3.0f * xx - yy);
            b[10] = C3[1] * xy * z;
            b[11] = C3[2] * y * (4.0f * zz - xx - yy);
            b[12] = C3[3] * z * (2.0f * zz - 3.0f * xx - 3.0f * yy);
            b[13] = C3[4] * x * (4.0f * zz - xx - yy);
            b[14] = C3[5] * z * (xx - yy);
            b[15] = C3[6] * x * (xx - 3.0f * yy);
        }
        for (int c = 0; c < 3; ++c) {
            float acc = b[0] * f_dc[3 * i + c];
            for (int k = 1; k <= K1; ++k) acc += b[k] * f_rest[(size_t)i * 3 * K1 + (size_t)c * K1 + (k - 1)];
            acc += 0.5f;
            out[4 * i + c] = acc < 0.0f ? 0.0f : (acc > 1.0f ? 1.0f : acc);
        }
        out[4 * i + 3] = 0.0f;
    }
}

/* ------------------------------------------------------------------------------------------------
 * performHitTesting, gs.js:361-395 (+ project 397-404): the label of the splat whose projected centre is
 * nearest to the click (x, y in canvas pixels, y up) within 10 px, nearer depth breaking exact ties.
 * Math.hypot is V8's (node 12 / V8 7.8, builtins-math.cc): scale by the maximum, Kahan-summed squares.
 * labels: per PACKED row, or NULL -> NO_SELECTION.  Returns the label, *index_out = winning row or -1.
 * Pinned by tests/golden/render_js.npz (hit_* arrays, produced under node).
 * ---------------------------------------------------------------------------------------------- */
static double js_hypot2(double a, double b) {
    if (isinf(a) || isinf(b)) return INFINITY;
    if (isnan(a) || isnan(b)) return NAN;
    const double v[2] = {fabs(a), fabs(b)};
    const double max = v[0] > v[1] ? v[0] : v[1];
    if (max == 0) return 0;
    double sum = 0, compensation = 0;
    for (int i = 0; i < 2; ++i) {
        const double n = v[i] / max;
        const double summand = (n * n) - compensation;
        const double preliminary = sum + summand;
        compensation = (preliminary - sum) - summand;
        sum = preliminary;
    }
    return sqrt(sum) * max;
}

int32_t gsxo_hit_test(int64_t n, const uint8_t* buffer, const int32_t* labels, const double view[16],
                      const double proj[16], double x, double y, double vw, double vh, int64_t* index_out) {
    double m[16];
    gsxo_multiply4(proj, view, m);
    double closestDist = INFINITY, closestDepth = INFINITY;
    int32_t selected = -999999;
    int64_t idx = -1;
    for (int64_t i = 0; i < n; ++i) {
        float f[3];
        memcpy(f, buffer + 32 * i, 12);
        const double pos[4] = {f[0], f[1], f[2], 1.0};
        double r[4];
        for (int k = 0; k < 4; ++k) r[k] = pos[0] * m[k] + pos[1] * m[k + 4] + pos[2] * m[k + 8] + pos[3] * m[k + 12];
        if (r[3] <= 0) continue;
        const double screenX = (r[0] / r[3] + 1) * 0.5 * vw;
        const double screenY = (r[1] / r[3] + 1) * 0.5 * vh;
        const double depth = r[2] / r[3];
        const double dist = js_hypot2(screenX - x, screenY - y);
        if (dist < 10 && (dist < closestDist || (dist == closestDist && depth < closestDepth))) {
            closestDist = dist;
            closestDepth = depth;
            selected = labels ? labels[i] : -999999;
            idx = i;
        }
    }
    if (index_out) *index_out = idx;
    return selected;
}
