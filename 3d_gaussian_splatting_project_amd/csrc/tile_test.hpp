// tile_test.hpp - device helpers shared by render.hip (binning, option exact_cull) and blend.hip (staging).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace gsx {

// Exact tile culling.  A pixel takes a fragment of the splat iff q(d) = (d.g0)^2 + (d.g1)^2 <= 4 for its
// centre's offset d from the splat centre (A = -q, `discard` if A < -4).  q is a convex quadratic, so its
// minimum over a tile's rectangle of pixel centres is 0 (centre inside) or lies on one of the 4 edges, where
// q is a 1-D quadratic.  A tile is kept iff that minimum <= 4.04 (1 % of slack over the fp32 rounding of
// the per-pixel evaluation): elongated and diagonal splats lose most of their bounding-box tiles.
__device__ __forceinline__ float edge_min_q(float ax, float ay, float bx, float by, float g0x, float g0y, float g1x,
                                            float g1y) {
    // p(t) = a + t (b - a), t in [0,1]; u(t) = p.g0, v(t) = p.g1 are affine in t
    const float u0 = ax * g0x + ay * g0y, v0 = ax * g1x + ay * g1y;
    const float du = (bx - ax) * g0x + (by - ay) * g0y, dv = (bx - ax) * g1x + (by - ay) * g1y;
    const float den = du * du + dv * dv;
    float t = den > 0.0f ? -(u0 * du + v0 * dv) / den : 0.0f;
    t = fminf(fmaxf(t, 0.0f), 1.0f);
    const float u = u0 + t * du, v = v0 + t * dv;
    return u * u + v * v;
}

__device__ __forceinline__ bool tile_touches(float cx, float cy, float g0x, float g0y, float g1x, float g1y, float H,
                                             uint32_t tx, uint32_t ty) {
    // rectangle of the tile's pixel centres, relative to the splat centre, GL window coordinates (y up)
    const float x0 = (float)(tx * 16u) + 0.5f - cx, x1 = x0 + 15.0f;
    const float y1 = H - ((float)(ty * 16u) + 0.5f) - cy, y0 = y1 - 15.0f;
    if (x0 <= 0.0f && x1 >= 0.0f && y0 <= 0.0f && y1 >= 0.0f) return true;
    float q = edge_min_q(x0, y0, x1, y0, g0x, g0y, g1x, g1y);
    q = fminf(q, edge_min_q(x0, y1, x1, y1, g0x, g0y, g1x, g1y));
    q = fminf(q, edge_min_q(x0, y0, x0, y1, g0x, g0y, g1x, g1y));
    q = fminf(q, edge_min_q(x1, y0, x1, y1, g0x, g0y, g1x, g1y));
    return !(q > 4.04f);  // NaN keeps the tile
}

}  // namespace gsx
