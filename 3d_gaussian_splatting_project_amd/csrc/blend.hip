// blend.hip — fragment shader + blend unit of the reference viewer as a tile kernel for gfx950.
//
// Restates gaussians_selection.js:782-799 (A = -|vPosition|^2, discard A < -4, B = exp(A) * alpha,
// premultiplied colour) and the blend state of gs.js:1036-1038 (dst += (1 - dst.a) * src, front to
// back) for one 16x16-pixel tile per workgroup.  The tile's depth-ordered splat list is staged
// through LDS 256 records at a time (one gather per thread), then every pixel walks the staged records
// with broadcast LDS reads.  (A per-wave 8x8-quadrant bounding-box skip was measured: 0.42 ms vs 0.33 ms
// per 1080p view of the 3 M-splat scene — the pairs that matter come from splats wider than a quadrant,
// so the test is pure overhead; removed.)  Compiled with -ffp-contract=off: the operations that
// decide `discard` round exactly like the oracle's; the accumulation uses explicit fmaf.
//
// A frame is blended in depth PHASES (render.hip): a phase's kernel continues from the image the previous phase left
// (first phase: the cleared canvas), marks the tiles whose every pixel has become opaque (`sat`: the next phases do not
// even bin splats into them), and the last phase appends the dropped-splat epilogue.  Tiles without pairs in a middle
// phase return at once.
#include <hip/hip_runtime.h>

#include "gsx_ctx.hpp"
#include "tile_test.hpp"

namespace gsx {

static constexpr int kTile = 16;
static constexpr int kBlendThreads = kTile * kTile;

struct Accum {
    float r, g, b, a;
};

__device__ __forceinline__ void blend_one(Accum& acc, float fxp, float fyp, const float4 r0, const float4 r1,
                                          const float2 r2) {
    const float dx = fxp - r0.x;
    const float dy = fyp - r0.y;
    const float vx = dx * r0.z + dy * r0.w;  // interpolated vPosition (see oracle/render_oracle.c gsxo_vertex)
    const float vy = dx * r1.x + dy * r1.y;
    const float A = -(vx * vx + vy * vy);
    if (!(A < -4.0f)) {  // `if (A < -4.0) discard;`
        const float B = __expf(A) * r2.y;
        const float om = 1.0f - acc.a;  // ONE_MINUS_DST_ALPHA, ONE
        acc.r = __builtin_fmaf(om, B * r1.z, acc.r);
        acc.g = __builtin_fmaf(om, B * r1.w, acc.g);
        acc.b = __builtin_fmaf(om, B * r2.x, acc.b);
        acc.a = __builtin_fmaf(om, B, acc.a);
    }
}

// A tile of a LATER phase that got no pairs (or is opaque already) has nothing to do - unless it is the last phase and the
// dropped-splat epilogue (splat 0 drawn again, gs.js:453-457) reaches it: only the tiles inside splat 0's bounding box, and
// only while they are not opaque (everything behind an opaque tile adds < 1e-5 in total, epilogue included).  Such a tile
// returns without reading or writing the canvas: the previous phase's pixels are final.  (Round 2 re-read and re-wrote the
// whole canvas in the last phase: 66 MB of the 184 MB a 1080p frame moved.)
__device__ __forceinline__ bool tile_has_nothing_to_do(int2 range, const uint8_t* __restrict__ sat, int tile, int tx, int ty, int first,
                                                       int last, const int* __restrict__ dropped, const int* __restrict__ pre,
                                                       long long n) {
    if (first) return false;
    const bool opaque = sat[tile] != 0;
    if (range.y > range.x && !opaque) return false;
    if (!last) return true;
    if (opaque || n <= 0 || dropped[0] <= 0) return true;
    const uint32_t r = (uint32_t)pre[2];  // splat 0's tile rectangle (pre_kernel)
    const int tx0 = r & 255u, tx1 = (r >> 8) & 255u, ty0 = (r >> 16) & 255u, ty1 = r >> 24;
    return tx < tx0 || tx > tx1 || ty < ty0 || ty > ty1;
}

// Longest-list-first launch order: the kernel ends when its slowest tile does (a tile that never
// saturates walks its whole list), so the long tiles must not be dealt last.  key = 63 - 2*log2(len)
// rounded to half octaves: one 6-bit radix pass over the tiles.
__global__ __launch_bounds__(256) void tile_order_key_kernel(const int2* __restrict__ ranges, int ntiles,
                                                             uint32_t* __restrict__ keys, uint32_t* __restrict__ vals) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= ntiles) return;
    const unsigned len = (unsigned)(ranges[t].y - ranges[t].x);
    unsigned k = 0;
    if (len) {
        const unsigned lg = 31u - (unsigned)__clz(len);
        k = 2u * lg + ((len >> (lg ? lg - 1 : 0)) & 1u) + 1u;
    }
    keys[t] = 63u - min(k, 63u);
    vals[t] = (uint32_t)t;
}

__global__ __launch_bounds__(kBlendThreads) void blend_kernel(const int2* __restrict__ ranges,
                                                              const uint32_t* __restrict__ tile_order,
                                                              const uint32_t* __restrict__ vals,
                                                              const float4* __restrict__ rec, int W, int H, int tiles_x,
                                                              const int* __restrict__ dropped, const int* __restrict__ pre, long long n,
                                                              unsigned long long* __restrict__ consumed,
                                                              float4* __restrict__ image, uint8_t* __restrict__ sat, int first,
                                                              int last) {
    __shared__ float4 s0[kBlendThreads];
    __shared__ float4 s1[kBlendThreads];
    __shared__ float2 s2[kBlendThreads];
    const int tile = tile_order ? (int)tile_order[blockIdx.x] : (int)blockIdx.x;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int px = tx * kTile + (threadIdx.x & (kTile - 1));
    const int py = ty * kTile + (threadIdx.x >> 4);
    const bool inside = px < W && py < H;
    const float fxp = (float)px + 0.5f;               // pixel centre, GL window coordinates
    const float fyp = (float)H - ((float)py + 0.5f);  // window y is up; image row py counts from the top
    const int2 range = ranges[tile];
    if (tile_has_nothing_to_do(range, sat, tile, tx, ty, first, last, dropped, pre, n)) return;
    Accum acc{0.f, 0.f, 0.f, 0.f};                    // gl.clear to (0,0,0,0), gs.js:1608
    if (!first && inside) {
        const float4 prev = image[(size_t)py * W + px];
        acc = Accum{prev.x, prev.y, prev.z, prev.w};
    }
    int staged = 0;  // list entries this tile actually read
    bool opaque = false;
    for (int base = range.x; base < range.y; base += kBlendThreads) {
        const int cnt = min(kBlendThreads, range.y - base);
        __syncthreads();
        if ((int)threadIdx.x < cnt) {
            const uint32_t id = vals[base + threadIdx.x];
            s0[threadIdx.x] = rec[3 * (size_t)id];
            s1[threadIdx.x] = rec[3 * (size_t)id + 1];
            s2[threadIdx.x] = *reinterpret_cast<const float2*>(rec + 3 * (size_t)id + 2);
        }
        __syncthreads();
        for (int k = 0; k < cnt; ++k) blend_one(acc, fxp, fyp, s0[k], s1[k], s2[k]);
        staged += cnt;
        // every further fragment is weighted by (1 - dst.a): once that is < 1e-5 on the whole tile the
        // rest of the list changes no channel by more than 1e-5 (the parity tolerance is 1e-4)
        opaque = __syncthreads_and(!inside || acc.a > 1.0f - 1.0e-5f) != 0;
        if (opaque) break;
    }
    if (opaque && threadIdx.x == 0) sat[tile] = 1;
    // runSort leaves the slots of the splats it drops (bucket 65536) at 0: splat 0 is drawn again,
    // last, once per dropped splat (gs.js:453-457 + 1076-1077, 1609)
    const int nd = (last && n > 0) ? *dropped : 0;
    if (nd > 0) {
        const float4 r0 = rec[0];
        const float4 r1 = rec[1];
        const float2 r2 = *reinterpret_cast<const float2*>(rec + 2);
        for (int k = 0; k < nd; ++k) blend_one(acc, fxp, fyp, r0, r1, r2);
    }
    if (inside) image[(size_t)py * W + px] = make_float4(acc.r, acc.g, acc.b, acc.a);
    if (threadIdx.x == 0 && staged) atomicAdd(consumed + (blockIdx.x & (kConsumedSlots - 1)) * 16, (unsigned long long)staged);  // statistics, spread over
                                                                                        // kConsumedSlots lines: same-address atomics serialise in one L2 channel
}

// ---- two pixels per thread ---------------------------------------------------------------------------------
// Same maths, but a thread owns the pixels (x, y) and (x, y + 8) of the tile and carries them as 2-vectors:
// hipcc turns the element-wise products and sums into v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 (two fp32
// results per lane and instruction) without the operand shuffles the one-pixel kernel needs, and a tile is
// 2 waves instead of 4, so every staged record is read from LDS half as often.  Per component the operation
// order is unchanged (mul, mul, add; no contraction), so `discard` decides exactly as before.
typedef float f2 __attribute__((ext_vector_type(2)));
static constexpr int kBlend2Threads = 128;
static constexpr int kBlend2Chunk = 256;  // (128: +3 % views/s, 512: -20 %; the one-wave kernel below is the default now)

struct Accum2 {
    f2 r, g, b, a;
};

__device__ __forceinline__ f2 splat_q2(float fxp, f2 fyp, const float4 r0, const float4 r1) {
    const float dx = fxp - r0.x;
    const f2 dy = fyp - r0.y;
    const f2 vx = dx * r0.z + dy * r0.w;
    const f2 vy = dx * r1.x + dy * r1.y;
    return vx * vx + vy * vy;  // A = -q; `if (A < -4.0) discard;` <=> q > 4 (negation is exact)
}

__device__ __forceinline__ void accumulate2(Accum2& acc, f2 q, const float4 r1, const float2 r2) {
    const bool k0 = !(q.x > 4.0f), k1 = !(q.y > 4.0f);
    if (k0 | k1) {
        f2 B;
        B.x = k0 ? __expf(-q.x) * r2.y : 0.0f;  // a discarded fragment adds exactly nothing: fma(om, 0, x) == x
        B.y = k1 ? __expf(-q.y) * r2.y : 0.0f;
        const f2 om = 1.0f - acc.a;  // ONE_MINUS_DST_ALPHA, ONE
        acc.r = __builtin_elementwise_fma(om, B * r1.z, acc.r);
        acc.g = __builtin_elementwise_fma(om, B * r1.w, acc.g);
        acc.b = __builtin_elementwise_fma(om, B * r2.x, acc.b);
        acc.a = __builtin_elementwise_fma(om, B, acc.a);
    }
}

__device__ __forceinline__ void blend_one2(Accum2& acc, float fxp, f2 fyp, const float4 r0, const float4 r1,
                                           const float2 r2) {
    accumulate2(acc, splat_q2(fxp, fyp, r0, r1), r1, r2);
}

__global__ __launch_bounds__(kBlend2Threads) void blend2_kernel(const int2* __restrict__ ranges,
                                                                const uint32_t* __restrict__ tile_order,
                                                                const uint32_t* __restrict__ vals,
                                                                const float4* __restrict__ rec, int W, int H, int tiles_x,
                                                                const int* __restrict__ dropped, const int* __restrict__ pre, long long n,
                                                                unsigned long long* __restrict__ consumed,
                                                                float4* __restrict__ image, uint8_t* __restrict__ sat, int first,
                                                                int last) {
    __shared__ float4 s0[kBlend2Chunk];
    __shared__ float4 s1[kBlend2Chunk];
    __shared__ float2 s2[kBlend2Chunk];
    const int tile = tile_order ? (int)tile_order[blockIdx.x] : (int)blockIdx.x;
    const int tx = tile % tiles_x, ty = tile / tiles_x;
    const int px = tx * kTile + (threadIdx.x & (kTile - 1));
    const int py0 = ty * kTile + (threadIdx.x >> 4), py1 = py0 + 8;
    const bool in0 = px < W && py0 < H, in1 = px < W && py1 < H;
    const float fxp = (float)px + 0.5f;
    f2 fyp;
    fyp.x = (float)H - ((float)py0 + 0.5f);
    fyp.y = (float)H - ((float)py1 + 0.5f);
    const int2 range = ranges[tile];
    if (tile_has_nothing_to_do(range, sat, tile, tx, ty, first, last, dropped, pre, n)) return;
    Accum2 acc{};
    if (!first) {
        float4 p0 = make_float4(0.f, 0.f, 0.f, 0.f), p1 = p0;
        if (in0) p0 = image[(size_t)py0 * W + px];
        if (in1) p1 = image[(size_t)py1 * W + px];
        acc.r.x = p0.x, acc.g.x = p0.y, acc.b.x = p0.z, acc.a.x = p0.w;
        acc.r.y = p1.x, acc.g.y = p1.y, acc.b.y = p1.z, acc.a.y = p1.w;
    }
    int staged = 0;
    bool opaque = false;
    for (int base = range.x; base < range.y; base += kBlend2Chunk) {
        const int cnt = min(kBlend2Chunk, range.y - base);
        __syncthreads();
        for (int t = threadIdx.x; t < cnt; t += kBlend2Threads) {
            const uint32_t id = vals[base + t];
            s0[t] = rec[3 * (size_t)id];
            s1[t] = rec[3 * (size_t)id + 1];
            s2[t] = *reinterpret_cast<const float2*>(rec + 3 * (size_t)id + 2);
        }
        __syncthreads();
        // the splats are independent up to the accumulation: evaluate 4 quadratics at once (4 dependency chains
        // in flight - the long tiles run one or two waves per CU, latency is what they wait for), then blend in order
        int k = 0;
        for (; k + 4 <= cnt; k += 4) {
            f2 q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) q[u] = splat_q2(fxp, fyp, s0[k + u], s1[k + u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) accumulate2(acc, q[u], s1[k + u], s2[k + u]);
        }
        for (; k < cnt; ++k) blend_one2(acc, fxp, fyp, s0[k], s1[k], s2[k]);
        staged += cnt;
        const bool done0 = !in0 || acc.a.x > 1.0f - 1.0e-5f, done1 = !in1 || acc.a.y > 1.0f - 1.0e-5f;
        opaque = __syncthreads_and(done0 && done1) != 0;
        if (opaque) break;
    }
    if (opaque && threadIdx.x == 0) sat[tile] = 1;
    const int nd = (last && n > 0) ? *dropped : 0;
    if (nd > 0) {
        const float4 r0 = rec[0];
        const float4 r1 = rec[1];
        const float2 r2 = *reinterpret_cast<const float2*>(rec + 2);
        for (int k = 0; k < nd; ++k) blend_one2(acc, fxp, fyp, r0, r1, r2);
    }
    if (in0) image[(size_t)py0 * W + px] = make_float4(acc.r.x, acc.g.x, acc.b.x, acc.a.x);
    if (in1) image[(size_t)py1 * W + px] = make_float4(acc.r.y, acc.g.y, acc.b.y, acc.a.y);
    if (threadIdx.x == 0 && staged) atomicAdd(consumed + (blockIdx.x & (kConsumedSlots - 1)) * 16, (unsigned long long)staged);  // statistics, spread over
                                                                                        // kConsumedSlots lines: same-address atomics serialise in one L2 channel
}

// (Round 3 tried the records through the SCALAR unit instead of LDS: every lane of a tile walks the same list, so list entry and
// record are wave-uniform; hipcc emits s_load_dword / s_load_dwordx4 for them and the VALU instructions take them as SGPR
// operands - no LDS, no staging barriers, four records in flight per wave, 94 SGPRs, pixels identical.  Measured at 3 M splats /
// 1080p: 0.485 ms of blend per view against 0.319, 1256-1261 views/s against 1448-1464 (profiles/r03/blend_scalar_ab.txt): three
// 64-byte lines per record through a 16 KB scalar cache cost more than the broadcast ds_reads they replace.  Removed.)

// ---- four pixels per thread, one wave per tile ----------------------------------------------------------------
// A tile that never saturates walks its whole list one dependent splat after the other; what it waits for is
// latency.  Here a lane owns FOUR pixels (rows y, y+4, y+8, y+12 as two packed pairs): four independent
// accumulation chains per lane hide that latency, the whole tile is one wave (no barrier partner to wait
// for), and every staged record is read from LDS once per tile.
static constexpr int kBlend4Threads = 64;
// Staging chunk and distance between two opacity votes, measured at 3 M splats / 1080p with four frames in flight
// (profiles/r03/blend_chunks.txt; views/s, pairs the blend really evaluates per view): a vote per chunk of 256 / 128 / 64 records
// 1410 / 1476 / 1540 (2.23 M / 2.05 M / 1.77 M pairs); chunk 64 with a vote every 32 / 16 / 8 records 1561-1581 / 1579-1587 /
// 1575-1583 (1.64 M / 1.58 M / 1.55 M); chunk 128 or 256 with a vote every 16: 1567-1571 / 1548; fetching the next chunk's records into
// registers while this one is blended: no gain (1533-1540 against 1549-1552 on the same box).  What a tile walks behind the record
// at which its last pixel saturates is pure waste, and round 2's 256-record chunks of the two-pixel kernel walked 40 % more
// records than needed.
static constexpr int kBlend4Chunk = 64;
static constexpr int kBlend4Group = 16;  // records between two "is the whole tile opaque?" votes (even)
static_assert(kBlend4Group % 2 == 0 && kBlend4Group >= 2, "the pair loop");

// BIN32 (option render_bin32): the pairs are sorted by 32x32-pixel bin and a list entry's top four bits name the bin's tiles the
// splat was binned for (render.hip: bin_kernel).  Workgroup b is tile (b & 3) of bin b >> 2; it walks the BIN's list 256 entries
// at a time, keeps the entries whose mask has its bit (four coalesced loads and four ballots, compacted in list order into
// `ids`) and stages the records of those 64 at a time as before: the same splats in the same order as its own list would hold.
static constexpr int kBlend4Super = 256;
template <bool BIN32>
__global__ __launch_bounds__(kBlend4Threads) void blend4_kernel(const int2* __restrict__ ranges,
                                                                const uint32_t* __restrict__ tile_order,
                                                                const uint32_t* __restrict__ vals,
                                                                const float4* __restrict__ rec, int W, int H, int tiles_x,
                                                                const int* __restrict__ dropped, const int* __restrict__ pre, long long n,
                                                                unsigned long long* __restrict__ consumed,
                                                                float4* __restrict__ image, uint8_t* __restrict__ sat, int first,
                                                                int last, int bins_x, int tiles_y) {
    __shared__ float4 s0[kBlend4Chunk];
    __shared__ float4 s1[kBlend4Chunk];
    __shared__ float2 s2[kBlend4Chunk];
    __shared__ uint32_t ids[BIN32 ? kBlend4Super + kBlend4Chunk : 1];
    int tile, tx, ty, list;
    uint32_t my_bit = 0u;
    if (BIN32) {
        list = (int)(blockIdx.x >> 2);
        const int sub = (int)(blockIdx.x & 3u);
        tx = 2 * (list % bins_x) + (sub & 1);
        ty = 2 * (list / bins_x) + (sub >> 1);
        if (tx >= tiles_x || ty >= tiles_y) return;  // the frame's last column / row of bins may be half empty
        tile = (int)blockIdx.x;                      // index of the tile's opacity byte: bin * 4 + bit
        my_bit = 1u << (28 + sub);
    } else {
        tile = tile_order ? (int)tile_order[blockIdx.x] : (int)blockIdx.x;
        tx = tile % tiles_x, ty = tile / tiles_x;
        list = tile;
    }
    const int px = tx * kTile + (threadIdx.x & (kTile - 1));
    const int row = ty * kTile + (threadIdx.x >> 4);  // rows row, row+4 (pair A: the tile's upper half), row+8, row+12 (pair B: the lower half)
    // A splat that reaches only one half of the tile skips the other pair's accumulation for the whole wave (the `if` of
    // accumulate2 is then false in every lane); with the rows interleaved (A = row, row+8) both pairs spanned the tile.
    const float fxp = (float)px + 0.5f;
    f2 fyA, fyB;
    fyA.x = (float)H - ((float)row + 0.5f);
    fyA.y = (float)H - ((float)(row + 4) + 0.5f);
    fyB.x = (float)H - ((float)(row + 8) + 0.5f);
    fyB.y = (float)H - ((float)(row + 12) + 0.5f);
    const bool in[4] = {px < W && row < H, px < W && row + 4 < H, px < W && row + 8 < H, px < W && row + 12 < H};
    const int2 range = ranges[list];
    if (tile_has_nothing_to_do(range, sat, tile, tx, ty, first, last, dropped, pre, n)) return;
    Accum2 accA{}, accB{};
    if (!first) {
        // (`in ? image[..] : z` made hipcc keep z in scratch and load through a selected address: 32 bytes of scratch per lane)
        float4 p0 = make_float4(0.f, 0.f, 0.f, 0.f), p1 = p0, p2 = p0, p3 = p0;
        if (in[0]) p0 = image[(size_t)row * W + px];
        if (in[1]) p1 = image[(size_t)(row + 4) * W + px];
        if (in[2]) p2 = image[(size_t)(row + 8) * W + px];
        if (in[3]) p3 = image[(size_t)(row + 12) * W + px];
        accA.r.x = p0.x, accA.g.x = p0.y, accA.b.x = p0.z, accA.a.x = p0.w;
        accA.r.y = p1.x, accA.g.y = p1.y, accA.b.y = p1.z, accA.a.y = p1.w;
        accB.r.x = p2.x, accB.g.x = p2.y, accB.b.x = p2.z, accB.a.x = p2.w;
        accB.r.y = p3.x, accB.g.y = p3.y, accB.b.y = p3.z, accB.a.y = p3.w;
    }
    int staged = 0;
    bool opaque = false;
    const unsigned long long lane_lt = (1ull << threadIdx.x) - 1ull;
    // BIN32: `ids` buffers this tile's entries (have of them) until 64 are there or the bin's list ends, so that the staged
    // chunks - and with them the records at which the opacity votes fall - are those of the tile's own list: bit-identical frames.
    int have = 0, sbase = range.x;
    for (;;) {
        if (BIN32) {
            while (have < kBlend4Chunk && sbase < range.y) {  // this tile's entries among the next 256 of the bin's list
                uint32_t v[kBlend4Super / kBlend4Threads];
#pragma unroll
                for (int j = 0; j < kBlend4Super / kBlend4Threads; ++j) {
                    const int e = sbase + j * kBlend4Threads + (int)threadIdx.x;
                    v[j] = e < range.y ? vals[e] : 0u;
                }
#pragma unroll
                for (int j = 0; j < kBlend4Super / kBlend4Threads; ++j) {
                    const bool mine = (v[j] & my_bit) != 0u;
                    const unsigned long long m = __ballot(mine);
                    if (mine) ids[have + __popcll(m & lane_lt)] = v[j] & 0x0fffffffu;
                    have += __popcll(m);
                }
                sbase += kBlend4Super;
            }
        } else {
            have = min(kBlend4Chunk, range.y - sbase);
        }
        if (have <= 0) break;
        const bool tail = sbase >= range.y;  // (BIN32) nothing left to fetch: a short last chunk is due
        int pos = 0;
        do {
        const int cnt = min(kBlend4Chunk, have - pos);
        __syncthreads();
        // A lane stages one record - and first asks whether its ellipse reaches this tile at all (the list holds the tiles of the
        // ellipse's BOUNDING BOX: a sixth of the pairs never produce a fragment).  One lane's ~110 instructions spare the whole
        // wave the ~26 it spends on finding that out pixel by pixel; the survivors are compacted with one ballot, in list order.
        // (In the bin kernels the same test costs more than it saves - option exact_cull: it runs per candidate tile there, 23 M
        // times per view, here 1.5 M times.)
        static_assert(kBlend4Chunk == kBlend4Threads, "one record per lane and chunk");
        bool keep = false;
        float4 a0, a1;
        float2 a2;
        if ((int)threadIdx.x < cnt) {
            const uint32_t id = BIN32 ? ids[pos + threadIdx.x] : vals[sbase + threadIdx.x];
            a0 = rec[3 * (size_t)id];
            a1 = rec[3 * (size_t)id + 1];
            a2 = *reinterpret_cast<const float2*>(rec + 3 * (size_t)id + 2);
            keep = tile_touches(a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, (float)H, (uint32_t)tx, (uint32_t)ty);
        }
        const unsigned long long kept = __ballot(keep);
        if (keep) {
            const int slot = __popcll(kept & lane_lt);
            s0[slot] = a0;
            s1[slot] = a1;
            s2[slot] = a2;
        }
        const int cnt_kept = __popcll(kept);
        __syncthreads();
        // the tile is ONE wave: "every pixel opaque" is a wave vote, no barrier - taken every kBlend4Group records, not once per
        // staged chunk (the records behind the point where the last pixel saturates are pure waste)
        for (int k0 = 0; k0 < cnt_kept && !opaque; k0 += kBlend4Group) {
            const int k1 = min(k0 + kBlend4Group, cnt_kept);
            int k = k0;
            for (; k + 2 <= k1; k += 2) {
                f2 qA[2], qB[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    qA[u] = splat_q2(fxp, fyA, s0[k + u], s1[k + u]);
                    qB[u] = splat_q2(fxp, fyB, s0[k + u], s1[k + u]);
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    accumulate2(accA, qA[u], s1[k + u], s2[k + u]);
                    accumulate2(accB, qB[u], s1[k + u], s2[k + u]);
                }
            }
            for (; k < k1; ++k) {
                blend_one2(accA, fxp, fyA, s0[k], s1[k], s2[k]);
                blend_one2(accB, fxp, fyB, s0[k], s1[k], s2[k]);
            }
            staged += k1 - k0;
            const float lim = 1.0f - 1.0e-5f;
            const bool done = (!in[0] || accA.a.x > lim) && (!in[1] || accA.a.y > lim) && (!in[2] || accB.a.x > lim) &&
                              (!in[3] || accB.a.y > lim);
            opaque = __all(done) != 0;
        }
        pos += cnt;
        } while (BIN32 && !opaque && (have - pos >= kBlend4Chunk || (tail && have > pos)));
        if (opaque) break;
        if (BIN32) {  // the entries behind the last full chunk move to the front
            const int left = have - pos;
            __syncthreads();
            const uint32_t keep_id = (int)threadIdx.x < left ? ids[pos + threadIdx.x] : 0u;
            __syncthreads();
            if ((int)threadIdx.x < left) ids[threadIdx.x] = keep_id;
            have = left;
            if (left == 0 && tail) break;
        } else {
            sbase += kBlend4Chunk;
        }
    }
    if (opaque && threadIdx.x == 0) sat[tile] = 1;
    const int nd = (last && n > 0) ? *dropped : 0;
    if (nd > 0) {
        const float4 r0 = rec[0];
        const float4 r1 = rec[1];
        const float2 r2 = *reinterpret_cast<const float2*>(rec + 2);
        for (int k = 0; k < nd; ++k) {
            blend_one2(accA, fxp, fyA, r0, r1, r2);
            blend_one2(accB, fxp, fyB, r0, r1, r2);
        }
    }
    if (in[0]) image[(size_t)row * W + px] = make_float4(accA.r.x, accA.g.x, accA.b.x, accA.a.x);
    if (in[1]) image[(size_t)(row + 4) * W + px] = make_float4(accA.r.y, accA.g.y, accA.b.y, accA.a.y);
    if (in[2]) image[(size_t)(row + 8) * W + px] = make_float4(accB.r.x, accB.g.x, accB.b.x, accB.a.x);
    if (in[3]) image[(size_t)(row + 12) * W + px] = make_float4(accB.r.y, accB.g.y, accB.b.y, accB.a.y);
    if (threadIdx.x == 0 && staged) atomicAdd(consumed + (blockIdx.x & (kConsumedSlots - 1)) * 16, (unsigned long long)staged);  // statistics, spread over
                                                                                        // kConsumedSlots lines: same-address atomics serialise in one L2 channel
}

int launch_blend(Ctx* c, const uint32_t* vals, int W, int H, int tiles_x, int tiles_y, const int* dropped_dev,
                 unsigned long long* consumed_dev, uint8_t* sat, int first_phase, int last_phase) {
    const int ntiles = tiles_x * tiles_y;
    const uint32_t* order = nullptr;
    if (c->opt_tile_lpt && !c->r_bin32 && c->r_P > 0 && ntiles > 256) {  // (tile_order_key_kernel reads per-tile ranges)
        GSX_HIP(c, c->r_tile_order.ensure(sizeof(uint32_t) * 4 * (size_t)ntiles));
        uint32_t* k0 = c->r_tile_order.as<uint32_t>();
        uint32_t *v0 = k0 + ntiles, *k1 = v0 + ntiles, *v1 = k1 + ntiles;
        hipLaunchKernelGGL(tile_order_key_kernel, dim3((ntiles + 255) / 256), dim3(256), 0, c->stream, c->r_ranges.as<int2>(),
                           ntiles, k0, v0);
        int where = 0;
        int rc = radix_sort_pairs(c, k0, v0, k1, v1, ntiles, 6, &where);
        if (rc) return rc;
        order = where ? v1 : v0;
    }
    ProfScope ps(c, "render_blend");
    if (c->opt_blend_pk2 == 2) {
        const int bins_x = (tiles_x + 1) / 2, bins_y = (tiles_y + 1) / 2;
        if (c->r_bin32)
            hipLaunchKernelGGL(blend4_kernel<true>, dim3(4 * bins_x * bins_y), dim3(kBlend4Threads), 0, c->stream, c->r_ranges.as<int2>(),
                               (const uint32_t*)nullptr, vals, c->r_rec.as<float4>(), W, H, tiles_x, dropped_dev,
                               c->r_pre.as<int>(), (long long)c->rn, consumed_dev, c->r_image.as<float4>(), sat, first_phase, last_phase,
                               bins_x, tiles_y);
        else
            hipLaunchKernelGGL(blend4_kernel<false>, dim3(ntiles), dim3(kBlend4Threads), 0, c->stream, c->r_ranges.as<int2>(), order, vals,
                               c->r_rec.as<float4>(), W, H, tiles_x, dropped_dev,
                               c->r_pre.as<int>(), (long long)c->rn, consumed_dev, c->r_image.as<float4>(), sat, first_phase, last_phase,
                               bins_x, tiles_y);
    } else if (c->opt_blend_pk2) {
        hipLaunchKernelGGL(blend2_kernel, dim3(ntiles), dim3(kBlend2Threads), 0, c->stream, c->r_ranges.as<int2>(), order, vals,
                           c->r_rec.as<float4>(), W, H, tiles_x, dropped_dev,
                           c->r_pre.as<int>(), (long long)c->rn, consumed_dev, c->r_image.as<float4>(), sat, first_phase, last_phase);
    } else {
        hipLaunchKernelGGL(blend_kernel, dim3(ntiles), dim3(kBlendThreads), 0, c->stream, c->r_ranges.as<int2>(), order, vals,
                           c->r_rec.as<float4>(), W, H, tiles_x, dropped_dev,
                           c->r_pre.as<int>(), (long long)c->rn, consumed_dev, c->r_image.as<float4>(), sat, first_phase, last_phase);
    }
    GSX_HIP(c, hipGetLastError());
    return GSX_OK;
}

}  // namespace gsx
