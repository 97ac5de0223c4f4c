// sort.hip — stable LSD radix sort of (u32 key, u32 value) pairs for gfx950, 8 bits per pass.
//
// Used for (a) the one-off Morton ordering of the Gaussians at upload (vote locality) and
// (b) the per-view (tile | depth16) ordering of the rasterizer, which restates the stable
// 16-bit counting sort of the reference viewer (gaussians_selection.js:417-462).
//
// Per pass, three launches:
//   radix_hist     tile (4096 keys) histograms           -> hist[digit][tile]
//   radix_rowscan  one workgroup per digit: exclusive scan along the tiles, row total -> rowsum
//   radix_scatter  wave-ballot multi-split: every wave ranks its 64 keys per round with 8
//                  __ballot()s (peers = lanes holding the same digit), LDS holds the per-wave digit
//                  counters (stable by construction: tile, wave, round, lane order); the tile is then
//                  reordered through LDS so that consecutive lanes store consecutive elements of one
//                  digit's run
// All loads are coalesced 4 B/lane; the stores are coalesced runs of equal digits.
#include <hip/hip_runtime.h>

#include <algorithm>

#include <cmath>
#include <algorithm>

#include "gsx_ctx.hpp"

namespace gsx {

static constexpr int kSortBlock = 256;
static constexpr int kSortWaves = kSortBlock / 64;
#ifndef GSX_SORT_ITEMS
#define GSX_SORT_ITEMS 16
#endif
static constexpr int kSortItems = GSX_SORT_ITEMS;              // keys per thread
static constexpr int kSortTile = kSortBlock * kSortItems;      // 4096 keys per workgroup (8192 measured 18 % slower; round 3, rasterizer with four
                                                               // frames in flight, variant libraries: 2048 keys -1.5 %, 1024 keys -6 %: profiles/r03/sort_items_ab.txt)
static constexpr int kWaveChunk = 64 * kSortItems;             // 1024 consecutive keys per wave
static constexpr uint32_t kSortDropKey = 0xffffffffu;          // radix_sort_pairs_drop: an element with this key is left out

// Element counts may live on the device (n_dev != nullptr: the rasterizer's pair count of the current depth phase, which
// the host never waits for): the launch is then sized for the CAPACITY n, workgroups past the actual count leave at
// once, and `ntiles` is only the row pitch of the histogram table.
__device__ __forceinline__ long long actual_count(long long n, const unsigned long long* __restrict__ n_dev) {
    if (!n_dev) return n;
    const unsigned long long v = *n_dev;
    return v > (unsigned long long)n ? 0 : (long long)v;  // over capacity: the caller redoes the frame with larger buffers
}

// DROP (first pass of the rasterizer's level-1 sort): elements whose key is kSortDropKey take no part - they are not counted
// and not scattered, and the pass's scatter kernel leaves the number of elements that remain in *n_out, which the later passes
// read as their count.  A stable LSD pass IS an order-preserving partition: leaving the splats no tile will ever see out of
// the sort costs nothing extra.
template <bool DROP>
__global__ __launch_bounds__(kSortBlock) void radix_hist_kernel(const uint32_t* __restrict__ keys, long long n,
                                                                 const unsigned long long* __restrict__ n_dev, int shift,
                                                                 uint32_t mask, uint32_t* __restrict__ hist, int ntiles) {
    __shared__ uint32_t h[256];
    n = actual_count(n, n_dev);
    if ((long long)blockIdx.x * kSortTile >= n) return;
    h[threadIdx.x] = 0;
    __syncthreads();
    const long long base = (long long)blockIdx.x * kSortTile;
#pragma unroll 4
    for (int k = 0; k < kSortItems; ++k) {
        const long long i = base + (long long)k * kSortBlock + threadIdx.x;
        if (i < n) {
            const uint32_t k = keys[i];
            if (!DROP || k != kSortDropKey) atomicAdd(&h[(k >> shift) & mask], 1u);
        }
    }
    __syncthreads();
    hist[(long long)threadIdx.x * ntiles + blockIdx.x] = h[threadIdx.x];
}

// block d: exclusive scan of hist[d][0..ntiles) in place; rowsum[d] = total
__global__ __launch_bounds__(kSortBlock) void radix_rowscan_kernel(uint32_t* __restrict__ hist, int ntiles, long long n,
                                                                    const unsigned long long* __restrict__ n_dev,
                                                                    uint32_t* __restrict__ rowsum) {
    __shared__ uint32_t wsum[kSortWaves];
    __shared__ uint32_t carry_s;
    uint32_t* row = hist + (long long)blockIdx.x * ntiles;
    ntiles = (int)((actual_count(n, n_dev) + kSortTile - 1) / kSortTile);  // tiles in use; the pitch stays the launch's
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int t0 = 0; t0 < ntiles; t0 += kSortBlock) {
        const int t = t0 + threadIdx.x;
        const uint32_t v = t < ntiles ? row[t] : 0u;
        uint32_t inc = v;  // inclusive wave scan
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(inc, o);
            if (lane >= o) inc += up;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint32_t off = carry_s;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        if (t < ntiles) row[t] = off + inc - v;
        __syncthreads();
        if (threadIdx.x == kSortBlock - 1) carry_s = off + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) rowsum[blockIdx.x] = carry_s;
}

template <bool DROP>
__global__ __launch_bounds__(kSortBlock) void radix_scatter_kernel(const uint32_t* __restrict__ keys_in,
                                                                    const uint32_t* __restrict__ vals_in,
                                                                    uint32_t* __restrict__ keys_out,
                                                                    uint32_t* __restrict__ vals_out, long long n,
                                                                    const unsigned long long* __restrict__ n_dev, int shift,
                                                                    uint32_t mask, const uint32_t* __restrict__ hist,
                                                                    const uint32_t* __restrict__ rowsum, int ntiles,
                                                                    unsigned long long* __restrict__ n_out) {
    __shared__ uint32_t tile_kept;                // elements of this tile that take part (DROP: without the dropped ones)
    __shared__ uint32_t gbase[256];               // global start of this tile's run of each digit
    __shared__ uint32_t dstart[256];              // start of each digit inside the tile's locally sorted order
    __shared__ uint32_t wcount[kSortWaves][256];  // per-wave digit counters, then tile-local exclusive bases
    __shared__ uint32_t wsum[kSortWaves];
    __shared__ uint32_t sk[kSortTile];            // the tile, locally sorted by digit (stable)
    __shared__ uint32_t sv[kSortTile];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int d = threadIdx.x;
    n = actual_count(n, n_dev);
    if ((long long)blockIdx.x * kSortTile >= n) return;  // workgroup-uniform

    // digit bases: exclusive scan of rowsum over the 256 digits + this tile's offset inside the digit
    {
        const uint32_t v = rowsum[d];
        uint32_t inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(inc, o);
            if (lane >= o) inc += up;
        }
        if (lane == 63) wsum[wave] = inc;
#pragma unroll
        for (int w = 0; w < kSortWaves; ++w) wcount[w][d] = 0;
        __syncthreads();
        uint32_t off = 0;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        gbase[d] = off + inc - v + hist[(long long)d * ntiles + blockIdx.x];
        if (DROP && blockIdx.x == 0 && d == kSortBlock - 1) *n_out = off + inc;  // all digits' totals: the elements that remain
    }
    __syncthreads();

    const long long tile_base = (long long)blockIdx.x * kSortTile;
    const long long wbase_idx = tile_base + (long long)wave * kWaveChunk;
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t key[kSortItems];
    uint32_t val[kSortItems];
    uint32_t rank[kSortItems];
#pragma unroll
    for (int r = 0; r < kSortItems; ++r) {
        const long long i = wbase_idx + r * 64 + lane;
        bool valid = i < n;
        key[r] = valid ? keys_in[i] : 0u;
        val[r] = valid ? vals_in[i] : 0u;
        if (DROP) valid = valid && key[r] != kSortDropKey;
        const uint32_t dg = (key[r] >> shift) & mask;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const unsigned long long m = __ballot((dg >> b) & 1u);
            peers &= ((dg >> b) & 1u) ? m : ~m;
        }
        const uint32_t before = wcount[wave][dg];
        rank[r] = before + (uint32_t)__popcll(peers & lt);
        // the wave runs in lockstep: every lane has read `before` before the leader's store issues
        if (valid && (peers & lt) == 0ull) wcount[wave][dg] = before + (uint32_t)__popcll(peers);
    }
    __syncthreads();
    {
        // tile-local exclusive start of digit d (scan over the digits of the tile's totals), then per-wave bases
        uint32_t tot = 0;
#pragma unroll
        for (int w = 0; w < kSortWaves; ++w) tot += wcount[w][d];
        uint32_t inc = tot;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(inc, o);
            if (lane >= o) inc += up;
        }
        __syncthreads();  // wsum is reused
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint32_t off = 0;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        uint32_t run = off + inc - tot;
        dstart[d] = run;
        if (d == kSortBlock - 1) tile_kept = off + inc;
#pragma unroll
        for (int w = 0; w < kSortWaves; ++w) {
            const uint32_t c = wcount[w][d];
            wcount[w][d] = run;
            run += c;
        }
    }
    __syncthreads();
    // local reorder through LDS, so that the global stores below are coalesced runs of one digit
#pragma unroll
    for (int r = 0; r < kSortItems; ++r) {
        const long long i = wbase_idx + r * 64 + lane;
        if (i < n && (!DROP || key[r] != kSortDropKey)) {
            const uint32_t dg = (key[r] >> shift) & mask;
            const uint32_t lpos = wcount[wave][dg] + rank[r];
            sk[lpos] = key[r];
            sv[lpos] = val[r];
        }
    }
    __syncthreads();
    const int count = (int)tile_kept;  // (= min(kSortTile, n - tile_base) when nothing is dropped)
    for (int j = threadIdx.x; j < count; j += kSortBlock) {
        const uint32_t k = sk[j];
        const uint32_t dg = (k >> shift) & mask;
        const uint32_t pos = gbase[dg] + ((uint32_t)j - dstart[dg]);
        keys_out[pos] = k;
        vals_out[pos] = sv[j];
    }
}

// Sorts n pairs by key bits [0, bits).  Ping-pongs between (k0,v0) and (k1,v1); *result_in is 0 or 1.
int radix_sort_pairs(Ctx* c, uint32_t* k0, uint32_t* v0, uint32_t* k1, uint32_t* v1, long long n, int bits,
                     int* result_in) {
    return radix_sort_pairs_dev(c, k0, v0, k1, v1, n, nullptr, bits, result_in);
}

int radix_sort_pairs_dev(Ctx* c, uint32_t* k0, uint32_t* v0, uint32_t* k1, uint32_t* v1, long long n,
                         const unsigned long long* n_dev, int bits, int* result_in) {
    return radix_sort_pairs_drop(c, k0, v0, k1, v1, n, n_dev, bits, result_in, nullptr);
}

// n: element count, or the buffers' capacity when the count is read from n_dev on the device (see actual_count).
// n_kept != nullptr: the elements whose key is kSortDropKey are left out by the first pass (see radix_hist_kernel) and
// *n_kept (device) receives the number of the others, which end up sorted in the first *n_kept slots of the result.
int radix_sort_pairs_drop(Ctx* c, uint32_t* k0, uint32_t* v0, uint32_t* k1, uint32_t* v1, long long n,
                          const unsigned long long* n_dev, int bits, int* result_in, unsigned long long* n_kept) {
    *result_in = 0;
    if (n_kept && (n_dev || n <= 0 || bits <= 0)) return fail(c, GSX_E_INVALID, "radix_sort_pairs_drop: needs a host count and at least one pass");
    if ((n <= 1 && !n_dev && !n_kept) || n <= 0 || bits <= 0) return GSX_OK;
    const int ntiles = (int)((n + kSortTile - 1) / kSortTile);
    GSX_HIP(c, c->sort_hist.ensure(sizeof(uint32_t) * ((size_t)256 * ntiles + 256)));
    uint32_t* hist = c->sort_hist.as<uint32_t>();
    uint32_t* rowsum = hist + (size_t)256 * ntiles;
    uint32_t *ki = k0, *vi = v0, *ko = k1, *vo = v1;
    int where = 0;
    // passes of (almost) equal width <= 8 bits: 13 bits -> 7+6, 17 -> 6+6+5, 30 -> 8+8+7+7.  Narrower
    // digits mean fewer, longer runs per workgroup in the scatter.
    const int passes = (bits + 7) / 8;
    int shift = 0;
    for (int pass = 0; pass < passes; ++pass) {
        const int width = (bits - shift + (passes - pass) - 1) / (passes - pass);
        const uint32_t mask = (1u << width) - 1u;  // key bits >= `bits` never take part
        {
            ProfScope ps(c, "radix_hist");
            if (n_kept && pass == 0)
                hipLaunchKernelGGL(radix_hist_kernel<true>, dim3(ntiles), dim3(kSortBlock), 0, c->stream, ki, n, n_dev, shift, mask, hist, ntiles);
            else
                hipLaunchKernelGGL(radix_hist_kernel<false>, dim3(ntiles), dim3(kSortBlock), 0, c->stream, ki, n, n_dev, shift, mask, hist, ntiles);
        }
        {
            ProfScope ps(c, "radix_rowscan");
            hipLaunchKernelGGL(radix_rowscan_kernel, dim3(256), dim3(kSortBlock), 0, c->stream, hist, ntiles, n, n_dev, rowsum);
        }
        {
            ProfScope ps(c, "radix_scatter");
            if (n_kept && pass == 0)
                hipLaunchKernelGGL(radix_scatter_kernel<true>, dim3(ntiles), dim3(kSortBlock), 0, c->stream, ki, vi, ko, vo, n, n_dev,
                                   shift, mask, hist, rowsum, ntiles, n_kept);
            else
                hipLaunchKernelGGL(radix_scatter_kernel<false>, dim3(ntiles), dim3(kSortBlock), 0, c->stream, ki, vi, ko, vo, n, n_dev,
                                   shift, mask, hist, rowsum, ntiles, (unsigned long long*)nullptr);
        }
        if (n_kept && pass == 0) n_dev = n_kept;  // the later passes sort what the first one kept
        GSX_HIP(c, hipGetLastError());
        std::swap(ki, ko);
        std::swap(vi, vo);
        where ^= 1;
        shift += width;
    }
    *result_in = where;
    return GSX_OK;
}

// ---------------------------------------------------------------------------------------------------
// ONE pass over up to 11 key bits: the rasterizer's pair sort by 32x32-pixel bin (2040 bins at 1080p).
// Two 8-bit-machinery passes (6 + 5 bits) cost two histograms, two row scans, two scatters with a reorder through LDS each, and
// the ranges of equal keys then need a kernel of their own.  Here: a 2048-counter LDS histogram per 4096-key tile, the same row
// scan (one workgroup per digit), and a scatter that ranks with 11 ballots per round and stores the VALUES straight to their
// final slots (a tile holds about two pairs per bin: there are no runs worth staging) - and since the digit is the whole key,
// a digit's global base and total ARE the range of its bin: workgroup 0 writes them out.  The keys are not written at all
// (nothing reads them behind the sort).  Stable by construction: tile, wave (1024 consecutive keys each), round, lane.
// ---------------------------------------------------------------------------------------------------
static constexpr int kWideLog = 11;
static constexpr int kWideDigits = 1 << kWideLog;
static constexpr int kWidePer = kWideDigits / kSortBlock;  // digits per thread in the per-digit steps

__global__ __launch_bounds__(kSortBlock) void wide_hist_kernel(const uint32_t* __restrict__ keys, long long n,
                                                                const unsigned long long* __restrict__ n_dev, uint32_t mask,
                                                                uint32_t* __restrict__ hist, int ntiles) {
    __shared__ uint32_t h[kWideDigits];
    n = actual_count(n, n_dev);
    if ((long long)blockIdx.x * kSortTile >= n) return;
#pragma unroll
    for (int k = 0; k < kWidePer; ++k) h[threadIdx.x + k * kSortBlock] = 0;
    __syncthreads();
    const long long base = (long long)blockIdx.x * kSortTile;
#pragma unroll 4
    for (int k = 0; k < kSortItems; ++k) {
        const long long i = base + (long long)k * kSortBlock + threadIdx.x;
        if (i < n) atomicAdd(&h[keys[i] & mask], 1u);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kWidePer; ++k) {
        const int d = threadIdx.x + k * kSortBlock;
        hist[(long long)d * ntiles + blockIdx.x] = h[d];
    }
}

__global__ __launch_bounds__(kSortBlock) void wide_scatter_kernel(const uint32_t* __restrict__ keys_in,
                                                                   const uint32_t* __restrict__ vals_in,
                                                                   uint32_t* __restrict__ vals_out, long long n,
                                                                   const unsigned long long* __restrict__ n_dev, uint32_t mask,
                                                                   const uint32_t* __restrict__ hist,
                                                                   const uint32_t* __restrict__ rowsum, int ntiles,
                                                                   int2* __restrict__ ranges, int nranges) {
    __shared__ uint32_t gbase[kWideDigits];               // global slot of this tile's first element of each digit
    __shared__ uint32_t wcount[kSortWaves][kWideDigits];  // per-wave digit counters, then the waves' exclusive bases inside the tile's run
    __shared__ uint32_t wsum[kSortWaves];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    n = actual_count(n, n_dev);
    const bool active = (long long)blockIdx.x * kSortTile < n;  // workgroup-uniform
    if (!active && blockIdx.x != 0) return;                      // (workgroup 0 writes the ranges even of an empty sort)
    {
        // digit bases: thread t owns the digits [8 t, 8 t + 8): exclusive scan of rowsum over all digits
        uint32_t rs[kWidePer];
        uint32_t sum = 0;
#pragma unroll
        for (int k = 0; k < kWidePer; ++k) {
            rs[k] = rowsum[threadIdx.x * kWidePer + k];
            sum += rs[k];
        }
        uint32_t inc = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(inc, o);
            if (lane >= o) inc += up;
        }
        if (lane == 63) wsum[wave] = inc;
        __syncthreads();
        uint32_t run = inc - sum;
        for (int w = 0; w < wave; ++w) run += wsum[w];
#pragma unroll
        for (int k = 0; k < kWidePer; ++k) {
            const int d = threadIdx.x * kWidePer + k;
            if (blockIdx.x == 0 && d < nranges) ranges[d] = make_int2((int)run, (int)(run + rs[k]));
            gbase[d] = run + (active ? hist[(long long)d * ntiles + blockIdx.x] : 0u);
            run += rs[k];
#pragma unroll
            for (int w = 0; w < kSortWaves; ++w) wcount[w][d] = 0;
        }
    }
    if (!active) return;
    __syncthreads();
    const long long wbase_idx = (long long)blockIdx.x * kSortTile + (long long)wave * kWaveChunk;
    const unsigned long long lt = (1ull << lane) - 1ull;
    uint32_t dig[kSortItems];
    uint32_t val[kSortItems];
    uint32_t rank[kSortItems];
#pragma unroll
    for (int r = 0; r < kSortItems; ++r) {
        const long long i = wbase_idx + r * 64 + lane;
        const bool valid = i < n;
        const uint32_t dg = valid ? keys_in[i] & mask : 0u;
        val[r] = valid ? vals_in[i] : 0u;
        dig[r] = dg;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < kWideLog; ++b) {
            const unsigned long long m = __ballot((dg >> b) & 1u);
            peers &= ((dg >> b) & 1u) ? m : ~m;
        }
        const uint32_t before = wcount[wave][dg];
        rank[r] = before + (uint32_t)__popcll(peers & lt);
        // the wave runs in lockstep: every lane has read `before` before the leader's store issues
        if (valid && (peers & lt) == 0ull) wcount[wave][dg] = before + (uint32_t)__popcll(peers);
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < kWidePer; ++k) {  // the waves' exclusive bases inside the tile's run of digit d
        const int d = threadIdx.x + k * kSortBlock;
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < kSortWaves; ++w) {
            const uint32_t c = wcount[w][d];
            wcount[w][d] = run;
            run += c;
        }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < kSortItems; ++r) {
        const long long i = wbase_idx + r * 64 + lane;
        if (i < n) vals_out[gbase[dig[r]] + wcount[wave][dig[r]] + rank[r]] = val[r];
    }
}

// Sorts the VALUES of n pairs (count on the device if n_dev) by key bits [0, bits), bits <= 11, in one pass: v0 -> v1; the keys
// are only read.  ranges[d] = [first, last + 1) slot of key d for d < nranges (<= 2048), written whatever the count.
int radix_sort_values_wide(Ctx* c, const uint32_t* k0, const uint32_t* v0, uint32_t* v1, long long n, const unsigned long long* n_dev,
                           int bits, int2* ranges, int nranges) {
    if (n <= 0 || bits < 1 || bits > kWideLog || nranges > kWideDigits || !ranges)
        return fail(c, GSX_E_INVALID, "radix_sort_values_wide: bad arguments");
    const int ntiles = (int)((n + kSortTile - 1) / kSortTile);
    GSX_HIP(c, c->sort_hist.ensure(sizeof(uint32_t) * ((size_t)kWideDigits * ntiles + kWideDigits)));
    uint32_t* hist = c->sort_hist.as<uint32_t>();
    uint32_t* rowsum = hist + (size_t)kWideDigits * ntiles;
    const uint32_t mask = (1u << bits) - 1u;
    {
        ProfScope ps(c, "radix_hist");
        hipLaunchKernelGGL(wide_hist_kernel, dim3(ntiles), dim3(kSortBlock), 0, c->stream, k0, n, n_dev, mask, hist, ntiles);
    }
    {
        ProfScope ps(c, "radix_rowscan");
        hipLaunchKernelGGL(radix_rowscan_kernel, dim3(kWideDigits), dim3(kSortBlock), 0, c->stream, hist, ntiles, n, n_dev, rowsum);
    }
    {
        ProfScope ps(c, "radix_scatter");
        hipLaunchKernelGGL(wide_scatter_kernel, dim3(ntiles), dim3(kSortBlock), 0, c->stream, k0, v0, v1, n, n_dev, mask, hist, rowsum,
                           ntiles, ranges, nranges);
    }
    GSX_HIP(c, hipGetLastError());
    return GSX_OK;
}

// ---------------------------------------------------------------------------------------------------
// Morton order of the positions (performance only: results never depend on the order)
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float finite_or(float v, float alt) { return (v - v == 0.0f) ? v : alt; }

__global__ __launch_bounds__(kSortBlock) void bbox_partial_kernel(const float* __restrict__ x,
                                                                   const float* __restrict__ y,
                                                                   const float* __restrict__ z, long long n,
                                                                   float* __restrict__ partial /*[grid][6]*/) {
    float lo[3] = {3.0e38f, 3.0e38f, 3.0e38f}, hi[3] = {-3.0e38f, -3.0e38f, -3.0e38f};
    for (long long i = (long long)blockIdx.x * kSortBlock + threadIdx.x; i < n; i += (long long)gridDim.x * kSortBlock) {
        const float v[3] = {x[i], y[i], z[i]};
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            lo[a] = fminf(lo[a], finite_or(v[a], 3.0e38f));
            hi[a] = fmaxf(hi[a], finite_or(v[a], -3.0e38f));
        }
    }
    __shared__ float s[kSortWaves][6];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo[a] = fminf(lo[a], __shfl_xor(lo[a], o));
            hi[a] = fmaxf(hi[a], __shfl_xor(hi[a], o));
        }
        if (lane == 0) {
            s[wave][a] = lo[a];
            s[wave][3 + a] = hi[a];
        }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        float r = s[0][threadIdx.x];
        for (int w = 1; w < kSortWaves; ++w) r = threadIdx.x < 3 ? fminf(r, s[w][threadIdx.x]) : fmaxf(r, s[w][threadIdx.x]);
        partial[blockIdx.x * 6 + threadIdx.x] = r;
    }
}

__global__ void bbox_final_kernel(const float* __restrict__ partial, int nparts, float* __restrict__ box) {
    if (threadIdx.x < 6) {
        float r = partial[threadIdx.x];
        for (int p = 1; p < nparts; ++p) {
            const float v = partial[p * 6 + threadIdx.x];
            r = threadIdx.x < 3 ? fminf(r, v) : fmaxf(r, v);
        }
        box[threadIdx.x] = r;
    }
}

// 21 bits -> every third bit of a 63-bit word
__device__ __forceinline__ unsigned long long spread21(unsigned long long v) {
    v &= 0x1fffffull;
    v = (v | (v << 32)) & 0x1f00000000ffffull;
    v = (v | (v << 16)) & 0x1f0000ff0000ffull;
    v = (v | (v << 8)) & 0x100f00f00f00f00full;
    v = (v | (v << 4)) & 0x10c30c30c30c30c3ull;
    return (v | (v << 2)) & 0x1249249249249249ull;
}

// 63-bit Morton code, 21 bits per axis: a scene with far outliers (floaters at 1000x the scene radius are common in
// trained captures) still has ~2000 cells across its dense part, where 10 bits per axis would leave two.
__global__ __launch_bounds__(kSortBlock) void morton_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                             const float* __restrict__ z, long long n,
                                                             const float* __restrict__ box,
                                                             uint32_t* __restrict__ code_lo, uint32_t* __restrict__ code_hi,
                                                             uint32_t* __restrict__ idx) {
    const long long i = (long long)blockIdx.x * kSortBlock + threadIdx.x;
    if (i >= n) return;
    const float v[3] = {x[i], y[i], z[i]};
    unsigned long long q[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double ext = (double)box[3 + a] - (double)box[a];
        double t = ext > 0.0 ? ((double)v[a] - (double)box[a]) / ext * 2097151.0 : 0.0;
        t = (t - t == 0.0) ? fmin(fmax(t, 0.0), 2097151.0) : 0.0;  // non-finite positions sort first
        q[a] = (unsigned long long)t;
    }
    const unsigned long long code = spread21(q[0]) | (spread21(q[1]) << 1) | (spread21(q[2]) << 2);
    code_lo[i] = (uint32_t)code;
    code_hi[i] = (uint32_t)(code >> 32);
    idx[i] = (uint32_t)i;
}

__global__ __launch_bounds__(kSortBlock) void gather1_kernel(const uint32_t* __restrict__ in, const uint32_t* __restrict__ perm,
                                                              long long n, uint32_t* __restrict__ out) {
    const long long i = (long long)blockIdx.x * kSortBlock + threadIdx.x;
    if (i < n) out[i] = in[perm[i]];
}

__global__ __launch_bounds__(kSortBlock) void gather3_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                              const float* __restrict__ z,
                                                              const uint32_t* __restrict__ perm, long long n,
                                                              float* __restrict__ ox, float* __restrict__ oy,
                                                              float* __restrict__ oz) {
    const long long i = (long long)blockIdx.x * kSortBlock + threadIdx.x;
    if (i >= n) return;
    const uint32_t s = perm[i];
    ox[i] = x[s];
    oy[i] = y[s];
    oz[i] = z[s];
}

// Reorders c->x/y/z along a 63-bit Morton curve; c->perm[i] = original index of sorted slot i.
int spatial_sort_positions(Ctx* c) {
    const long long n = c->n;
    c->sorted = false;
    if (n < 2) return GSX_OK;
    const size_t nb = sizeof(uint32_t) * (size_t)n;
    DevBuf code0, code1, codeh, idx1, box, tx, ty, tz;
    GSX_HIP(c, code0.ensure(nb));
    GSX_HIP(c, code1.ensure(nb));
    GSX_HIP(c, codeh.ensure(nb));
    GSX_HIP(c, idx1.ensure(nb));
    GSX_HIP(c, c->perm.ensure(nb));
    const int parts = 512;
    GSX_HIP(c, box.ensure(sizeof(float) * (6 * parts + 6)));
    float* partial = box.as<float>();
    float* bb = partial + 6 * parts;
    const unsigned grid = (unsigned)((n + kSortBlock - 1) / kSortBlock);
    hipLaunchKernelGGL(bbox_partial_kernel, dim3(parts), dim3(kSortBlock), 0, c->stream, c->x.as<float>(), c->y.as<float>(),
                       c->z.as<float>(), n, partial);
    hipLaunchKernelGGL(bbox_final_kernel, dim3(1), dim3(64), 0, c->stream, partial, parts, bb);
    hipLaunchKernelGGL(morton_kernel, dim3(grid), dim3(kSortBlock), 0, c->stream, c->x.as<float>(), c->y.as<float>(),
                       c->z.as<float>(), n, bb, code0.as<uint32_t>(), codeh.as<uint32_t>(), c->perm.as<uint32_t>());
    GSX_HIP(c, hipGetLastError());
    // stable LSD sort of the 63-bit code: by its low word, then by its high word (gathered into the new order)
    uint32_t* idx[2] = {c->perm.as<uint32_t>(), idx1.as<uint32_t>()};
    int where = 0;
    int rc = radix_sort_pairs(c, code0.as<uint32_t>(), idx[0], code1.as<uint32_t>(), idx[1], n, 32, &where);
    if (!rc) {
        uint32_t* sorted_idx = idx[where];
        uint32_t* other_idx = idx[where ^ 1];
        hipLaunchKernelGGL(gather1_kernel, dim3(grid), dim3(kSortBlock), 0, c->stream, codeh.as<uint32_t>(), sorted_idx, n,
                           code0.as<uint32_t>());
        GSX_HIP(c, hipGetLastError());
        int where2 = 0;
        rc = radix_sort_pairs(c, code0.as<uint32_t>(), sorted_idx, code1.as<uint32_t>(), other_idx, n, 31, &where2);
        if (!rc) {
            uint32_t* final_idx = where2 ? other_idx : sorted_idx;
            if (final_idx != c->perm.as<uint32_t>()) {
                hipError_t e = hipMemcpyAsync(c->perm.p, final_idx, nb, hipMemcpyDeviceToDevice, c->stream);
                if (e != hipSuccess) rc = fail(c, GSX_E_HIP, "spatial sort: %s", hipGetErrorString(e));
            }
        }
    }
    if (!rc) {
        hipError_t e = tx.ensure(sizeof(float) * n);
        if (e == hipSuccess) e = ty.ensure(sizeof(float) * n);
        if (e == hipSuccess) e = tz.ensure(sizeof(float) * n);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(gather3_kernel, dim3(grid), dim3(kSortBlock), 0, c->stream, c->x.as<float>(), c->y.as<float>(),
                               c->z.as<float>(), c->perm.as<uint32_t>(), n, tx.as<float>(), ty.as<float>(), tz.as<float>());
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) rc = fail(c, GSX_E_HIP, "spatial sort: %s", hipGetErrorString(e));
        else {
            std::swap(c->x, tx);
            std::swap(c->y, ty);
            std::swap(c->z, tz);
            c->sorted = true;
        }
    }
    (void)hipStreamSynchronize(c->stream);
    for (DevBuf* b : {&code0, &code1, &codeh, &idx1, &box, &tx, &ty, &tz}) b->release();
    return rc;
}

}  // namespace gsx
