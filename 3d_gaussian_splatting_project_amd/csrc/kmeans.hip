// kmeans.hip - the reference's second labeler, 3D_clustering/k_means.py:107-151 `k_means_with_color`
// (SURVEY §8f-4), as HIP kernels for gfx950.  Same `label` field as the majority vote, bit-exact:
//
//   assign   (k_means.py:116-122)  every row (x, y, z, f_dc_0..2) goes to its nearest centroid.  scipy's KD-tree
//            compares float64 SQUARED distances summed as ((d0^2 + d1^2) + d2^2) + d3^2, + d4^2, + d5^2; the kernel
//            evaluates exactly that, without FMA (this file is compiled with -ffp-contract=off), for all k
//            centroids (held in LDS as doubles) and keeps the first minimum.
//   update   (k_means.py:125-128)  numpy's float32 mean over axis 0 adds the member rows IN INDEX ORDER into a float32
//            accumulator.  Floating-point addition does not reassociate, so the sum is made the same way: a stable
//            radix sort of the row indices by label puts every cluster's members in index order, and one wave per
//            workgroup per cluster adds them one after the other (fifteen waves stage the rows through a transposed
//            LDS ring, six lanes - one per dimension - do the adds, four members per LDS read).  The division by float32(count) and the convergence test (:132-136) run on
//            the host.
//
// Data layout in HBM: rows float[n][6] (24 B, AoS as in the PLY), labels u32[n], order u32[n] (row indices grouped by
// label), centroids double[k][6], sums float[k][6], counts u32[k].
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <vector>

#include "gsx_ctx.hpp"

namespace gsx {

static constexpr int kKmBlock = 256;
static constexpr int kKmMaxK = 2048;  // centroids live in LDS: 2048 * 48 B = 96 KB

__global__ __launch_bounds__(kKmBlock) void kmeans_assign_kernel(const float* __restrict__ rows, long long n,
                                                                 const double* __restrict__ centroids, int k,
                                                                 uint32_t* __restrict__ labels, uint32_t* __restrict__ idx,
                                                                 uint32_t* __restrict__ counts) {
    extern __shared__ double cs[];  // [k][6], then u32 counts[k]
    uint32_t* lc = reinterpret_cast<uint32_t*>(cs + (size_t)k * 6);
    for (int t = threadIdx.x; t < k * 6; t += kKmBlock) cs[t] = centroids[t];
    for (int t = threadIdx.x; t < k; t += kKmBlock) lc[t] = 0;
    __syncthreads();
    const long long i = (long long)blockIdx.x * kKmBlock + threadIdx.x;
    if (i < n) {
        const float2* r = reinterpret_cast<const float2*>(rows + i * 6);  // rows are 8-byte aligned (24 B)
        const float2 a = r[0], b = r[1], c = r[2];
        const double p0 = a.x, p1 = a.y, p2 = b.x, p3 = b.y, p4 = c.x, p5 = c.y;
        double best = 0.0;
        int arg = 0;
        for (int j = 0; j < k; ++j) {
            const double* q = cs + j * 6;  // same address in all lanes: LDS broadcast
            const double d0 = p0 - q[0], d1 = p1 - q[1], d2 = p2 - q[2], d3 = p3 - q[3], d4 = p4 - q[4], d5 = p5 - q[5];
            double s = ((d0 * d0 + d1 * d1) + d2 * d2) + d3 * d3;  // scipy's 4-accumulator loop ...
            s = s + d4 * d4;                                        // ... and its scalar tail
            s = s + d5 * d5;
            if (j == 0 || s < best) {  // first minimum; NaN distances never win (as in a < comparison)
                best = s;
                arg = j;
            }
        }
        labels[i] = (uint32_t)arg;
        idx[i] = (uint32_t)i;
        atomicAdd(&lc[arg], 1u);
    }
    __syncthreads();
    for (int t = threadIdx.x; t < k; t += kKmBlock)
        if (lc[t]) atomicAdd(&counts[t], lc[t]);
}

// One workgroup per cluster: sums[c][d] = ((row[o0][d] + row[o1][d]) + row[o2][d]) + ... in float32, members in
// index order.  The chain of additions is serial by definition (it is the kernel's floor: one dependent v_add_f32 per
// member), so everything else is kept out of its way: waves 1..15 fetch the next kSumRows member rows into one half of
// an LDS ring, TRANSPOSED to [dimension][row], while lanes 0..5 of wave 0 (one per dimension) add the previous kSumRows
// from the other half - four members per ds_read_b128, sixteen in flight before their sixteen ordered additions.
static constexpr int kSumThreads = 1024;
static constexpr int kSumLoaders = kSumThreads - 64;
static constexpr int kSumPerLoader = 2;
static constexpr int kSumRows = kSumLoaders * kSumPerLoader;  // 1920 rows per stage: 2 x 6 x 1920 floats = 90 KB of LDS
__global__ __launch_bounds__(kSumThreads) void kmeans_sum_kernel(const float* __restrict__ rows,
                                                                 const uint32_t* __restrict__ order,
                                                                 const uint32_t* __restrict__ offsets /*[k+1]*/,
                                                                 float* __restrict__ sums) {
    extern __shared__ float4 stage_raw[];  // [2][6][kSumRows] floats, 16-byte aligned
    float* stage = reinterpret_cast<float*>(stage_raw);
    const int c = blockIdx.x, tid = threadIdx.x;
    const uint32_t beg = offsets[c], end = offsets[c + 1];
    float acc = 0.0f;
    bool first = true;  // numpy's reduction starts FROM the first row, not from +0.0 (the sign of a zero sum)
    auto load = [&](uint32_t t0, int buf) {  // loader waves only
        if (tid < 64) return;
        float* sb = stage + (size_t)buf * 6 * kSumRows;
#pragma unroll
        for (int q = 0; q < kSumPerLoader; ++q) {
            const int local = (tid - 64) + q * kSumLoaders;
            const uint32_t t = t0 + (uint32_t)local;
            if (t < end) {
                const float2* r = reinterpret_cast<const float2*>(rows + (size_t)order[t] * 6);
                const float2 a = r[0], b = r[1], d = r[2];
                sb[0 * kSumRows + local] = a.x;
                sb[1 * kSumRows + local] = a.y;
                sb[2 * kSumRows + local] = b.x;
                sb[3 * kSumRows + local] = b.y;
                sb[4 * kSumRows + local] = d.x;
                sb[5 * kSumRows + local] = d.y;
            }
        }
    };
    int buf = 0;
    load(beg, 0);
    for (uint32_t t0 = beg; t0 < end; t0 += kSumRows) {  // block-uniform trip count
        __syncthreads();                                  // stage[buf] is complete, stage[buf ^ 1] is free
        if (t0 + kSumRows < end) load(t0 + kSumRows, buf ^ 1);
        if (tid < 6) {
            const float* col = stage + ((size_t)buf * 6 + tid) * kSumRows;  // my dimension of the staged members
            const int m = (int)min((uint32_t)kSumRows, end - t0);
            int j = 0;
            if (first && m > 0) {
                acc = col[0];
                first = false;
                j = 1;
            }
            for (; j < m && (j & 3); ++j) acc = acc + col[j];  // up to a 16-byte boundary
            for (; j + 16 <= m; j += 16) {                      // 4 x ds_read_b128 in flight, then the 16 additions in order
                const float4 v0 = *reinterpret_cast<const float4*>(col + j);
                const float4 v1 = *reinterpret_cast<const float4*>(col + j + 4);
                const float4 v2 = *reinterpret_cast<const float4*>(col + j + 8);
                const float4 v3 = *reinterpret_cast<const float4*>(col + j + 12);
                acc = acc + v0.x; acc = acc + v0.y; acc = acc + v0.z; acc = acc + v0.w;
                acc = acc + v1.x; acc = acc + v1.y; acc = acc + v1.z; acc = acc + v1.w;
                acc = acc + v2.x; acc = acc + v2.y; acc = acc + v2.z; acc = acc + v2.w;
                acc = acc + v3.x; acc = acc + v3.y; acc = acc + v3.z; acc = acc + v3.w;
            }
            for (; j < m; ++j) acc = acc + col[j];
        }
        buf ^= 1;
    }
    if (tid < 6) sums[c * 6 + tid] = acc;
}

// ---------------------------------------------------------------------------------------------------------------
int kmeans(Ctx* c, int64_t n, const float* points, const float* colors, int k, const int64_t* init_index, int max_iter,
           double tol, int32_t* labels_out, float* centroids_out, int32_t* iterations_out, int32_t* converged_out) {
    if (n < 1 || !points || !colors || !init_index || !labels_out) return fail(c, GSX_E_INVALID, "kmeans: bad arguments");
    if (k < 1 || k > kKmMaxK || k > n) return fail(c, GSX_E_RANGE, "kmeans: k must be in [1, min(n, %d)]", kKmMaxK);
    if (n > ((int64_t)1 << 31) - 1) return fail(c, GSX_E_UNSUPPORTED, "kmeans: n > 2^31-1");
    if (max_iter < 0) return fail(c, GSX_E_INVALID, "kmeans: max_iter < 0");
    for (int j = 0; j < k; ++j)
        if (init_index[j] < 0 || init_index[j] >= n) return fail(c, GSX_E_RANGE, "kmeans: init_index[%d] out of range", j);
    GSX_HIP(c, hipSetDevice(c->device));

    // rows = (points | colors), float32 (k_means.py:109)
    std::vector<float> host((size_t)n * 6);
    for (int64_t i = 0; i < n; ++i) {
        std::memcpy(&host[(size_t)i * 6], points + i * 3, 12);
        std::memcpy(&host[(size_t)i * 6 + 3], colors + i * 3, 12);
    }
    DevBuf rows, labels, idx0, keys1, idx1, cent, sums, counts, offsets;
    const size_t nb = sizeof(uint32_t) * (size_t)n;
    GSX_HIP(c, rows.ensure(sizeof(float) * host.size()));
    GSX_HIP(c, labels.ensure(nb));
    GSX_HIP(c, idx0.ensure(nb));
    GSX_HIP(c, keys1.ensure(nb));
    GSX_HIP(c, idx1.ensure(nb));
    GSX_HIP(c, cent.ensure(sizeof(double) * 6 * k));
    GSX_HIP(c, sums.ensure(sizeof(float) * 6 * k));
    GSX_HIP(c, counts.ensure(sizeof(uint32_t) * k));
    GSX_HIP(c, offsets.ensure(sizeof(uint32_t) * (k + 1)));
    GSX_HIP(c, hipMemcpyAsync(rows.p, host.data(), sizeof(float) * host.size(), hipMemcpyHostToDevice, c->stream));

    std::vector<float> centroids((size_t)k * 6), next((size_t)k * 6), hs((size_t)k * 6);
    for (int j = 0; j < k; ++j) std::memcpy(&centroids[(size_t)j * 6], &host[(size_t)init_index[j] * 6], 24);  // :111
    std::vector<double> c64((size_t)k * 6);
    std::vector<uint32_t> hc(k), ho(k + 1);
    const unsigned grid = (unsigned)((n + kKmBlock - 1) / kKmBlock);
    const size_t lds = sizeof(double) * 6 * k + sizeof(uint32_t) * k;
    int bits = 1;
    while ((1 << bits) < k) ++bits;

    auto assign = [&]() -> int {
        for (size_t t = 0; t < c64.size(); ++t) c64[t] = (double)centroids[t];  // the KD-tree holds float64 copies
        GSX_HIP(c, hipMemcpyAsync(cent.p, c64.data(), sizeof(double) * c64.size(), hipMemcpyHostToDevice, c->stream));
        GSX_HIP(c, hipMemsetAsync(counts.p, 0, sizeof(uint32_t) * k, c->stream));
        ProfScope ps(c, "kmeans_assign");
        hipLaunchKernelGGL(kmeans_assign_kernel, dim3(grid), dim3(kKmBlock), lds, c->stream, rows.as<float>(), (long long)n,
                           cent.as<double>(), k, labels.as<uint32_t>(), idx0.as<uint32_t>(), counts.as<uint32_t>());
        GSX_HIP(c, hipGetLastError());
        return GSX_OK;
    };
    if (lds > 64 * 1024)
        GSX_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(kmeans_assign_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));

    GSX_HIP(c, hipFuncSetAttribute(reinterpret_cast<const void*>(kmeans_sum_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)(sizeof(float) * 2 * 6 * kSumRows)));
    int iters = 0, converged = 0;
    for (int it = 0; it < max_iter; ++it) {  // :113
        int rc = assign();
        if (rc) return rc;
        // members of every cluster in index order: stable sort of the indices by label (the labels buffer is the key)
        int where = 0;
        rc = radix_sort_pairs(c, labels.as<uint32_t>(), idx0.as<uint32_t>(), keys1.as<uint32_t>(), idx1.as<uint32_t>(), n, bits,
                              &where);
        if (rc) return rc;
        const uint32_t* order = where ? idx1.as<uint32_t>() : idx0.as<uint32_t>();
        GSX_HIP(c, hipMemcpyAsync(hc.data(), counts.p, sizeof(uint32_t) * k, hipMemcpyDeviceToHost, c->stream));
        GSX_HIP(c, hipStreamSynchronize(c->stream));
        ho[0] = 0;
        for (int j = 0; j < k; ++j) ho[j + 1] = ho[j] + hc[j];
        GSX_HIP(c, hipMemcpyAsync(offsets.p, ho.data(), sizeof(uint32_t) * (k + 1), hipMemcpyHostToDevice, c->stream));
        {
            ProfScope ps(c, "kmeans_sum");
            hipLaunchKernelGGL(kmeans_sum_kernel, dim3(k), dim3(kSumThreads), sizeof(float) * 2 * 6 * kSumRows, c->stream, rows.as<float>(), order,
                               offsets.as<uint32_t>(), sums.as<float>());
            GSX_HIP(c, hipGetLastError());
        }
        GSX_HIP(c, hipMemcpyAsync(hs.data(), sums.p, sizeof(float) * 6 * k, hipMemcpyDeviceToHost, c->stream));
        GSX_HIP(c, hipStreamSynchronize(c->stream));
        double sq = 0.0;
        for (int j = 0; j < k; ++j)
            for (int d = 0; d < 6; ++d) {
                const size_t t = (size_t)j * 6 + d;
                next[t] = hc[j] ? hs[t] / (float)hc[j] : centroids[t];  // float32 mean (:126); empty cluster keeps its centroid
                const float diff = next[t] - centroids[t];              // float32 difference (:132)
                sq += (double)diff * (double)diff;
            }
        ++iters;
        // :132-136.  numpy takes this norm in float32 through BLAS sdot; here it is accumulated in double - the two
        // can only decide differently if the norm is within ~1e-6 (relative) of tol
        if (std::sqrt(sq) < tol) {
            converged = 1;
            break;
        }
        centroids = next;  // :138
    }
    int rc = assign();  // :142-147 final labels from the centroids in hand
    if (rc) return rc;
    GSX_HIP(c, hipMemcpyAsync(labels_out, labels.p, nb, hipMemcpyDeviceToHost, c->stream));
    GSX_HIP(c, hipStreamSynchronize(c->stream));
    if (centroids_out) std::memcpy(centroids_out, centroids.data(), sizeof(float) * centroids.size());
    if (iterations_out) *iterations_out = iters;
    if (converged_out) *converged_out = converged;
    return GSX_OK;
}

}  // namespace gsx
