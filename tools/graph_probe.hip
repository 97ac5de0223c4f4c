// GPU box: what a hipGraph replay of a frame-like chain of small dependent kernels costs against plain stream launches.
//   hipcc --offload-arch=gfx950 -O2 tools/graph_probe.hip -o /tmp/graph_probe && /tmp/graph_probe
// 40 dependent launches of a kernel that does ~nothing (the rasterizer's frame is ~35 launches, most of them far too small to
// fill the GPU), one stream and four streams (four host threads are NOT used here: one thread issues everything).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void tiny(unsigned* p, int k) { if (threadIdx.x == 0 && blockIdx.x == 0) p[k & 63] += 1u; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
    const int L = 40, S = 4, reps = 200;
    unsigned* d; CK(hipMalloc(&d, 4096)); CK(hipMemset(d, 0, 4096));
    hipStream_t st[S]; for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto us = [](auto a, auto b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    for (int streams : {1, 4}) {
        // plain launches
        for (int warm = 0; warm < 2; ++warm) {
            auto t0 = now();
            for (int r = 0; r < reps; ++r)
                for (int s = 0; s < streams; ++s) {
                    for (int k = 0; k < L; ++k) hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, st[s], d + 64 * s, k);
                }
            for (int s = 0; s < streams; ++s) CK(hipStreamSynchronize(st[s]));
            if (warm) printf("streams %d: plain launches  %.1f us per %d-launch chain (%.2f us per launch)\n", streams, us(t0, now()) / (reps * streams), L, us(t0, now()) / (reps * streams * L));
        }
        // the same chain as a graph per stream
        std::vector<hipGraphExec_t> ex(streams);
        for (int s = 0; s < streams; ++s) {
            hipGraph_t g;
            CK(hipStreamBeginCapture(st[s], hipStreamCaptureModeThreadLocal));
            for (int k = 0; k < L; ++k) hipLaunchKernelGGL(tiny, dim3(64), dim3(256), 0, st[s], d + 64 * s, k);
            CK(hipStreamEndCapture(st[s], &g));
            CK(hipGraphInstantiate(&ex[s], g, nullptr, nullptr, 0));
            CK(hipGraphDestroy(g));
        }
        for (int warm = 0; warm < 2; ++warm) {
            auto t0 = now();
            for (int r = 0; r < reps; ++r)
                for (int s = 0; s < streams; ++s) CK(hipGraphLaunch(ex[s], st[s]));
            for (int s = 0; s < streams; ++s) CK(hipStreamSynchronize(st[s]));
            if (warm) printf("streams %d: graph replay    %.1f us per %d-launch chain (%.2f us per launch)\n", streams, us(t0, now()) / (reps * streams), L, us(t0, now()) / (reps * streams * L));
        }
        for (auto gx : ex) CK(hipGraphExecDestroy(gx));
    }
    return 0;
}
