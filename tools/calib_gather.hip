// Calibration of rocprofv3's FETCH_SIZE / TCC counters for the access pattern of the vote kernel:
// 1-byte gathers, one distinct 128-byte line per lane, from a buffer far larger than the 256 MiB
// Infinity Cache (MI355X_MICROARCH.md: "Other access widths are uncalibrated: calibrate on a known
// byte count in your own access pattern").  Also runs a 16 B/lane streaming read of the same buffer.
//   hipcc --offload-arch=gfx950 -O3 tools/calib_gather.hip -o tools/calib_gather
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d out -- tools/calib_gather
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e = (x);                                                           \
        if (e != hipSuccess) {                                                        \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));               \
            std::exit(1);                                                             \
        }                                                                             \
    } while (0)

// every lane touches its own 128-B line exactly once (a bijection on [0, lines)): lines*1 B useful
__global__ void gather_lines_kernel(const unsigned char* __restrict__ buf, unsigned long long lines,
                                    unsigned long long mult, unsigned magic, unsigned* __restrict__ sink) {
    const unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= lines) return;
    const unsigned long long line = (i * mult) % lines;  // mult odd and lines a power of two -> permutation
    const unsigned v = buf[line * 128ull + (i & 127ull)];
    if (v == magic) sink[0] = v;  // magic is a run-time value the buffer never holds: keeps the load alive
}

__global__ void stream_kernel(const uint4* __restrict__ buf, unsigned long long n16, unsigned magic,
                              unsigned* __restrict__ sink) {
    unsigned acc = 0;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16;
         i += (unsigned long long)gridDim.x * blockDim.x) {
        const uint4 v = buf[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == magic) sink[0] = acc;
}

int main() {
    const unsigned long long bytes = 2ull << 30;  // 2 GiB
    const unsigned long long lines = bytes / 128;  // 16 Mi lines (power of two)
    unsigned char* buf;
    unsigned* sink;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc(&sink, 4));
    CHECK(hipMemset(buf, 1, bytes));
    CHECK(hipDeviceSynchronize());
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(gather_lines_kernel, dim3((unsigned)(lines / 256)), dim3(256), 0, 0, buf, lines, 2654435761ull, 7u, sink);
        CHECK(hipDeviceSynchronize());
        hipLaunchKernelGGL(stream_kernel, dim3(4096), dim3(256), 0, 0, reinterpret_cast<const uint4*>(buf), bytes / 16, 0x12345u, sink);
        CHECK(hipDeviceSynchronize());
    }
    std::printf("gather_lines_kernel: %llu distinct 128-B lines = %llu bytes of lines, %llu useful bytes\n", lines, lines * 128,
                lines);
    std::printf("stream_kernel: %llu bytes\n", bytes);
    return 0;
}
