// Issue-rate probe for the fp64 / integer VALU instructions the vote kernel is made of (gfx950).
// Each kernel runs 8 independent chains of ONE instruction per lane; 8 waves per SIMD hide latency, so
// time * clock / instructions-per-SIMD = issue cycles per wave instruction.  Reported relative to v_add_u32.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ __launch_bounds__(256) void probe(double* out, int iters, double seed) {
    double a[8];
    for (int k = 0; k < 8; ++k) a[k] = seed + k + threadIdx.x * 1e-3;
    double b = seed * 1.000001, c = seed * 0.5;
    unsigned u[8];
    for (int k = 0; k < 8; ++k) u[k] = threadIdx.x + k;
    unsigned sg[8] = {1, 2, 3, 4, 5, 6, 7, 8};
    const unsigned lanesel = (unsigned)iters & 63u;
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    v4u q4[2] = {};
    __shared__ unsigned char lds_probe[1024];
    if (OP == 24) lds_probe[threadIdx.x] = (unsigned char)threadIdx.x;
    const unsigned ldsaddr = (unsigned)(size_t)lds_probe + (((unsigned)iters & 3u) << 4);  // wave-uniform address: a broadcast read
    __syncthreads();
    for (int i = 0; i < iters; ++i) {
#define ONE(k)                                                                                                   \
    if (OP == 0) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 7]));                          \
    if (OP == 1) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));                           \
    if (OP == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[k]) : "v"(b));                                       \
    if (OP == 3) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[k]) : "v"(b));                                       \
    if (OP == 4) asm volatile("v_rcp_f64 %0, %0" : "+v"(a[k]));                                                    \
    if (OP == 5) asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(a[k]) : "v"(b) : "vcc");                \
    if (OP == 6) asm volatile("v_div_fmas_f64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c) : "vcc");              \
    if (OP == 7) asm volatile("v_div_fixup_f64 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));                     \
    if (OP == 8) asm volatile("v_cmp_lt_f64 vcc, %0, %1" : : "v"(a[k]), "v"(b) : "vcc");                           \
    if (OP == 9) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(u[k]) : "v"(a[k]));                                    \
    if (OP == 10) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(u[k]) : "v"(u[(k + 1) & 7]), "v"(u[(k + 2) & 7])); \
    if (OP == 11) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(u[k]) : "v"(u[(k + 1) & 7]), "v"(u[(k + 2) & 7]));    \
    if (OP == 12) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(a[k]) : "s"(seed), "v"(c));                          \
    if (OP == 13) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a[k]) : "v"(b));                              \
    if (OP == 14) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(u[k]) : "v"(u[(k + 1) & 7]), "v"(u[(k + 2) & 7])); \
    if (OP == 15) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(b), "v"(c));                        \
    if (OP == 16) asm volatile("v_rcp_f32 %0, %0" : "+v"(u[k]));                                                    \
    if (OP == 17) asm volatile("v_readlane_b32 %0, %1, %2" : "=s"(sg[k]) : "v"(u[k]), "s"(lanesel));                \
    if (OP == 18) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(u[k]) : "v"(a[k]));                                    \
    if (OP == 19) asm volatile("v_fract_f32 %0, %0" : "+v"(u[k]));                                                  \
    if (OP == 20) asm volatile("v_cvt_flr_i32_f32 %0, %0" : "+v"(u[k]));                                            \
    if (OP == 21) asm volatile("v_cmp_le_f32 vcc, %0, %1" : : "v"(u[k]), "v"(u[(k + 1) & 7]) : "vcc");              \
    if (OP == 22) asm volatile("v_max_f32 %0, |%0|, |%1|" : "+v"(u[k]) : "v"(u[(k + 1) & 7]));                      \
    if (OP == 23) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[k]) : "s"(seed));                                 \
    if (OP == 24) asm volatile("ds_read_b128 %0, %1" : "=v"(q4[k & 1]) : "v"(ldsaddr) : "memory");                   \
    if (OP == 25) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[k]) : "v"(u[k]));                                    \
    if (OP == 26) asm volatile("v_fract_f64 %0, %0" : "+v"(a[k]));                                                  \
    if (OP == 27) asm volatile("v_mov_b32 %0, %1" : "=v"(u[k]) : "s"(sg[k]));                                       \
    if (OP == 28) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(b));                                    \
    if (OP == 29) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 7]));
        REP8(ONE)
#undef ONE
    }
    double s = 0;
    if (OP == 24) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    for (int k = 0; k < 8; ++k) s += a[k] + u[k] + sg[k];
    s += q4[0].x + q4[1].y;
    if (s == 1.2345e300) out[0] = s;
}

template <int OP>
static double run(const char* name, double base) {
    double* out;
    hipMalloc(&out, 8);
    const int iters = 4000;
    const int blocks = 256 * 8;  // 8 workgroups of 4 waves per CU = 8 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    probe<OP><<<blocks, 256>>>(out, 100, 1.5);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<OP><<<blocks, 256>>>(out, iters, 1.5);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double per_simd = (double)iters * 8 * 8;  // wave instructions per SIMD
    const double ns = ms * 1e6 / per_simd;
    printf("%-18s %8.3f ms  %7.3f ns/wave-instr/SIMD  x%.2f of v_add_u32\n", name, ms, ns, base > 0 ? ns / base : 1.0);
    hipFree(out);
    return ns;
}

int main() {
    const double b = run<0>("v_add_u32", 0);
    run<11>("v_fma_f32", b);
    run<1>("v_fma_f64", b);
    run<12>("v_fmac_f64 (sgpr)", b);
    run<2>("v_mul_f64", b);
    run<3>("v_add_f64", b);
    run<4>("v_rcp_f64", b);
    run<5>("v_div_scale_f64", b);
    run<6>("v_div_fmas_f64", b);
    run<7>("v_div_fixup_f64", b);
    run<8>("v_cmp_lt_f64", b);
    run<9>("v_cvt_i32_f64", b);
    run<10>("v_mad_u32_u24", b);
    run<13>("v_lshl_add_u64", b);
    run<14>("v_or3_b32", b);
    run<29>("v_mul_f32", b);
    run<15>("v_pk_fma_f32", b);
    run<23>("v_pk_mul_f32 (sgpr)", b);
    run<28>("v_pk_add_f32", b);
    run<16>("v_rcp_f32", b);
    run<17>("v_readlane_b32", b);
    run<27>("v_mov_b32 (sgpr)", b);
    run<18>("v_cvt_f32_f64", b);
    run<25>("v_cvt_f64_f32", b);
    run<19>("v_fract_f32", b);
    run<26>("v_fract_f64", b);
    run<20>("v_cvt_flr_i32_f32", b);
    run<21>("v_cmp_le_f32", b);
    run<22>("v_max_f32 |a|,|b|", b);
    run<24>("ds_read_b128 bcast", b);
    return 0;
}
