// mfma_gap_bench.hip -- two waves per SIMD, each alternating a 64-MFMA "product" with a gap of
// non-matrix work; does the partner's MFMA stream fill the gap?  How does the gap's content matter?
//   gap kind 0: s_sleep (no issue at all)       1: 48 integer VALU (v_mov/v_add)       2: 16 v_mul_f64 + 32 VALU
//   offset: waves 4..7 start `offset` x 1024 cycles late
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_gap_bench.hip -o tools/_build/mfma_gap_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int KIND>
__global__ __launch_bounds__(512) void k(int iters, int offset, int gap_units, double seed, double* sink)
{
    const int wave = threadIdx.x >> 6;
    if (wave >= 4) for (int i = 0; i < offset; ++i) __builtin_amdgcn_s_sleep(16);
    double4_t acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = double4_t{seed, seed, seed, seed};
    double f[16];
    int g[32];
    for (int i = 0; i < 16; ++i) f[i] = seed + i;
    for (int i = 0; i < 32; ++i) g[i] = threadIdx.x + i;
    double a = seed * 0.5, b = 1.0 - seed * 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[t], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        for (int u = 0; u < gap_units; ++u) {
            if (KIND == 0) __builtin_amdgcn_s_sleep(4);  // 256 cycles
            if (KIND == 1) {
#pragma unroll
                for (int i = 0; i < 32; ++i) g[i] = g[i] * 3 + i;
            }
            if (KIND == 2) {
#pragma unroll
                for (int i = 0; i < 16; ++i) f[i] = f[i] * b;
#pragma unroll
                for (int i = 0; i < 16; ++i) g[i] = g[i] * 3 + i;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    double s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 16; ++i) s += f[i];
    int gs = 0;
    for (int i = 0; i < 32; ++i) gs += g[i];
    if (s == 12345.678 || gs == 123456789) *sink = s + gs;
}

template <int KIND>
static int run(const char* name, int waves, int offset, int gap_units)
{
    double* sink;
    CHECK(hipMalloc(&sink, 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int iters = 2000;
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(waves * 64), 0, 0, 50, offset, gap_units, 0.25, sink);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(waves * 64), 0, 0, iters, offset, gap_units, 0.25, sink);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double mfma_per_simd = double(waves / 4) * iters * 64;
    const double cycles = best * 2.4e6;  // nominal clock
    printf("%-28s waves/SIMD %d offset %2d gap_units %2d: %8.3f ms  pipe utilisation %.3f  (cycles per product incl. gap, per wave: %.0f)\n", name,
           waves / 4, offset, gap_units, best, mfma_per_simd * 64.5 / cycles, cycles / iters);
    CHECK(hipFree(sink));
    return 0;
}

int main()
{
    for (int gu : {4, 8}) {
        run<0>("gap = s_sleep", 4, 0, gu);
        run<0>("gap = s_sleep", 8, 0, gu);
        run<0>("gap = s_sleep", 8, 2, gu);
        run<0>("gap = s_sleep", 8, 4, gu);
    }
    for (int gu : {1, 2, 4}) {
        run<1>("gap = 32 int VALU x units", 4, 0, gu);
        run<1>("gap = 32 int VALU x units", 8, 0, gu);
        run<1>("gap = 32 int VALU x units", 8, 2, gu);
        run<1>("gap = 32 int VALU x units", 8, 4, gu);
    }
    for (int gu : {1, 2, 4}) {
        run<2>("gap = 16 f64 mul + 16 int", 4, 0, gu);
        run<2>("gap = 16 f64 mul + 16 int", 8, 0, gu);
        run<2>("gap = 16 f64 mul + 16 int", 8, 2, gu);
        run<2>("gap = 16 f64 mul + 16 int", 8, 4, gu);
    }
    return 0;
}
