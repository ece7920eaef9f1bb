// mfma_f64_bench.hip -- measures the fp64 ceilings the pruning kernel is priced against on gfx950:
//   (a) v_mfma_f64_16x16x4_f64 issue rate, 1 and 2 waves per SIMD
//   (b) v_fma_f64 (VALU) rate
//   (c) both at once: MFMA waves and VALU waves sharing every SIMD; and one wave interleaving both
// build: hipcc --offload-arch=gfx950 -O3 tools/mfma_f64_bench.hip -o tools/_build/mfma_f64_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double double4_t __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); return 1; } } while (0)

// mode bit0: this wave issues MFMA, bit1: this wave issues VALU FMA
__device__ __forceinline__ void body(int iters, bool do_mfma, bool do_valu, double seed, double* sink)
{
    double4_t acc[8];
    double f[16];
    for (int i = 0; i < 8; ++i) acc[i] = double4_t{seed, seed, seed, seed};
    for (int i = 0; i < 16; ++i) f[i] = seed + i;
    double a = seed * 0.5, b = 1.0 - seed * 1e-9;
    if (do_mfma && do_valu) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
                f[2 * i] = __builtin_fma(f[2 * i], b, a);
                f[2 * i + 1] = __builtin_fma(f[2 * i + 1], b, a);
            }
        }
    } else if (do_mfma) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        }
    } else if (do_valu) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) f[i] = __builtin_fma(f[i], b, a);
        }
    }
    double s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 16; ++i) s += f[i];
    if (s == 12345.678) *sink = s;
}

// split: 0 = every wave does `mode`; 1 = waves 0..3 MFMA, waves 4..7 VALU
__global__ __launch_bounds__(512) void k(int iters, int mode, int split, double seed, double* sink)
{
    const int wave = threadIdx.x >> 6;
    bool m = mode & 1, v = mode & 2;
    if (split) { m = wave < 4; v = wave >= 4; }
    body(iters, m, v, seed, sink);
}

static int run(const char* name, int waves_per_block, int mode, int split, int iters, double mfma_per_iter_wave, double fma_per_iter_wave)
{
    double* sink;
    CHECK(hipMalloc(&sink, 8));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int blocks = 256;
    hipLaunchKernelGGL(k, dim3(blocks), dim3(waves_per_block * 64), 0, 0, 1000, mode, split, 0.25, sink);  // warm-up
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k, dim3(blocks), dim3(waves_per_block * 64), 0, 0, iters, mode, split, 0.25, sink);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double waves = double(blocks) * waves_per_block;
    double mfma_waves = waves, valu_waves = waves;
    if (split) { mfma_waves = waves / 2; valu_waves = waves / 2; }
    const double mfma_flops = (mode & 1 || split) ? mfma_waves * iters * mfma_per_iter_wave * 2048.0 : 0.0;  // 2*16*16*4
    const double valu_flops = (mode & 2 || split) ? valu_waves * iters * fma_per_iter_wave * 128.0 : 0.0;   // 64 lanes * 2
    printf("%-44s %8.3f ms  mfma %7.2f TF/s  valu %7.2f TF/s  sum %7.2f TF/s\n", name, best, mfma_flops / best * 1e-9,
           valu_flops / best * 1e-9, (mfma_flops + valu_flops) / best * 1e-9);
    // cycles per MFMA per SIMD assuming 2.4 GHz: time * 2.4e6 / (mfma per SIMD)
    if (mfma_flops > 0) {
        const double mfma_per_simd = mfma_waves * iters * mfma_per_iter_wave / (256.0 * 4.0);
        printf("    -> %.1f ns per MFMA per SIMD (= %.1f cycles at 2.4 GHz)\n", best * 1e6 / mfma_per_simd, best * 2.4e6 / mfma_per_simd);
    }
    CHECK(hipFree(sink));
    return 0;
}

int main()
{
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    printf("device: %s (%s), %d CUs, clock %d MHz\n", p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000);
    const int it = 20000;
    run("MFMA f64 16x16x4, 1 wave/SIMD", 4, 1, 0, it, 8, 0);
    run("MFMA f64 16x16x4, 2 waves/SIMD", 8, 1, 0, it, 8, 0);
    run("VALU v_fma_f64, 1 wave/SIMD", 4, 2, 0, it, 0, 16);
    run("VALU v_fma_f64, 2 waves/SIMD", 8, 2, 0, it, 0, 16);
    run("MFMA waves + VALU waves on each SIMD", 8, 0, 1, it, 8, 16);
    run("one wave interleaving MFMA and 2 FMA each", 4, 3, 0, it, 8, 16);
    run("two waves/SIMD interleaving MFMA and 2 FMA each", 8, 3, 0, it, 8, 16);
    return 0;
}
