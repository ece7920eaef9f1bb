// two_tile_bench.hip -- VERDICT r04 item 7, priced on the hardware before rewriting prune_fixed: the product stream of kernels.hip
// (64 x v_mfma_f64_16x16x4_f64 per 16-column tile and tree edge; the edge's 32 KiB branch matrix double-buffered in LDS by LDS-DMA, two
// ds_read_b128 of fragments per k-step, sixteen 16-byte leaf-row pieces and a sixteen-multiply epilogue per tile and product, one barrier
// per product) in two shapes:
//   A  one tile per wave, TWO waves per SIMD (two 4-wave work-groups per CU; <= 256 registers)          -- today's kernel
//   B  two tiles per wave, ONE wave per SIMD (one work-group per CU; 512 registers): a product is 128 MFMAs that share every fragment
//      read and every LDS-DMA piece; tile 0's epilogue issues in the shadow of tile 1's last MFMAs, tile 1's is exposed
// Same bytes staged per product, same leaf-row traffic per tile.  Prints TFLOP/s of the MFMAs and the fraction of the fp64 peak.
//   hipcc --offload-arch=gfx950 -O3 tools/two_tile_bench.hip -o tools/_build/two_tile_bench && tools/_build/two_tile_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int FRAG_DOUBLES = 4096;   // one branch matrix in fragment order: 32 KiB
constexpr int N_SLOTS = 114;         // 58mammals: 2 x 57 internal edges

template <int G>   // tiles per wave
__global__ __launch_bounds__(256, G == 1 ? 2 : 1) void stream_kernel(const double* __restrict__ frag, const double* __restrict__ leaf_tab, int n_products, double* __restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    double2_t* lfrag = reinterpret_cast<double2_t*>(lds_raw);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g4 = lane >> 4;
    const unsigned char* src_lane = reinterpret_cast<const unsigned char*>(frag) + wave * 8192 + lane * 16;
    auto stage_piece = [&](auto jc, int slot, int buffer) {
        constexpr int J = decltype(jc)::value;
        const unsigned char* src = src_lane + static_cast<size_t>(slot) * (FRAG_DOUBLES * 8) + (J / 4) * 4096;
        unsigned char* dst = lds_raw + buffer * (FRAG_DOUBLES * 8) + wave * 8192 + (J / 4) * 4096;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)dst, 16, (J % 4) * 1024, 0);
    };
#define STAGE8(slot, buffer)                                                                                                        \
    stage_piece(std::integral_constant<int, 0>{}, slot, buffer); stage_piece(std::integral_constant<int, 1>{}, slot, buffer);      \
    stage_piece(std::integral_constant<int, 2>{}, slot, buffer); stage_piece(std::integral_constant<int, 3>{}, slot, buffer);      \
    stage_piece(std::integral_constant<int, 4>{}, slot, buffer); stage_piece(std::integral_constant<int, 5>{}, slot, buffer);      \
    stage_piece(std::integral_constant<int, 6>{}, slot, buffer); stage_piece(std::integral_constant<int, 7>{}, slot, buffer);
    STAGE8(0, 0)
    double cur[G][16];
    double4_t acc[G][4];
#pragma unroll
    for (int t = 0; t < G; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) cur[t][q] = 1.0 + 1e-3 * (lane + q + t);
    // leaf rows: a pseudo-random row of a 65-row table per tile and product, eight 16-byte pieces per lane (stride 64 bytes between pieces)
    unsigned code = (blockIdx.x * 7 + wave * 13 + (lane & 15) * 5) % 65u;
    for (int pr = 0; pr < n_products; ++pr) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const double2_t* buf = lfrag + (pr & 1) * (FRAG_DOUBLES / 2);
        const int next_slot = (blockIdx.x + pr + 1) % N_SLOTS;
        double2_t row[G][8];
        const double2_t* rp[G];
#pragma unroll
        for (int t = 0; t < G; ++t) {
            code = (code * 29u + 11u + t) % 65u;
            rp[t] = reinterpret_cast<const double2_t*>(leaf_tab + (static_cast<size_t>((pr + t) % 58) * 65 + code) * 64) + g4;
        }
#pragma unroll
        for (int t = 0; t < G; ++t)
#pragma unroll
            for (int u = 0; u < 4; ++u) acc[t][u] = double4_t{0.0, 0.0, 0.0, 0.0};
        double2_t n01 = buf[lane], n23 = buf[64 + lane];
        auto step = [&](auto qc) {
            constexpr int q = decltype(qc)::value;
            const double2_t f01 = n01, f23 = n23;
            __builtin_amdgcn_sched_barrier(0);
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(f01.x, cur[0][q], acc[0][0], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (q + 1 < 16) {
                n01 = buf[((q + 1) * 2 + 0) * 64 + lane];
                n23 = buf[((q + 1) * 2 + 1) * 64 + lane];
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f01.y, cur[0][q], acc[0][1], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (q < 8) stage_piece(std::integral_constant<int, (q < 8 ? q : 0)>{}, next_slot, (pr + 1) & 1);
            __builtin_amdgcn_sched_barrier(0);
            acc[0][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(f23.x, cur[0][q], acc[0][2], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (q < 8) row[0][q] = rp[0][q * 4];
            __builtin_amdgcn_sched_barrier(0);
            acc[0][3] = __builtin_amdgcn_mfma_f64_16x16x4f64(f23.y, cur[0][q], acc[0][3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (G == 2) {
                acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(f01.x, cur[1][q], acc[1][0], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (q < 8) row[1][q] = rp[1][q * 4];
                __builtin_amdgcn_sched_barrier(0);
                acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f01.y, cur[1][q], acc[1][1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                acc[1][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(f23.x, cur[1][q], acc[1][2], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (q == 15) {   // tile 0's epilogue in the shadow of tile 1's last MFMAs
#pragma unroll
                    for (int u = 0; u < 16; ++u) cur[0][u] = acc[0][u >> 2][u & 3] * ((u & 1) ? row[0][u >> 1].y : row[0][u >> 1].x) * 0.015625;
                }
                __builtin_amdgcn_sched_barrier(0);
                acc[1][3] = __builtin_amdgcn_mfma_f64_16x16x4f64(f23.y, cur[1][q], acc[1][3], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
#define STEP(Q) step(std::integral_constant<int, Q>{});
        STEP(0) STEP(1) STEP(2) STEP(3) STEP(4) STEP(5) STEP(6) STEP(7) STEP(8) STEP(9) STEP(10) STEP(11) STEP(12) STEP(13) STEP(14) STEP(15)
#undef STEP
        // the epilogue: the product's result times the leaf row (the real kernel: leaf rows of one or two children, a clamp, ~20 address instructions)
        constexpr int LAST = G - 1;
#pragma unroll
        for (int u = 0; u < 16; ++u) cur[LAST][u] = acc[LAST][u >> 2][u & 3] * ((u & 1) ? row[LAST][u >> 1].y : row[LAST][u >> 1].x) * 0.015625;
    }
    double s = 0.0;
#pragma unroll
    for (int t = 0; t < G; ++t)
#pragma unroll
        for (int q = 0; q < 16; ++q) s += cur[t][q];
    if (s == 1.2345e300) out[0] = s;
}

template <int G>
static void run(const char* name, const double* d_frag, const double* d_leaf, double* d_out, int n_products)
{
    const size_t lds = G == 1 ? 2 * FRAG_DOUBLES * 8 : 100 * 1024;   // (B: more than half a CU's LDS, so that one work-group per CU is what runs)
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&stream_kernel<G>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)));
    const int blocks = G == 1 ? 512 * 8 : 256 * 8;   // eight generations of work-groups either way: the same number of tiles
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    hipLaunchKernelGGL(stream_kernel<G>, dim3(blocks), dim3(256), lds, 0, d_frag, d_leaf, 32, d_out);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(stream_kernel<G>, dim3(blocks), dim3(256), lds, 0, d_frag, d_leaf, n_products, d_out);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double mfma = static_cast<double>(blocks) * 4 * G * n_products * 64.0;
    const double tf = mfma * 2048.0 / (best * 1e-3) / 1e12;
    printf("%-64s %8.3f ms  %6.2f TFLOP/s = %.3f of the fp64 peak (78.6)\n", name, best, tf, tf / 78.6);
}

int main()
{
    std::vector<double> h(static_cast<size_t>(N_SLOTS) * FRAG_DOUBLES), hl(static_cast<size_t>(58) * 65 * 64);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 1e-2 * ((i * 2654435761u) % 97) / 97.0;
    for (size_t i = 0; i < hl.size(); ++i) hl[i] = 0.5 + ((i * 40503u) % 89) / 178.0;
    double *d_frag, *d_leaf, *d_out;
    CHECK(hipMalloc(&d_frag, h.size() * 8));
    CHECK(hipMalloc(&d_leaf, hl.size() * 8));
    CHECK(hipMalloc(&d_out, 8));
    CHECK(hipMemcpy(d_frag, h.data(), h.size() * 8, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(d_leaf, hl.data(), hl.size() * 8, hipMemcpyHostToDevice));
    const int n = 228;   // two passes over the 114 edges, like a work-group of prune_fixed
    run<1>("A: one tile per wave, two waves per SIMD (today's shape)", d_frag, d_leaf, d_out, n);
    run<2>("B: two tiles per wave, one wave per SIMD", d_frag, d_leaf, d_out, n);
    run<1>("A again", d_frag, d_leaf, d_out, n);
    run<2>("B again", d_frag, d_leaf, d_out, n);
    return 0;
}
