// Micro-benchmark of prune_factored's inner product (factored.hip `matvec`): a 64x64 matrix whose rows a work-group's four waves keep
// as MFMA A operands, applied to a 64x16 tile that is all-gathered through LDS before every product.  Variants isolate the cost of
// the barrier, of the LDS exchange and of the accumulator chains.   hipcc -O3 --offload-arch=gfx950 -o tools/_build/matvec_bench tools/matvec_bench.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef double double2_t __attribute__((ext_vector_type(2)));
typedef double double4_t __attribute__((ext_vector_type(4)));

// VAR 0: as in the kernel (two accumulator chains); 1: no barrier (wrong, timing only); 2: no LDS at all (operands from registers);
// 3: four accumulator chains; 4: one chain
template <int VAR>
__global__ __launch_bounds__(256, 2) void bench(const double* M, double* out, int iters)
{
    __shared__ __attribute__((aligned(16))) double2_t xbuf[2 * 4 * 2 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, sj = lane & 15, g = lane >> 4;
    double MA[16];
    for (int q = 0; q < 16; ++q) MA[q] = M[(16 * wave + sj) * 64 + 4 * q + g];
    double v[4] = {1.0 + lane * 1e-3, 0.5, 0.25 + wave, 0.125};
    int nprod = 0;
    for (int it = 0; it < iters; ++it) {
        double2_t b[8];
        if (VAR != 2) {
            double2_t* xb = xbuf + (nprod & 1) * (4 * 2 * 64);
            ++nprod;
            xb[(wave * 2 + 0) * 64 + lane] = double2_t{v[0], v[1]};
            xb[(wave * 2 + 1) * 64 + lane] = double2_t{v[2], v[3]};
            if (VAR != 1) __syncthreads();
#pragma unroll
            for (int i = 0; i < 8; ++i) b[i] = xb[i * 64 + lane];
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) b[i] = double2_t{v[i & 3], v[(i + 1) & 3]};
        }
        double4_t a0 = {0, 0, 0, 0}, a1 = {0, 0, 0, 0}, a2 = {0, 0, 0, 0}, a3 = {0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (VAR == 3) {
                if (i & 1) {
                    a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(MA[2 * i], b[i].x, a2, 0, 0, 0);
                    a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(MA[2 * i + 1], b[i].y, a3, 0, 0, 0);
                } else {
                    a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(MA[2 * i], b[i].x, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(MA[2 * i + 1], b[i].y, a1, 0, 0, 0);
                }
            } else if (VAR == 4) {
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(MA[2 * i], b[i].x, a0, 0, 0, 0);
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(MA[2 * i + 1], b[i].y, a0, 0, 0, 0);
            } else {
                a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(MA[2 * i], b[i].x, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(MA[2 * i + 1], b[i].y, a1, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = ((a0[r] + a1[r]) + (a2[r] + a3[r])) * 1e-2;
    }
    out[blockIdx.x * 256 + tid] = v[0] + v[1] + v[2] + v[3];
}

template <int VAR>
void run(const char* what, const double* dM, double* dout, int iters)
{
    for (int grid : {256, 512, 768, 1536, 3072}) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipLaunchKernelGGL(bench<VAR>, dim3(grid), dim3(256), 0, 0, dM, dout, iters);
        hipEventRecord(e0);
        hipLaunchKernelGGL(bench<VAR>, dim3(grid), dim3(256), 0, 0, dM, dout, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        // cycles at 2.4 GHz per product, per work-group round (grid / 256 work-groups per CU in sequence or side by side)
        printf("%-44s grid %5d: %8.3f ms = %7.0f cycles per product and work-group, %6.0f per product and CU\n", what, grid, ms, ms * 1e-3 * 2.4e9 / iters,
               ms * 1e-3 * 2.4e9 / iters / (grid / 256.0));
    }
}

int main()
{
    std::vector<double> M(4096);
    for (int i = 0; i < 4096; ++i) M[i] = ((i * 2654435761u) % 1000) * 1e-3;
    double *dM, *dout;
    hipMalloc(&dM, 4096 * 8);
    hipMalloc(&dout, 3072 * 256 * 8);
    hipMemcpy(dM, M.data(), 4096 * 8, hipMemcpyHostToDevice);
    const int iters = 2000;
    run<0>("as in prune_factored (2 chains)", dM, dout, iters);
    run<1>("no barrier", dM, dout, iters);
    run<2>("no LDS exchange", dM, dout, iters);
    run<3>("4 accumulator chains", dM, dout, iters);
    run<4>("1 accumulator chain", dM, dout, iters);
    return 0;
}
