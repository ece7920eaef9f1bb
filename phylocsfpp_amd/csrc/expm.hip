// expm.hip -- batched 64x64 transition matrices on gfx950.
// P = S * diag(exp(lambda*t)) * S^-1 as one 64x64x64 fp64 MFMA product per job, followed by the
// reference's row post-processing (negative entries within 1e-6 clamped to 0, diagonal replaced by
// 1 - off-diagonal row sum, validation of entries and raw row sums; src/instance.hpp:602-640) and
// a scatter into the layout the pruning kernel reads (fragment order for internal children,
// leaf-message table incl. the row-sum row for leaves).  One work-group of 4 waves per job.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace pcsf {

typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void expm_kernel(const ExpmParams p)
{
    __shared__ double P[64 * 65];      // padded rows: the row pass reads along rows, the scatter along columns
    __shared__ double rowsum[64];
    __shared__ int bad;

    const ExpmJob job = p.jobs[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double* __restrict__ R = p.eig_right + static_cast<size_t>(job.eig) * 4096;
    const double* __restrict__ L = p.eig_left + static_cast<size_t>(job.eig) * 4096;
    const double* __restrict__ lam = p.eig_lambda + static_cast<size_t>(job.eig) * 64;
    if (tid == 0) bad = 0;

    // wave w computes rows 16w..16w+15 of P, all four 16-column tiles
    double4_t acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = double4_t{0.0, 0.0, 0.0, 0.0};
    const int i = 16 * wave + (lane & 15);
#pragma unroll 4
    for (int q = 0; q < 16; ++q) {
        const int k = 4 * q + (lane >> 4);
        const double a = R[i * 64 + k];
        const double e = exp(lam[k] * job.t);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const double b = L[k * 64 + 16 * c + (lane & 15)] * e;
            acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
        }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int r = 0; r < 4; ++r) P[(16 * wave + (lane >> 4) + 4 * r) * 65 + 16 * c + (lane & 15)] = acc[c][r];
    __syncthreads();

    // row post-processing, one thread per row, in index order like the reference
    if (tid < 64) {
        double* row = P + tid * 65;
        double total = 0.0, diag = 1.0;
        bool ok = true;
        for (int j = 0; j < 64; ++j) {
            const double v = row[j];
            total += v;
            if (v < 0.0) {
                if (fabs(v) > 1e-6) ok = false;
                row[j] = 0.0;
            }
            if (j != tid) diag -= row[j];
        }
        if (!(fabs(total - 1.0) <= 1e-6)) ok = false;
        row[tid] = diag;
        double s = 0.0;  // message of a gap leaf: sum of the final row, in index order (src/fixed_lik.hpp:111-117)
        for (int j = 0; j < 64; ++j) s += row[j];
        rowsum[tid] = s;
        if (!ok) bad = 1;
    }
    __syncthreads();
    if (tid == 0 && bad) p.fail[job.table] = 1u;

    if (job.slot >= 0) {
        // fragment order: frag[((q*2+h)*64 + l)*2 + e] = P[16*(2h+e) + (l&15)][4q + (l>>4)]
        double* dst = p.frag + (static_cast<size_t>(job.table) * p.n_edges + job.slot) * 4096;
        for (int idx = tid; idx < 4096; idx += 256) {
            const int e = idx & 1, l = (idx >> 1) & 63, h = (idx >> 7) & 1, q = idx >> 8;
            dst[idx] = P[(16 * (2 * h + e) + (l & 15)) * 65 + 4 * q + (l >> 4)];
        }
    }
    if (job.leaf >= 0) {
        // leaf table: tab[((code*8 + q2)*4 + g)*2 + e] = P[4(2 q2 + e) + g][code], code 64 = row sum
        double* dst = p.leaf_tab + (static_cast<size_t>(job.table) * p.n_leaves + job.leaf) * LEAF_TABLE_DOUBLES_DEV;
        for (int idx = tid; idx < LEAF_TABLE_DOUBLES_DEV; idx += 256) {
            const int e = idx & 1, g = (idx >> 1) & 3, q2 = (idx >> 3) & 7, code = idx >> 6;
            const int a = 4 * (2 * q2 + e) + g;
            dst[idx] = code < 64 ? P[a * 65 + code] : rowsum[a];
        }
    }
    if (p.raw_out) {
        double* dst = p.raw_out + static_cast<size_t>(blockIdx.x) * 4096;
        for (int idx = tid; idx < 4096; idx += 256) dst[idx] = P[(idx >> 6) * 65 + (idx & 63)];
    }
}

hipError_t launch_expm(const ExpmParams& p, hipStream_t stream)
{
    if (p.n_jobs == 0) return hipSuccess;
    hipLaunchKernelGGL(expm_kernel, dim3(p.n_jobs), dim3(256), 0, stream, p);
    return hipGetLastError();
}

}  // namespace pcsf
