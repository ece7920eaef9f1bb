// expm.hip -- batched 64x64 transition matrices on gfx950.
// P = S * diag(exp(lambda*t)) * S^-1 as one 64x64x64 fp64 MFMA product per job, followed by the
// reference's row post-processing (negative entries within 1e-6 clamped to 0, diagonal replaced by
// 1 - off-diagonal row sum, validation of entries and raw row sums; src/instance.hpp:602-640) and
// a scatter into the layout the pruning kernel reads (fragment order for internal children,
// leaf-message table incl. the row-sum row for leaves).  One work-group of 4 waves per job.
//
// OFF THE PRODUCT PATH since round 3: no block of pcsf_score_blocks forms a branch matrix any more (prune_tiles / prune_factored apply
// R, diag(e^{lambda t}) and L factor by factor; FIXED tracks build their tables once per context on the host, api.cpp).  This kernel
// only serves the debug / test entry points that need the reference's ENTRY-WISE post-processing of P: pcsf_block_loglik_eig (the
// parity tests' layer 1: the oracle's eigen-system injected), pcsf_debug_branch_matrices, and PCSF_BLOCK_KERNEL=factored with
// PCSF_FACTORED_MAX_TILES (round 2's explicit path, kept for A/B).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "kernels.h"

namespace pcsf {

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

// jobs per work-group: consecutive jobs nearly always share an eigen-system (a block queues its n-1 branches
// back to back), so a work-group keeps S as MFMA operand fragments in registers (16 doubles per lane) and
// S^-1 in LDS (32 KiB) and only re-reads them when the eigen-system index changes.  One job per work-group re-read 64 KiB of
// eigenvectors per 0.5 MFLOP product and left the kernel at ~6 TFLOP/s (79 % of the OMEGA run).
constexpr int EXPM_CHUNK = 9;

__global__ __launch_bounds__(256, 2) void expm_kernel(const ExpmParams p)
{
    extern __shared__ __attribute__((aligned(16))) double lds_expm[];
    double* P = lds_expm;               // [64][65] padded rows: the row pass reads along rows, the scatter along columns
    double* Ls = lds_expm + 64 * 65;    // [64][64] S^-1 of the current eigen-system
    __shared__ double rowsum[64], lam_s[64], scale_s[64];
    __shared__ int bad;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t n_jobs = p.n_evals * p.n_branches;
    const uint32_t job0 = blockIdx.x * EXPM_CHUNK;
    const uint32_t job1 = min(n_jobs, job0 + EXPM_CHUNK);
    struct Job {
        uint32_t eig, table;
        int32_t slot, leaf;
        double t;
        uint64_t codes[2];
    };
    auto fetch = [&](uint32_t jx) {  // evaluation record + per-branch constants -> one job
        const uint32_t e = jx / p.n_branches, b = jx - e * p.n_branches;
        const ExpmEval ev = p.evals[e];
        Job j;
        j.eig = ev.eig;
        j.table = ev.table;
        j.slot = p.branch_slot[b];
        j.leaf = static_cast<int>(b) < p.n_leaves ? static_cast<int>(b) : -1;
        j.t = static_cast<double>(static_cast<float>(static_cast<double>(p.branch_len[b]) * ev.x));
        j.codes[0] = j.codes[1] = ~uint64_t(0);
        if (j.leaf >= 0 && p.leaf_codes) {
            const uint64_t* m = p.leaf_codes + (static_cast<size_t>(ev.block) * p.n_leaves + j.leaf) * 2;
            j.codes[0] = m[0];
            j.codes[1] = m[1];
        }
        return j;
    };
    const int i = 16 * wave + (lane & 15);  // wave w computes rows 16w..16w+15 of P, all four 16-column tiles
    double r_frag[16];
    uint32_t have_eig = 0xffffffffu;

    Job ahead = fetch(job0 < job1 ? job0 : 0);
    for (uint32_t jx = job0; jx < job1; ++jx) {
        const Job job = ahead;
        ahead = fetch(jx + 1 < job1 ? jx + 1 : jx);  // the next job's parameters travel while this one computes
        if (job.eig != have_eig) {
            const double* __restrict__ R = p.eig_right + static_cast<size_t>(job.eig) * 4096;
            const double* __restrict__ L = p.eig_left + static_cast<size_t>(job.eig) * 4096;
            const double* __restrict__ lam = p.eig_lambda + static_cast<size_t>(job.eig) * 64;
#pragma unroll
            for (int q = 0; q < 16; ++q) r_frag[q] = R[i * 64 + 4 * q + (lane >> 4)];
            have_eig = job.eig;
            __syncthreads();  // lam_s / Ls may still be read by the previous job
            if (tid < 64) lam_s[tid] = lam[tid];
            for (int idx = tid; idx < 2048; idx += 256)
                reinterpret_cast<double2_t*>(Ls)[idx] = reinterpret_cast<const double2_t*>(L)[idx];
            __syncthreads();
        }
        if (tid == 0) bad = 0;
        if (wave == static_cast<int>((jx + 2) & 3)) scale_s[lane] = exp(lam_s[lane] * job.t);  // one exponential per eigenvalue and job
        __syncthreads();

        double4_t acc[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = double4_t{0.0, 0.0, 0.0, 0.0};
        // operands of k-step q+1 are requested from LDS before the MFMAs of k-step q are issued
        const double* lrow = Ls + (lane >> 4) * 64 + (lane & 15);
        double ln[4], en;
#pragma unroll
        for (int c = 0; c < 4; ++c) ln[c] = lrow[16 * c];
        en = scale_s[lane >> 4];
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            double lc[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) lc[c] = ln[c];
            const double e = en;
            __builtin_amdgcn_sched_barrier(0);
            if (q + 1 < 16) {
#pragma unroll
                for (int c = 0; c < 4; ++c) ln[c] = lrow[(q + 1) * 256 + 16 * c];
                en = scale_s[4 * (q + 1) + (lane >> 4)];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(r_frag[q], lc[c] * e, acc[c], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) P[(16 * wave + (lane >> 4) + 4 * r) * 65 + 16 * c + (lane & 15)] = acc[c][r];
        __syncthreads();

        // Row post-processing, one thread per row, sums in index order like the reference; three sweeps over the row
        // so that no sweep needs a per-entry select: (1) raw row sum, most negative entry, clamp negatives to zero;
        // (2) with a zero parked on the diagonal, diag = 1 - (sum of the row); (3) the final row's sum (the message of
        // a gap leaf, src/fixed_lik.hpp:111-117).  Vector-ALU instructions are time taken from the matrix pipe
        // (DESIGN.md 3.1), so the wave that does this -- and the one that computes the exponentials -- changes with
        // every job: with one work-group per SIMD-pair of waves, a fixed wave would load one SIMD only.
        if (wave == static_cast<int>(jx & 3)) {
            double* row = P + lane * 65;
            double total = 0.0, most_negative = 0.0;
#pragma unroll 1
            for (int j0 = 0; j0 < 64; j0 += 8) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = row[j0 + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    total += v[u];
                    most_negative = fmin(most_negative, v[u]);
                    row[j0 + u] = fmax(v[u], 0.0);
                }
            }
            const bool ok = (fabs(total - 1.0) <= 1e-6) && !(most_negative < -1e-6);
            row[lane] = 0.0;
            double diag = 1.0;
#pragma unroll 1
            for (int j0 = 0; j0 < 64; j0 += 8) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = row[j0 + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) diag -= v[u];
            }
            row[lane] = diag;
            double s = 0.0;
#pragma unroll 1
            for (int j0 = 0; j0 < 64; j0 += 8) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = row[j0 + u];
#pragma unroll
                for (int u = 0; u < 8; ++u) s += v[u];
            }
            rowsum[lane] = s;
            if (!ok) bad = 1;
        }
        __syncthreads();
        if (tid == 0 && bad) p.fail[job.table] = 1u;

        if (job.slot >= 0) {
            // fragment order: frag[((q*2+h)*64 + l)*2 + e] = P[16*(2h+e) + (l&15)][4q + (l>>4)]; the two e of a lane
            // leave as one 16-byte store
            double2_t* dst = reinterpret_cast<double2_t*>(p.frag + (static_cast<size_t>(job.table) * p.n_edges + job.slot) * 4096);
            for (int idx = tid; idx < 2048; idx += 256) {
                const int l = idx & 63, h = (idx >> 6) & 1, q = idx >> 7;
                const double* src = P + (32 * h + (l & 15)) * 65 + 4 * q + (l >> 4);
                dst[idx] = double2_t{src[0], src[16 * 65]};
            }
        }
        if (job.leaf >= 0) {
            // leaf table: tab[((code*8 + q2)*4 + g)*2 + e] = P[4(2 q2 + e) + g][code], code 64 = row sum
            double2_t* dst = reinterpret_cast<double2_t*>(p.leaf_tab + (static_cast<size_t>(job.table) * p.n_leaves + job.leaf) * LEAF_TABLE_DOUBLES_DEV);
            for (int idx = tid; idx < LEAF_TABLE_DOUBLES_DEV / 2; idx += 256) {
                const int g = idx & 3, q2 = (idx >> 2) & 7, code = idx >> 5;
                if (!((job.codes[code >> 6] >> (code & 63)) & 1)) continue;  // row not read by this block
                const int a = 8 * q2 + g;  // e = 0; e = 1 is state a + 4
                dst[idx] = code < 64 ? double2_t{P[a * 65 + code], P[(a + 4) * 65 + code]} : double2_t{rowsum[a], rowsum[a + 4]};
            }
        }
        if (p.raw_out) {
            double* dst = p.raw_out + static_cast<size_t>(jx) * 4096;
            for (int idx = tid; idx < 4096; idx += 256) dst[idx] = P[(idx >> 6) * 65 + (idx & 63)];
        }
        __syncthreads();  // P is rewritten by the next job
    }
}

hipError_t launch_expm(const ExpmParams& p, hipStream_t stream)
{
    const uint64_t n_jobs = static_cast<uint64_t>(p.n_evals) * p.n_branches;
    if (n_jobs == 0) return hipSuccess;
    if (n_jobs > 0x7fffffffULL) return hipErrorInvalidValue;
    constexpr size_t lds = (64 * 65 + 64 * 64) * sizeof(double);
    static std::atomic<uint64_t> attr_set{0};  // per device (kernels.hip launch_impl)
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    if (!(attr_set.load(std::memory_order_acquire) >> (dev & 63) & 1)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&expm_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (e != hipSuccess) return e;
        attr_set.fetch_or(uint64_t(1) << (dev & 63), std::memory_order_release);
    }
    hipLaunchKernelGGL(expm_kernel, dim3(static_cast<unsigned>((n_jobs + EXPM_CHUNK - 1) / EXPM_CHUNK)), dim3(256), lds, stream, p);
    return hipGetLastError();
}

}  // namespace pcsf
