// eigen.hip -- batched eigen-systems of the kappa/omega F3x4 codon rate matrix on gfx950.
//
// OMEGA re-derives the rate matrix and its eigen-system at every kappa evaluation of every block
// (reference src/omega.hpp:38-128 + src/instance.hpp:309-434: gsl_eigen_nonsymmv and a complex LU inverse;
// ~170-200 evaluations per block, half of them with a new matrix).  Here one work-group per job
//   1. builds pi and Q from the twelve q_settings exactly like hostmath.hpp::rate_matrix_omega
//      (same per-element arithmetic and the same summation order for the diagonal and the normaliser),
//   2. symmetrises B = D^1/2 Q D^-1/2 (Q is reversible w.r.t. pi),
//   3. diagonalises B with a PARALLEL-ORDER cyclic Jacobi iteration in LDS: 63 steps per sweep, each
//      applying 32 disjoint plane rotations (round-robin tournament pairing) as B <- J^T B J, V <- V J.
//      Only rotations, so the near-zero eigenvalue keeps its relative accuracy (the equilibrium prior is read
//      off its eigenvector), same stopping rule as the host solver (off^2 <= 1e-58 * diag^2),
//   4. writes right = D^-1/2 V, left = V^T D^1/2, lambda, and the root prior (+ ln(prior)*prior) of the
//      eigen-system in the lane order the pruning kernel reads.
// The host solver (hostmath.hpp::eigen_reversible, sequential row-cyclic order) stays the reference for tests:
// the two agree to rounding (tests/test_gpu_blocks.py), not bit for bit -- rotation order differs.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "kernels.h"

namespace pcsf {

typedef double double2_loc __attribute__((ext_vector_type(2)));

namespace {
constexpr int LD = 65;  // padded row length: column walks hit 32 different banks

// genetic code in A,C,G,T order, '*' = stop (as hostmath.hpp::amino_acid)
__device__ __forceinline__ char amino_acid_dev(int codon)
{
    const char* code = "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*Y*YSSSS*CWCLFLF";
    return code[codon];
}

__device__ __forceinline__ double f3x4_weight(const double* qs, int codon)
{
    const int d0 = codon >> 4, d1 = (codon >> 2) & 3, d2 = codon & 3;
    const double f1 = ((d0 == 3) ? 1.0 : qs[3 + d0]) / (1.0 + qs[3] + qs[4] + qs[5]);
    const double f2 = ((d1 == 3) ? 1.0 : qs[6 + d1]) / (1.0 + qs[6] + qs[7] + qs[8]);
    const double f3 = ((d2 == 3) ? 1.0 : qs[9 + d2]) / (1.0 + qs[9] + qs[10] + qs[11]);
    return f1 * f2 * f3;
}
}  // namespace

__device__ __forceinline__ unsigned long long eig_clock()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

__global__ __launch_bounds__(256) void eigen_omega_kernel(const EigenParams p)
{
    const bool stamping = p.stamps != nullptr && blockIdx.x == 0;
    const unsigned long long t_kernel = stamping ? eig_clock() : 0ull;
    unsigned long long acc_a = 0, acc_rot = 0, acc_v = 0, acc_b = 0, n_steps = 0, n_sweeps = 0;
    extern __shared__ __attribute__((aligned(16))) double lds_eig[];  // 2 x 64 x 65 doubles = 66,560 B
    double* A = lds_eig;
    double* V = lds_eig + 64 * LD;
    // rot[buf][pair] = (cos, sin), pq[buf][pair] = p | q << 8: the rotations of two consecutive steps.  Packed, because the
    // kernel is bound by the NUMBER of LDS instructions the CU's one LDS pipe can take (PMC: 136 per wave and step, 544 per
    // step; a (p, q, c, s) look-up per pair as four reads was a third of them), not by bytes or bank conflicts.
    __shared__ __attribute__((aligned(16))) double rot[2 * 32 * 2];
    __shared__ double pi[64], root[64], iroot[64], red[8];
    __shared__ int pq[64], pp[1];
    __shared__ double norm_s;
    __shared__ int flag;

    const EigenJob job = p.jobs[blockIdx.x];
    const double* qs = job.qs;
    const int tid = threadIdx.x;
    if (tid == 0) flag = 0;
    if (tid < 64) {
        const double stop_mass = f3x4_weight(qs, 48) + f3x4_weight(qs, 50) + f3x4_weight(qs, 56);  // TAA, TAG, TGA
        const double denom = 1.0 - ((1.0 - qs[2]) * stop_mass);
        pi[tid] = f3x4_weight(qs, tid) / denom;
    }
    __syncthreads();
    if (tid < 64) {
        if (!(pi[tid] > 0.0)) flag = 1;
        root[tid] = sqrt(pi[tid]);
        iroot[tid] = 1.0 / root[tid];
    }
    const double kappa = qs[0], omega = qs[1];
    for (int idx = tid; idx < 4096; idx += 256) {
        const int i = idx >> 6, j = idx & 63;
        double rate = 0.0;
        const int x = i ^ j;
        const int ndiff = ((x & 0x30) != 0) + ((x & 0x0c) != 0) + ((x & 0x03) != 0);
        if (ndiff == 1) {
            const int shift = (x & 0x30) ? 4 : ((x & 0x0c) ? 2 : 0);
            const int a = (i >> shift) & 3, b = (j >> shift) & 3;
            const bool transition = (a + b == 2) || (a + b == 4);
            rate = transition ? kappa : 1.0;
            const char ai = amino_acid_dev(i), aj = amino_acid_dev(j);
            rate *= (ai != '*' && aj != '*' && ai != aj) ? omega : 1.0;
            rate *= pi[j];
        }
        A[i * LD + j] = rate;
        V[i * LD + j] = i == j ? 1.0 : 0.0;
    }
    __syncthreads();
    if (flag) {  // pi has a non-positive entry: the model is invalid (host: eigen_reversible returns false)
        if (tid == 0) p.fail[job.table] = 1u;
        return;
    }
    if (tid < 64) {
        double diag = 0.0;
        for (int j = 0; j < 64; ++j)
            if (j != tid) diag -= A[tid * LD + j];
        A[tid * LD + tid] = diag;
    }
    __syncthreads();
    if (tid == 0) {
        double norm = 0.0;
        for (int i = 0; i < 64; ++i) norm -= pi[i] * A[i * LD + i];
        norm_s = norm;
    }
    __syncthreads();
    for (int idx = tid; idx < 4096; idx += 256) {
        const int i = idx >> 6, j = idx & 63;
        A[i * LD + j] = root[i] * (A[i * LD + j] / norm_s) * iroot[j];
    }
    __syncthreads();
    for (int idx = tid; idx < 4096; idx += 256) {
        const int i = idx >> 6, j = idx & 63;
        if (i < j) {
            const double m = 0.5 * (A[i * LD + j] + A[j * LD + i]);
            A[i * LD + j] = m;
            A[j * LD + i] = m;
        }
    }
    __syncthreads();

    const int lane = tid & 63, wave = tid >> 6;
    // Warm start: an optimiser moves kappa in small steps, so the eigenbasis V0 of an earlier matrix nearly diagonalises
    // this one.  B <- V0^T B V0 (two 64^3 products, each thread 16 entries), V <- V0, and the sweeps below finish the
    // job in 2-4 rounds instead of ~9.  V0 is exactly as orthogonal as the iteration left it, so the result is an
    // eigen-decomposition of B to the same accuracy as a cold start.
    if (job.warm >= 0 && p.eig_basis) {
        const double* __restrict__ V0 = p.eig_basis + static_cast<size_t>(job.warm) * 4096;
        for (int idx = tid; idx < 4096; idx += 256) V[(idx >> 6) * LD + (idx & 63)] = V0[idx];
        __syncthreads();
        double t[16];
        // T = B V0: thread owns row r = tid >> 2 and columns c = 16*(tid & 3) .. +15
        const int r = tid >> 2, c0 = 16 * (tid & 3);
#pragma unroll
        for (int u = 0; u < 16; ++u) t[u] = 0.0;
        for (int j = 0; j < 64; ++j) {
            const double b = A[r * LD + j];
#pragma unroll
            for (int u = 0; u < 16; ++u) t[u] += b * V[j * LD + c0 + u];
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 16; ++u) A[r * LD + c0 + u] = t[u];
        __syncthreads();
        // C = V0^T T: entry (r, c) = sum_i V0[i][r] T[i][c]
#pragma unroll
        for (int u = 0; u < 16; ++u) t[u] = 0.0;
        for (int i = 0; i < 64; ++i) {
            const double v = V[i * LD + r];
#pragma unroll
            for (int u = 0; u < 16; ++u) t[u] += v * A[i * LD + c0 + u];
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 16; ++u) A[r * LD + c0 + u] = t[u];
        __syncthreads();
        for (int idx = tid; idx < 4096; idx += 256) {  // exact symmetry again
            const int i = idx >> 6, j = idx & 63;
            if (i < j) {
                const double m = 0.5 * (A[i * LD + j] + A[j * LD + i]);
                A[i * LD + j] = m;
                A[j * LD + i] = m;
            }
        }
        __syncthreads();
    }
    // rotations of one step into buffer `buf`: tournament pairing of 0..63 (63 stays, the others rotate), lanes 0..31 of wave 0
    auto rotations = [&](int step, int buf) {
        if (tid < 32) {
            int a, b;
            if (tid == 0) {
                a = step;
                b = 63;
            } else {
                a = (step + tid) % 63;
                b = (step - tid + 63) % 63;
            }
            const int pi_ = a < b ? a : b, qi = a < b ? b : a;
            const double g = A[pi_ * LD + qi];
            double c = 1.0, s = 0.0;
            if (!(fabs(g) < 1e-290)) {
                const double zeta = (A[qi * LD + qi] - A[pi_ * LD + pi_]) / (2.0 * g);
                const double t = (fabs(zeta) > 1e150) ? 0.5 / zeta : copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                c = 1.0 / sqrt(1.0 + t * t);
                s = t * c;
            }
            reinterpret_cast<double2_loc*>(rot)[32 * buf + tid] = double2_loc{c, s};
            pq[32 * buf + tid] = pi_ | (qi << 8);
        }
    };
    for (int sweep = 0; sweep < 60; ++sweep) {
        // convergence: squared off-diagonal mass against squared diagonal mass
        double off = 0.0, total = 0.0;
        for (int idx = tid; idx < 4096; idx += 256) {
            const int i = idx >> 6, j = idx & 63;
            const double v = A[i * LD + j];
            if (i == j) total += v * v;
            else off += v * v;
        }
        for (int s = 32; s > 0; s >>= 1) {
            off += __shfl_xor(off, s);
            total += __shfl_xor(total, s);
        }
        if (lane == 0) {
            red[wave] = off;
            red[4 + wave] = total;
        }
        __syncthreads();
        off = red[0] + red[1] + red[2] + red[3];
        total = red[4] + red[5] + red[6] + red[7];
        __syncthreads();
        if (off <= 1e-58 * total) break;

        // One Jacobi step = (A) B <- J^T B J by all four waves, barrier, (B) V <- V J by waves 1-3 WHILE wave 0 computes the
        // rotations of the next step from the updated B, barrier.  The rotation parameters are a chain of ~150 dependent
        // fp64 instructions (two square roots, three divisions) on 32 lanes: as a phase of its own it was a third of a
        // step during which 224 of 256 threads waited; the arithmetic and its order per element are unchanged.
        rotations(0, 0);
        __syncthreads();
        n_sweeps += 1;
        for (int step = 0; step < 63; ++step) {
            const unsigned long long t0 = stamping ? eig_clock() : 0ull;
            const int cur = step & 1;
            const int* __restrict__ pqc = pq + 32 * cur;
            const double2_loc* __restrict__ rotc = reinterpret_cast<const double2_loc*>(rot) + 32 * cur;
            // (A) B <- J^T B J in ONE pass over 2x2 blocks: block (m, m') = rows {p,q} of pair m x columns {p',q'} of pair
            // m' is first rotated from the right by pair m', then from the left by pair m -- the same arithmetic as a
            // column pass followed by a row pass, with every entry read and written once.  All operands are read before
            // any result is written (the items are disjoint, but the compiler cannot know that for accesses to one LDS
            // array and would otherwise wait for an LDS round trip per item).
            {
                double b00[4], b01[4], b10[4], b11[4], cl[4], sl[4], cr[4], sr[4];
                int pl[4], ql[4], pr[4], qr[4];
                // a thread's four blocks share the column pair mr = tid & 31 and have row pairs (tid >> 5) + 8 i
                const int mr = tid & 31, pq_r = pqc[mr];
                const double2_loc rot_r = rotc[mr];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int m = (tid >> 5) + 8 * i, pq_l = pqc[m];
                    const double2_loc rot_l = rotc[m];
                    pl[i] = pq_l & 0xff;
                    ql[i] = pq_l >> 8;
                    cl[i] = rot_l.x;
                    sl[i] = rot_l.y;
                    pr[i] = pq_r & 0xff;
                    qr[i] = pq_r >> 8;
                    cr[i] = rot_r.x;
                    sr[i] = rot_r.y;
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    b00[i] = A[pl[i] * LD + pr[i]];
                    b01[i] = A[pl[i] * LD + qr[i]];
                    b10[i] = A[ql[i] * LD + pr[i]];
                    b11[i] = A[ql[i] * LD + qr[i]];
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int blk = tid + 256 * i;
                    // columns p', q' (right rotation) of both rows ...
                    const double t1 = cr[i] * b00[i] - sr[i] * b01[i], t2 = sr[i] * b00[i] + cr[i] * b01[i];
                    const double u1 = cr[i] * b10[i] - sr[i] * b11[i], u2 = sr[i] * b10[i] + cr[i] * b11[i];
                    // ... then rows p, q (left rotation)
                    double n00 = cl[i] * t1 - sl[i] * u1, n01 = cl[i] * t2 - sl[i] * u2;
                    double n10 = sl[i] * t1 + cl[i] * u1, n11 = sl[i] * t2 + cl[i] * u2;
                    if ((blk >> 5) == (blk & 31)) n01 = n10 = 0.0;  // the rotated pair itself: annihilated (or already below 1e-290)
                    A[pl[i] * LD + pr[i]] = n00;
                    A[pl[i] * LD + qr[i]] = n01;
                    A[ql[i] * LD + pr[i]] = n10;
                    A[ql[i] * LD + qr[i]] = n11;
                }
            }
            __syncthreads();
            const unsigned long long t1 = stamping ? eig_clock() : 0ull;
            // (B) wave 0: rotations of the next step; waves 1..3: V <- V J, row `lane`, pairs wave-1, wave+2, ...
            if (wave == 0) {
                if (step + 1 < 63) rotations(step + 1, cur ^ 1);
                if (stamping) acc_rot += eig_clock() - t1;
            } else {
                int pm[11], qm[11];
                double cm[11], sm[11], vx[11], vy[11];
#pragma unroll
                for (int i = 0; i < 11; ++i) {
                    const int m = (wave - 1) + 3 * i;
                    const int mm = m < 32 ? m : 31;
                    const int pq_m = pqc[mm];
                    const double2_loc rot_m = rotc[mm];
                    pm[i] = pq_m & 0xff;
                    qm[i] = pq_m >> 8;
                    cm[i] = rot_m.x;
                    sm[i] = rot_m.y;
                    vx[i] = V[lane * LD + pm[i]];
                    vy[i] = V[lane * LD + qm[i]];
                }
#pragma unroll
                for (int i = 0; i < 11; ++i) {
                    if ((wave - 1) + 3 * i < 32) {
                        V[lane * LD + pm[i]] = cm[i] * vx[i] - sm[i] * vy[i];
                        V[lane * LD + qm[i]] = sm[i] * vx[i] + cm[i] * vy[i];
                    }
                }
                if (stamping) acc_v += eig_clock() - t1;
            }
            __syncthreads();
            if (stamping) {
                const unsigned long long t2 = eig_clock();
                acc_a += t1 - t0;
                acc_b += t2 - t1;
                n_steps += 1;
            }
        }
    }
    if (stamping) {
        if (tid == 0) {
            atomicAdd(p.stamps + 0, acc_a);
            atomicAdd(p.stamps + 1, acc_rot);
            atomicAdd(p.stamps + 3, acc_b);
            atomicAdd(p.stamps + 4, n_steps);
            atomicAdd(p.stamps + 5, n_sweeps);
            atomicAdd(p.stamps + 6, eig_clock() - t_kernel);
        }
        if (tid == 64) atomicAdd(p.stamps + 2, acc_v);
    }

    // eigen-system of Q: right = D^-1/2 V (eigenvectors in columns), left = V^T D^1/2 (in rows)
    double* __restrict__ R = p.eig_right + static_cast<size_t>(job.eig) * 4096;
    double* __restrict__ L = p.eig_left + static_cast<size_t>(job.eig) * 4096;
    double* __restrict__ lam = p.eig_lambda + static_cast<size_t>(job.eig) * 64;
    double* __restrict__ basis = p.eig_basis ? p.eig_basis + static_cast<size_t>(job.eig) * 4096 : nullptr;
    for (int idx = tid; idx < 4096; idx += 256) {
        const int r = idx >> 6, c = idx & 63;
        R[idx] = iroot[r] * V[r * LD + c];   // right[i][k], i = r, k = c
        L[idx] = V[c * LD + r] * root[c];    // left[k][i],  k = r, i = c
        if (basis) basis[idx] = V[r * LD + c];
    }
    if (tid < 64) lam[tid] = A[tid * LD + tid];
    // equilibrium: left eigenvector of the eigenvalue closest to zero, scaled to sum 1 (index order)
    if (tid == 0) {
        int z = 0;
        for (int k = 1; k < 64; ++k)
            if (fabs(A[k * LD + k]) < fabs(A[z * LD + z])) z = k;
        double mass = 0.0;
        for (int j = 0; j < 64; ++j) mass += V[j * LD + z] * root[j];
        norm_s = mass;
        pp[0] = z;
    }
    __syncthreads();
    if (tid < 64) {
        const int z = pp[0];
        const double pr = (V[tid * LD + z] * root[tid]) / norm_s;
        const int lane_pos = (tid & 3) * 16 + (tid >> 2);  // lane order: [g][q] = state 4q+g
        // the root prior belongs to the eigen-system: row job.eig (the pruning kernel finds it through wg_prior)
        if (p.prior) p.prior[static_cast<size_t>(job.eig) * 64 + lane_pos] = pr;
        if (p.anc_weight) p.anc_weight[static_cast<size_t>(job.eig) * 64 + lane_pos] = log(pr) * pr;
        if (p.prior_plain) p.prior_plain[static_cast<size_t>(job.eig) * 64 + tid] = pr;
    }
}

hipError_t launch_eigen_omega(const EigenParams& p, hipStream_t stream)
{
    if (p.n_jobs == 0) return hipSuccess;
    constexpr size_t lds = 2 * 64 * LD * sizeof(double);
    static std::atomic<uint64_t> attr_set{0};  // per device (kernels.hip launch_impl)
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    if (!(attr_set.load(std::memory_order_acquire) >> (dev & 63) & 1)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&eigen_omega_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (e != hipSuccess) return e;
        attr_set.fetch_or(uint64_t(1) << (dev & 63), std::memory_order_release);
    }
    hipLaunchKernelGGL(eigen_omega_kernel, dim3(p.n_jobs), dim3(256), lds, stream, p);
    return hipGetLastError();
}

}  // namespace pcsf
