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
//      off its eigenvector), stopping rule off^2 <= 1e-58 * diag^2 like the host solver (register-resident form: 1e-36, see there),
//   4. writes right = D^-1/2 V, left = V^T D^1/2, lambda, and the root prior (+ ln(prior)*prior) of the
//      eigen-system in the lane order the pruning kernel reads.
// The host solver (hostmath.hpp::eigen_reversible, sequential row-cyclic order) stays the reference for tests:
// the two agree to rounding (tests/test_gpu_blocks.py), not bit for bit -- rotation order differs.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <cstdlib>
#include <type_traits>

#include "kernels.h"

#ifndef PCSF_EIG_ABL
#define PCSF_EIG_ABL 0
#endif
// per-phase stamps inside a step of the register-resident form: a build flag (-DPCSF_EIGEN_STEP_STAMPS=1, tools/build_variant.sh), because
// even a never-taken stamp branch splits the step into basic blocks and stops the compiler from overlapping its phases
#ifndef PCSF_EIGEN_STEP_STAMPS
#define PCSF_EIGEN_STEP_STAMPS 0
#endif
#if PCSF_EIGEN_STEP_STAMPS
#define EIG_STEP_STAMPS true
#define EIG_STEP_CLOCK() (stamping ? eig_clock() : 0ull)
#else
#define EIG_STEP_STAMPS false
#define EIG_STEP_CLOCK() 0ull
#endif

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

// ---- register-resident variant (REG): the pair schedule and its bookkeeping -----------------------------------------------------
// The 64 indices form 16 groups of four; a ROUND puts two groups into each of eight super-groups (round-robin tournament of the
// groups, group 15 fixed), and wave 0's thread (R, C) keeps the 8x8 block rows(super-group R) x columns(super-group C) of B in 64
// registers, wave 1's thread the block rows 8R..8R+7 x columns(super-group C) of V.  Within a round every rotation pair lies inside
// one super-group, so B <- J^T B J and V <- V J touch registers only: four "cross" steps pair slot i of the first group with slot
// (i + t) & 3 of the second; round 0 adds three steps for the pairs inside the groups.  15 rounds = 63 steps = all 2,016 pairs once
// (tests/test_schedule.py checks the cover).  Between rounds the blocks pass through LDS once (lane-major, conflict-free writes).
namespace rr {
__host__ __device__ __forceinline__ constexpr int pair_a(int pat, int k)
{
    // patterns 0..2: inside the groups {0..3} and {4..7}; 3..6: cross, t = pat - 3
    return pat == 0 ? 2 * k : pat == 1 ? (k < 2 ? k : k + 2) : pat == 2 ? (k == 0 ? 0 : k == 1 ? 1 : k == 2 ? 4 : 5) : k;
}
__host__ __device__ __forceinline__ constexpr int pair_b(int pat, int k)
{
    return pat == 0 ? 2 * k + 1 : pat == 1 ? (k < 2 ? k + 2 : k + 4) : pat == 2 ? (k == 0 ? 3 : k == 1 ? 2 : k == 2 ? 7 : 6) : 4 + ((k + (pat - 3)) & 3);
}
// Is the 2x2 block (row pair m) x (column pair n) of step pattern `pat` one the NEXT step's rotations are read from?  Those are B[a][a],
// B[b][b], B[a][b] of every pair (a, b) of pattern `next`: the four diagonal 2x2 blocks and the four holding the next pairs' (a, b).
__host__ __device__ __forceinline__ constexpr bool feeds_next(int pat, int next, int m, int n)
{
    const int r0 = pair_a(pat, m), r1 = pair_b(pat, m), c0 = pair_a(pat, n), c1 = pair_b(pat, n);
    for (int k = 0; k < 4; ++k) {
        const int a = pair_a(next, k), b = pair_b(next, k);
        const bool row_a = a == r0 || a == r1, row_b = b == r0 || b == r1, col_a = a == c0 || a == c1, col_b = b == c0 || b == c1;
        if ((row_a && col_a) || (row_b && col_b) || (row_a && col_b)) return true;
    }
    return false;
}
// the two groups of super-group G in round r
__host__ __device__ __forceinline__ int group_of(int r, int G, int half)
{
    if (G == 0) return half ? 15 : r;
    const int g = half ? r + 15 - G : r + G;
    return g >= 15 ? g - 15 : g;
}
// where group g sits in round r: super-group | half << 3
__host__ __device__ __forceinline__ int locate(int r, int g)
{
    if (g == 15) return 0 | (1 << 3);
    int d = g - r;
    if (d < 0) d += 15;
    if (d == 0) return 0;
    return d <= 7 ? d : ((15 - d) | (1 << 3));
}
// Regrouping between two rounds is the SAME permutation every time (every group moves one seat round the table): the half
// `half` of super-group G takes over from super-group | half << 3
__host__ __device__ __forceinline__ constexpr int shift_source(int G, int half)
{
    return half == 0 ? (G <= 6 ? (G + 1) : (7 | (1 << 3))) : (G == 0 ? (0 | (1 << 3)) : G == 1 ? 0 : ((G - 1) | (1 << 3)));
}
}  // namespace rr

// REG: 128 threads (wave 0: B, wave 1: V; two work-groups fit a CU); otherwise 256 threads with B and V in LDS
template <bool REG>
__global__ __launch_bounds__(REG ? 128 : 256) void eigen_omega_kernel(const EigenParams p)
{
    constexpr int NT = REG ? 128 : 256;     // threads
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
    __shared__ char aa_s[64];   // the genetic code, staged once: a look-up in the string literal is a dependent global load per matrix entry
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
        aa_s[tid] = amino_acid_dev(tid);
    }
    __syncthreads();
    if (tid < 64) {
        if (!(pi[tid] > 0.0)) flag = 1;
        root[tid] = sqrt(pi[tid]);
        iroot[tid] = 1.0 / root[tid];
    }
    const double kappa = qs[0], omega = qs[1];
    for (int idx = tid; idx < 4096; idx += NT) {
        const int i = idx >> 6, j = idx & 63;
        double rate = 0.0;
        const int x = i ^ j;
        const int ndiff = ((x & 0x30) != 0) + ((x & 0x0c) != 0) + ((x & 0x03) != 0);
        if (ndiff == 1) {
            const int shift = (x & 0x30) ? 4 : ((x & 0x0c) ? 2 : 0);
            const int a = (i >> shift) & 3, b = (j >> shift) & 3;
            const bool transition = (a + b == 2) || (a + b == 4);
            rate = transition ? kappa : 1.0;
            const char ai = aa_s[i], aj = aa_s[j];
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
    for (int idx = tid; idx < 4096; idx += NT) {
        const int i = idx >> 6, j = idx & 63;
        A[i * LD + j] = root[i] * (A[i * LD + j] / norm_s) * iroot[j];
    }
    __syncthreads();
    for (int idx = tid; idx < 4096; idx += NT) {
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
        for (int idx = tid; idx < 4096; idx += NT) V[(idx >> 6) * LD + (idx & 63)] = V0[idx];
        __syncthreads();
        // Both 64^3 products on the matrix cores (v_mfma_f64_16x16x4_f64: A operand lane (s, g) = A[s][g], B operand = B[g][s], result
        // register r = row 4r + g, column s): wave w owns the output tiles of rows 16 TI w .. 16 TI (w + 1) - 1, 16 k-steps per tile, the
        // operands read from the two LDS matrices (row length 65: two lanes per bank at worst).  As scalar loops -- 32 entries and 4,096
        // LDS reads per thread and product -- they were 50 us of a 0.31 ms warm solve (profiles/r03/eigen_kernel.txt, variant R0).
        typedef double double4_loc __attribute__((ext_vector_type(4)));
        constexpr int TI = 4 / (NT / 64);   // row tiles per wave
        const int ms = lane & 15, mg = lane >> 4;
        double4_loc acc[TI][4];
        // T = B V0
#pragma unroll
        for (int a = 0; a < TI; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = double4_loc{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
        for (int kk = 0; kk < 16; ++kk) {
            double fa[TI], fb[4];
#pragma unroll
            for (int a = 0; a < TI; ++a) fa[a] = A[(16 * (wave * TI + a) + ms) * LD + 4 * kk + mg];
#pragma unroll
            for (int b = 0; b < 4; ++b) fb[b] = V[(4 * kk + mg) * LD + 16 * b + ms];
#pragma unroll
            for (int a = 0; a < TI; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], fb[b], acc[a][b], 0, 0, 0);
        }
        __syncthreads();   // every wave has read its rows of B
#pragma unroll
        for (int a = 0; a < TI; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) A[(16 * (wave * TI + a) + 4 * r + mg) * LD + 16 * b + ms] = acc[a][b][r];
        __syncthreads();
        // C = V0^T T: A operand V0^T[i][k] = V0[k][i]
#pragma unroll
        for (int a = 0; a < TI; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = double4_loc{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
        for (int kk = 0; kk < 16; ++kk) {
            double fa[TI], fb[4];
#pragma unroll
            for (int a = 0; a < TI; ++a) fa[a] = V[(4 * kk + mg) * LD + 16 * (wave * TI + a) + ms];
#pragma unroll
            for (int b = 0; b < 4; ++b) fb[b] = A[(4 * kk + mg) * LD + 16 * b + ms];
#pragma unroll
            for (int a = 0; a < TI; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[a], fb[b], acc[a][b], 0, 0, 0);
        }
        __syncthreads();   // every wave has read T
#pragma unroll
        for (int a = 0; a < TI; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b)
#pragma unroll
                for (int r = 0; r < 4; ++r) A[(16 * (wave * TI + a) + 4 * r + mg) * LD + 16 * b + ms] = acc[a][b][r];
        __syncthreads();
        for (int idx = tid; idx < 4096; idx += NT) {  // exact symmetry again
            const int i = idx >> 6, j = idx & 63;
            if (i < j) {
                const double m = 0.5 * (A[i * LD + j] + A[j * LD + i]);
                A[i * LD + j] = m;
                A[j * LD + i] = m;
            }
        }
        __syncthreads();
    }
    if constexpr (!REG) {
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
    bool lds_converged = false;
    for (int sweep = 0; sweep < 60; ++sweep) {
        // convergence: squared off-diagonal mass against squared diagonal mass
        double off = 0.0, total = 0.0;
        for (int idx = tid; idx < 4096; idx += NT) {
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
        if (off <= 1e-58 * total) {
            lds_converged = true;
            break;
        }

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
    if (!lds_converged && tid == 0) p.fail[job.table] = 1u;   // sweeps used up (a NaN makes the stop test false for ever): flagged, see below
    } else {
    // ================= register-resident sweeps: wave 0 owns B, wave 1 owns V ==================
    double2_loc* rt = reinterpret_cast<double2_loc*>(rot);   // [buf][super-group * 4 + pair] = (cos, sin)
    __shared__ __attribute__((aligned(16))) double dg[32 * 2];  // [super-group * 4 + pair] = (B[b][b] - B[a][a], 2 B[a][b]) of the next step's pairs
    const int R = lane >> 3, C = lane & 7;
    // One rotation from (B[a][a], B[b][b], B[a][b]).  The same t = sgn(zeta) / (|zeta| + sqrt(1 + zeta^2)), zeta = (B[b][b] - B[a][a]) / (2 B[a][b]),
    // as the `rotations` lambda above, written as t = 2g sgn(d) / (|d| + sqrt(d^2 + 4 g^2)) on operands scaled by a power of two (no
    // overflow / underflow branch) and turned into cos / sin with two reciprocal square roots (hardware seed + three Newton steps)
    // instead of three IEEE divisions and two square roots: this chain sits between two B updates, nothing can hide it.
    // c^2 + s^2 = (w^2 + h^2) / (2 r w) = 1 to an ulp.
    auto rsqrt_nr = [](double x) {
        double y = __builtin_amdgcn_rsq(x);
        const double hx = 0.5 * x;
#pragma unroll
        for (int it = 0; it < 3; ++it) y = y * __builtin_fma(-hx, y * y, 1.5);
        return y;
    };
    auto rotation_of = [&](double d, double h) {   // d = B[b][b] - B[a][a], h = 2 B[a][b]
        const bool tiny = fabs(h) < 2.0 * 1e-290;   // |B[a][b]| < 1e-290: no rotation (identity); computed without a branch, selected below
        const int eh = __builtin_amdgcn_frexp_exp(h), ed = __builtin_amdgcn_frexp_exp(d);
        const int e = (d == 0.0 || eh > ed) ? eh : ed;   // frexp_exp(0) = 0 must not win; h != 0 unless tiny
        d = __builtin_amdgcn_ldexp(d, -e);
        h = __builtin_amdgcn_ldexp(h, -e);
        // r = sqrt(d^2 + h^2), w = r + |d|:  cos = w / sqrt(2 r w), sin = sgn(d) h / sqrt(2 r w)   (tan = sgn(d) h / w as above)
        const double x = __builtin_fma(d, d, h * h);            // in [1/4, 2)
        const double w = __builtin_fma(x, rsqrt_nr(x), fabs(d));
        const double q = rsqrt_nr(2.0 * (w - fabs(d)) * w);
        const double c = w * q;
        const double sn = copysign(h, d * h) * q;   // sgn(d) h; d = +-0 rotates by 45 degrees like sgn(zeta) = +1
        return double2_loc{tiny ? 1.0 : c, tiny ? 0.0 : sn};
    };
    if (wave == 0) {
        double b[8][8];
        double* stage = A;   // 64 elements x 64 lanes once the block is in registers
        {   // round 0 blocks from the canonical matrix
            int rb[2], cb[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                rb[h] = 4 * rr::group_of(0, R, h);
                cb[h] = 4 * rr::group_of(0, C, h);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) b[i][j] = A[(rb[i >> 2] + (i & 3)) * LD + cb[j >> 2] + (j & 3)];
        }
        const bool diag = R == C;
        int base[2][2];   // where the four 4x4 quarters of the next round's block sit in the staged blocks of this round (16-byte items)
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y) {
                const int lr = rr::shift_source(R, x), lc = rr::shift_source(C, y);
                base[x][y] = 1024 * (lr >> 3) + 128 * (lc >> 3) + 8 * (lr & 7) + (lc & 7);
            }
        auto regroup = [&]() {   // element pairs (i, 2jp), (i, 2jp+1) as one 16-byte item, lane-major: no bank conflicts
            double2_loc* st2 = reinterpret_cast<double2_loc*>(stage);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int jp = 0; jp < 4; ++jp) st2[(i * 4 + jp) * 64 + lane] = double2_loc{b[i][2 * jp], b[i][2 * jp + 1]};
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int jp = 0; jp < 4; ++jp) {
                    const double2_loc e = st2[base[i >> 2][jp >> 1] + 256 * (i & 3) + 64 * (jp & 1)];
                    b[i][2 * jp] = e.x;
                    b[i][2 * jp + 1] = e.y;
                }
        };
        // parameters of the step with pattern PAT into buffer `buf`: diagonal blocks publish (B[b][b] - B[a][a], 2 B[a][b]) of their pairs
        // (all rotation_of reads of them), every lane rotates one pair (lanes 32..63 repeat 0..31: no branch between the LDS round trip
        // and the arithmetic, so the compiler is free to run other work of the step inside their latency)
        double hmax = 0.0;   // largest |2 B[a][b]| among the rotations published since the last stop test
        auto publish_operands = [&](auto pat_tag) {
            constexpr int PAT = decltype(pat_tag)::value;
            if (diag) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int a = rr::pair_a(PAT, k), bq = rr::pair_b(PAT, k);
                    *reinterpret_cast<double2_loc*>(&dg[(R * 4 + k) * 2]) = double2_loc{b[bq][bq] - b[a][a], 2.0 * b[a][bq]};
                }
            }
        };
        auto publish_rotations = [&](int buf) {
            const double2_loc dh = *reinterpret_cast<const double2_loc*>(&dg[(lane & 31) * 2]);
            hmax = fmax(hmax, fabs(dh.y));
#if PCSF_EIG_ABL & 2
            const double2_loc rot_new = double2_loc{0.8 + 1e-9 * dh.x, 0.6 + 1e-9 * dh.y};
#else
            const double2_loc rot_new = rotation_of(dh.x, dh.y);
#endif
            return rot_new;
        };
        auto store_rotations = [&](double2_loc rot_new, int buf) {
            rt[32 * buf + (lane & 31)] = rot_new;   // lanes 32..63 store what lanes 0..31 do (a conditional store would pull the arithmetic into its branch)
        };
        // B <- J^T B J of one step.  PART 0: the 2x2 blocks the next step's rotations are read from (rr::feeds_next), PART 1: the others,
        // PART 2: all of them.
        double cl[4], sl[4], cr[4], sr[4];
        auto load_rotations = [&](int buf) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double2_loc l = rt[32 * buf + R * 4 + k], r2 = rt[32 * buf + C * 4 + k];
                cl[k] = l.x;
                sl[k] = l.y;
                cr[k] = r2.x;
                sr[k] = r2.y;
            }
        };
        auto b_blocks = [&](auto pat_tag, auto next_tag, auto part_tag) {
            constexpr int PAT = decltype(pat_tag)::value, NEXT = decltype(next_tag)::value, PART = decltype(part_tag)::value;
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) {
                    if (PART != 2 && rr::feeds_next(PAT, NEXT, m, n) != (PART == 0)) continue;
                    const int a = rr::pair_a(PAT, m), bq = rr::pair_b(PAT, m), a2 = rr::pair_a(PAT, n), b2 = rr::pair_b(PAT, n);
                    const double b00 = b[a][a2], b01 = b[a][b2], b10 = b[bq][a2], b11 = b[bq][b2];
                    // columns (right rotation of pair n), then rows (left rotation of pair m)
                    const double t1 = __builtin_fma(cr[n], b00, -(sr[n] * b01)), t2 = __builtin_fma(sr[n], b00, cr[n] * b01);
                    const double u1 = __builtin_fma(cr[n], b10, -(sr[n] * b11)), u2 = __builtin_fma(sr[n], b10, cr[n] * b11);
                    b[a][a2] = __builtin_fma(cl[m], t1, -(sl[m] * u1));
                    b[a][b2] = __builtin_fma(cl[m], t2, -(sl[m] * u2));
                    b[bq][a2] = __builtin_fma(sl[m], t1, cl[m] * u1);
                    b[bq][b2] = __builtin_fma(sl[m], t2, cl[m] * u2);
                }
        };
        auto annihilate = [&](auto pat_tag) {   // the rotated pairs themselves: annihilated (or already below 1e-290); they sit in the diagonal 2x2 blocks
            constexpr int PAT = decltype(pat_tag)::value;
            if (diag) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int a = rr::pair_a(PAT, k), bq = rr::pair_b(PAT, k);
                    b[a][bq] = 0.0;
                    b[bq][a] = 0.0;
                }
            }
        };
        publish_operands(std::integral_constant<int, 0>{});
        store_rotations(publish_rotations(0), 0);
        load_rotations(0);   // a wave's LDS operations complete in order: it reads back what it has just stored
        int buf = 0;
        // diagnostics (PCSF_EIGEN_STAMPS): cycles in the B update, in publishing the next rotations, in the regrouping, at the barrier
        auto timed_step = [&](auto pat, auto next_pat, int regroup_from, int regroup_to) {
            const unsigned long long t0 = EIG_STEP_CLOCK();
            using std::integral_constant;
            unsigned long long t1 = 0, t2 = 0;
            if (regroup_from >= 0) {
#if !(PCSF_EIG_ABL & 8)
                b_blocks(pat, next_pat, integral_constant<int, 2>{});
                annihilate(pat);
#endif
                t1 = EIG_STEP_CLOCK();
#if !(PCSF_EIG_ABL & 1)
                regroup();
#endif
                t2 = EIG_STEP_CLOCK();
                publish_operands(next_pat);
                store_rotations(publish_rotations(buf ^ 1), buf ^ 1);
            } else {
                // the blocks the next rotations depend on first; the rest of the update runs inside the latency of the operands' trip
                // through LDS and of the rotation arithmetic (one dependent chain of ~35 fp64 instructions)
#if !(PCSF_EIG_ABL & 8)
                b_blocks(pat, next_pat, integral_constant<int, 0>{});
                annihilate(pat);
#endif
                publish_operands(next_pat);
                t1 = t2 = EIG_STEP_CLOCK();
                const double2_loc rot_new = publish_rotations(buf ^ 1);
#if !(PCSF_EIG_ABL & 8)
                b_blocks(pat, next_pat, integral_constant<int, 1>{});
#endif
                store_rotations(rot_new, buf ^ 1);
            }
            const unsigned long long t3 = EIG_STEP_CLOCK();
            buf ^= 1;
            // The next step's rotations come straight back (eight reads behind the store, in order), and the barrier only waits for the
            // STORE: wave 1 reads the rotations after it, this wave's own reads land while it passes the barrier.
            load_rotations(buf);
            asm volatile("s_waitcnt lgkmcnt(8)\n\ts_barrier" ::: "memory");
            if (EIG_STEP_STAMPS && stamping) {
                acc_a += t1 - t0;
                acc_v += t2 - t1;
                acc_rot += t3 - t2;
                acc_b += eig_clock() - t3;
            }
        };
        // Convergence is checked before every ROUND (4 steps; 7 in round 0), not once per sweep: the iteration converges quadratically,
        // so a job is usually done a good part of a sweep before the sweep's end, and a launch lasts as long as its slowest job.
        int r = 0;   // the seating: round within the sweep
        bool converged = false;
        double fro2 = 0.0;   // |B|_F^2 (rotations keep it)
        for (int round = 0; round < 60 * 15; ++round) {
            // The stop test proper costs 128 fp64 instructions and two wave reductions; it can only pass once the off-diagonal mass is
            // below 1e-18 of the matrix's, and then every rotation published afterwards has |2 B[a][b]| below 2e-18 |B|_F.  So the test is
            // run only after a round whose rotations all stayed below 1e-12 |B|_F (one ballot): at worst it passes a round late.
            const bool test = round == 0 || __builtin_amdgcn_ballot_w64(hmax * hmax > 1e-24 * fro2) == 0ull;
            hmax = 0.0;
            double off = 1.0, total = 0.0;
#if PCSF_EIG_ABL & 4
            if (false)
#endif
            if (test) {
                off = 0.0;
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const double x = b[i][j] * b[i][j];
                        if (i == j) {
                            total += diag ? x : 0.0;
                            off += diag ? 0.0 : x;
                        } else {
                            off += x;
                        }
                    }
                for (int sft = 32; sft > 0; sft >>= 1) {
                    off += __shfl_xor(off, sft);
                    total += __shfl_xor(total, sft);
                }
                fro2 = off + total;
            }
            // The host solver and the LDS form stop at off^2 <= 1e-58 diag^2.  Here: 1e-36, i.e. an off-diagonal mass below 1e-18 of the
            // diagonal's -- already under the rounding of every product formed from the result.
            constexpr double STOP = 1e-36;
#ifdef PCSF_EIG_FIXED_ROUNDS
            off = round >= PCSF_EIG_FIXED_ROUNDS ? 0.0 : 1.0;
            total = 1.0;
#endif
            if (lane == 0) flag = (off <= STOP * total) ? 2 : 0;
            __syncthreads();
            if (off <= STOP * total) {
                converged = true;
                break;
            }
            using std::integral_constant;
            if (r == 0) {
                n_sweeps += 1;
                timed_step(integral_constant<int, 0>{}, integral_constant<int, 1>{}, -1, 0);
                timed_step(integral_constant<int, 1>{}, integral_constant<int, 2>{}, -1, 0);
                timed_step(integral_constant<int, 2>{}, integral_constant<int, 3>{}, -1, 0);
            }
            timed_step(integral_constant<int, 3>{}, integral_constant<int, 4>{}, -1, 0);
            timed_step(integral_constant<int, 4>{}, integral_constant<int, 5>{}, -1, 0);
            timed_step(integral_constant<int, 5>{}, integral_constant<int, 6>{}, -1, 0);
            if (r == 14) timed_step(integral_constant<int, 6>{}, integral_constant<int, 0>{}, 14, 0);
            else timed_step(integral_constant<int, 6>{}, integral_constant<int, 3>{}, r, r + 1);
            n_steps += r == 0 ? 7 : 4;
            r = r == 14 ? 0 : r + 1;
        }
        // Rounds used up without meeting the stop rule (a NaN in the rate matrix makes every comparison false; so would a matrix the
        // iteration cannot finish): the eigen-system written below is whatever the registers hold -- the evaluation is flagged
        // (NaN row, like a model the reference fails to instantiate) instead of passing silently into the pruning kernels.
        if (!converged && lane == 0) p.fail[job.table] = 1u;
        // eigenvalues into the canonical matrix (the seating of the round the iteration stopped in)
        if (diag) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int row = 4 * rr::group_of(r, R, i >> 2) + (i & 3);
                A[row * LD + row] = b[i][i];
            }
        }
    } else {
        // V <- V J: lane = row of V, all 64 entries of the row in registers in SEAT order (seat 8G + s = slot s of super-group G), so
        // every rotation is lane-local, the step patterns are static register pairs, and regrouping is a static register shift.
        double v[64];
#pragma unroll
        for (int G = 0; G < 8; ++G)
#pragma unroll
            for (int sl = 0; sl < 8; ++sl) v[8 * G + sl] = V[lane * LD + 4 * rr::group_of(0, G, sl >> 2) + (sl & 3)];
        // The 32 rotations of a step are 32 broadcast reads (the same address in every lane).  Left to itself the compiler keeps two or
        // three of them in flight and the step waits out ~12 LDS round trips -- longer than wave 0's whole step; here they are issued in
        // batches of eight (two super-groups), each batch one ahead of the arithmetic that uses it.
        auto v_step = [&](auto pat_tag, int buf) {
            constexpr int PAT = decltype(pat_tag)::value;
            double2_loc rn[2][8];
            auto fetch = [&](int batch, double2_loc (&dst)[8]) {
#pragma unroll
                for (int u = 0; u < 8; ++u) dst[u] = rt[32 * buf + batch * 8 + u];
            };
            fetch(0, rn[0]);
#if PCSF_EIG_ABL & 16
            if (false)
#endif
#pragma unroll
            for (int batch = 0; batch < 4; ++batch) {
                __builtin_amdgcn_sched_barrier(0);
                if (batch + 1 < 4) fetch(batch + 1, rn[(batch + 1) & 1]);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int G = batch * 2 + (u >> 2), k = u & 3;
                    const double2_loc r2 = rn[batch & 1][u];
                    const int a = 8 * G + rr::pair_a(PAT, k), bq = 8 * G + rr::pair_b(PAT, k);
                    const double x = v[a], y = v[bq];
                    v[a] = __builtin_fma(r2.x, x, -(r2.y * y));
                    v[bq] = __builtin_fma(r2.y, x, r2.x * y);
                }
            }
        };
        // the 15 moving groups form one cycle (0,0) <- (1,0) <- ... <- (7,0) <- (7,1) <- (6,1) <- ... <- (1,1) <- (0,0): walk it in place
        auto shift = [&]() {
            double keep[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) keep[q] = v[q];
            int G = 0, half = 0;
#pragma unroll
            for (int n = 0; n < 14; ++n) {
                const int src = rr::shift_source(G, half), sG = src & 7, sh = src >> 3;
#pragma unroll
                for (int q = 0; q < 4; ++q) v[8 * G + 4 * half + q] = v[8 * sG + 4 * sh + q];
                G = sG;
                half = sh;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) v[8 * 1 + 4 * 1 + q] = keep[q];   // (1,1) takes over from (0,0)
        };
        int buf = 0;
        int r = 0;
        for (int round = 0; round < 60 * 15; ++round) {
            __syncthreads();
            if (flag == 2) break;
            if (r == 0) {
                v_step(std::integral_constant<int, 0>{}, buf);
                buf ^= 1;
                __syncthreads();
                v_step(std::integral_constant<int, 1>{}, buf);
                buf ^= 1;
                __syncthreads();
                v_step(std::integral_constant<int, 2>{}, buf);
                buf ^= 1;
                __syncthreads();
            }
            v_step(std::integral_constant<int, 3>{}, buf);
            buf ^= 1;
            __syncthreads();
            v_step(std::integral_constant<int, 4>{}, buf);
            buf ^= 1;
            __syncthreads();
            v_step(std::integral_constant<int, 5>{}, buf);
            buf ^= 1;
            __syncthreads();
            v_step(std::integral_constant<int, 6>{}, buf);
            shift();
            buf ^= 1;
            __syncthreads();
            r = r == 14 ? 0 : r + 1;
        }
        // eigenvectors back into the canonical matrix (the seating of the round the iteration stopped in)
#pragma unroll
        for (int G = 0; G < 8; ++G)
#pragma unroll
            for (int sl = 0; sl < 8; ++sl) V[lane * LD + 4 * rr::group_of(r, G, sl >> 2) + (sl & 3)] = v[8 * G + sl];
    }
    __syncthreads();
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
        if (tid == (REG ? 0 : 64)) atomicAdd(p.stamps + 2, acc_v);
    }

    // eigen-system of Q: right = D^-1/2 V (eigenvectors in columns), left = V^T D^1/2 (in rows)
    double* __restrict__ R = p.eig_right ? p.eig_right + static_cast<size_t>(job.eig) * 4096 : nullptr;   // (null: only the fragment-ordered copies are wanted)
    double* __restrict__ L = p.eig_left ? p.eig_left + static_cast<size_t>(job.eig) * 4096 : nullptr;
    double* __restrict__ lam = p.eig_lambda + static_cast<size_t>(job.eig) * 64;
    double* __restrict__ basis = (p.eig_basis && (p.basis_slots == 0 || job.eig < p.basis_slots)) ? p.eig_basis + static_cast<size_t>(job.eig) * 4096 : nullptr;
    for (int idx = tid; idx < 4096; idx += NT) {
        const int r = idx >> 6, c = idx & 63;
        if (R) R[idx] = iroot[r] * V[r * LD + c];   // right[i][k], i = r, k = c
        if (L) L[idx] = V[c * LD + r] * root[c];    // left[k][i],  k = r, i = c
        if (basis) basis[idx] = V[r * LD + c];
    }
    if (p.eig_right_frag) {   // ... and in prune_factored's A-operand order (kernels.h FactoredParams): coalesced loads for every tile that uses the system
        double* __restrict__ Rf = p.eig_right_frag + static_cast<size_t>(job.eig) * 4096;
        double* __restrict__ Lf = p.eig_left_frag + static_cast<size_t>(job.eig) * 4096;
        for (int idx = tid; idx < 4096; idx += NT) {
            const int lane = idx & 63, r = 16 * (idx >> 10) + (lane & 15), c = 4 * ((idx >> 6) & 15) + (lane >> 4);
            Rf[idx] = iroot[r] * V[r * LD + c];
            Lf[idx] = V[c * LD + r] * root[c];
        }
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
        if (p.eig_w) p.eig_w[static_cast<size_t>(job.eig) * 64 + lane_pos] = root[tid] / iroot[tid];
        if (p.eig_c) {   // c_k = sum_i left[k][i], k = tid
            double c = 0.0;
            for (int i = 0; i < 64; ++i) c += V[i * LD + tid] * root[i];
            p.eig_c[static_cast<size_t>(job.eig) * 64 + tid] = c;
        }
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
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&eigen_omega_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&eigen_omega_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (e != hipSuccess) return e;
        attr_set.fetch_or(uint64_t(1) << (dev & 63), std::memory_order_release);
    }
    // Two forms of the same iteration.  Default: blocks in registers (128 threads: wave 0 holds B, wave 1 V; two work-groups per CU) --
    // 0.69 ms per cold launch against 0.98 ms, and OMEGA 1.16-1.22x faster at 516 / 2,064 / 8,256 blocks.  PCSF_EIGEN_LDS=1: the
    // first version (matrices in LDS, 256 threads), kept for A/B runs and tests.
    const char* lds_env = getenv("PCSF_EIGEN_LDS");   // read at every launch: tests switch forms inside one process
    const bool in_lds = lds_env && lds_env[0] == '1';
    if (in_lds) hipLaunchKernelGGL(eigen_omega_kernel<false>, dim3(p.n_jobs), dim3(256), lds, stream, p);
    else hipLaunchKernelGGL(eigen_omega_kernel<true>, dim3(p.n_jobs), dim3(128), lds, stream, p);
    return hipGetLastError();
}

// tests: the 63 x 32 index pairs of one sweep of the register-resident schedule, in step order, and locate() of every group
void eigen_schedule_for_tests(int* pairs /* [63][32][2] */, int* located /* [15][16] */)
{
    int step = 0;
    for (int r = 0; r < 15; ++r)
        for (int pat = (r == 0 ? 0 : 3); pat < 7; ++pat, ++step)
            for (int G = 0; G < 8; ++G)
                for (int k = 0; k < 4; ++k) {
                    const int a = rr::pair_a(pat, k), b = rr::pair_b(pat, k);
                    pairs[(step * 32 + G * 4 + k) * 2 + 0] = 4 * rr::group_of(r, G, a >> 2) + (a & 3);
                    pairs[(step * 32 + G * 4 + k) * 2 + 1] = 4 * rr::group_of(r, G, b >> 2) + (b & 3);
                }
    for (int r = 0; r < 15; ++r)
        for (int g = 0; g < 16; ++g) located[r * 16 + g] = rr::locate(r, g);
}

}  // namespace pcsf
