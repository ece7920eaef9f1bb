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
#include <utility>

#include "kernels.h"

#ifndef PCSF_EIG_ABL
#define PCSF_EIG_ABL 0
#endif
// per-phase stamps inside a step of the register-resident form: a build flag (-DPCSF_EIGEN_STEP_STAMPS=1, tools/build_variant.sh), because
// even a never-taken stamp branch splits the step into basic blocks and stops the compiler from overlapping its phases
#ifndef PCSF_EIGEN_STEP_STAMPS
#define PCSF_EIGEN_STEP_STAMPS 0
#endif
// diagnostics (-DPCSF_EIGEN_PHASE_PRINT=1): work-group 0 prints the s_memtime cycles between the marks of its prologue and epilogue
#ifndef PCSF_EIGEN_PHASE_PRINT
#define PCSF_EIGEN_PHASE_PRINT 0
#endif
#if PCSF_EIGEN_PHASE_PRINT
#define EIG_PHASE(name)                                                                                  \
    do {                                                                                                 \
        if (blockIdx.x == 0 && threadIdx.x == 0) {                                                       \
            const unsigned long long now_ = eig_clock();                                                 \
            printf("eigen phase %llu cycles: " name "\n", now_ - phase_clock);                           \
            phase_clock = eig_clock();                                                                   \
        }                                                                                                \
    } while (0)
#else
#define EIG_PHASE(name) do { } while (0)
#endif
#ifndef PCSF_F2_RESEAT_STAMPS
#define PCSF_F2_RESEAT_STAMPS 0   // FORM 2, with the step-stamps build: the stamps' four slots split wave 0's re-seatings instead of its steps
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

// ---- seven-wave variant (FORM 2, round 5): the same schedule on QUADS of lanes ------------------------------------------------------
// rr's rounds, seating and step patterns.  A cross step (pattern 3 + t) pairs seat k with seat 4 + ((k + t) & 3): the two columns a lane
// holds.  The three steps of round 0 for the pairs inside the groups (patterns 0..2) pair seat e of a group with seat e ^ X, X = 1, 2, 3:
// they run in the t = 0 layout, the partner column in lane j ^ X of the quad.
namespace rq {
// canonical index of the seat (8 G + q) in round r -- rr's own seating
__host__ __device__ __forceinline__ int canon(int r, int seat) { return 4 * rr::group_of(r, seat >> 3, (seat >> 2) & 1) + (seat & 3); }
// DPP quad_perm controls: lane j reads lane (j + t) & 3 / lane j ^ x
constexpr int quad_rot(int t) { return ((0 + t) & 3) | (((1 + t) & 3) << 2) | (((2 + t) & 3) << 4) | (((3 + t) & 3) << 6); }
constexpr int quad_xor(int x) { return (0 ^ x) | ((1 ^ x) << 2) | ((2 ^ x) << 4) | ((3 ^ x) << 6); }
// one of two REGISTERS: written as `c ? a[i] : a[k]` the compiler makes it one load from a selected address -- and an array that is
// indexed at run time lives in scratch memory
__device__ __forceinline__ double pick(bool c, double a, double b)
{
    asm("" : "+v"(a), "+v"(b));
    return c ? a : b;
}
template <int CTRL>
__device__ __forceinline__ double quad_perm(double x)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// The end of a round as a V wave sees it (W = 0: seats of the super-groups 0..3, W = 1: 4..7, 8 G + q each): every group of four seats
// but (0, 1) moves one place round the table (rr::shift_source); the cycle (0,0) <- (1,0) <- ... <- (7,0) <- (7,1) <- (6,1) <- ... <- (1,1)
// <- (0,0) crosses the two waves at (3,0) <- (4,0) and (4,1) <- (3,1): `leaving` is what the wave hands over, `arriving` what it receives.
template <int W>
__host__ __device__ __forceinline__ constexpr int leaving_group() { return W == 0 ? 28 : 0; }   // first seat of (3,1) / of (4,0), wave-local
template <int W, typename T>
__host__ __device__ __forceinline__ void shift_groups(T* v, const T* arriving)
{
    auto mv = [&](int to, int from) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[to + q] = v[from + q];
    };
    if constexpr (W == 0) {
        T keep[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) keep[q] = v[q];
        mv(0, 8);     // (0,0) <- (1,0)
        mv(8, 16);    // (1,0) <- (2,0)
        mv(16, 24);   // (2,0) <- (3,0)
#pragma unroll
        for (int q = 0; q < 4; ++q) v[24 + q] = arriving[q];   // (3,0) <- (4,0)
        mv(28, 20);   // (3,1) <- (2,1)
        mv(20, 12);   // (2,1) <- (1,1)
#pragma unroll
        for (int q = 0; q < 4; ++q) v[12 + q] = keep[q];       // (1,1) <- (0,0)
    } else {
        mv(0, 8);     // (4,0) <- (5,0)
        mv(8, 16);    // (5,0) <- (6,0)
        mv(16, 24);   // (6,0) <- (7,0)
        mv(24, 28);   // (7,0) <- (7,1)
        mv(28, 20);   // (7,1) <- (6,1)
        mv(20, 12);   // (6,1) <- (5,1)
        mv(12, 4);    // (5,1) <- (4,1)
#pragma unroll
        for (int q = 0; q < 4; ++q) v[4 + q] = arriving[q];    // (4,1) <- (3,1)
    }
}
}  // namespace rq

// FORM 0: 256 threads, B and V in LDS (the first version).  FORM 1: 128 threads, wave 0 holds B, wave 1 V, in registers (rounds 2-4; two
// work-groups fit a CU).  FORM 2 (round 5): 448 threads -- a rotation wave, four waves of B in quads of lanes, two waves of V (see there).
template <int FORM>
__global__ __launch_bounds__(FORM == 1 ? 128 : FORM == 2 ? 448 : 256, FORM == 2 ? 4 : 1) void eigen_omega_kernel(const EigenParams p)
{
    constexpr int NT = FORM == 1 ? 128 : FORM == 2 ? 448 : 256;     // threads
    const bool stamping = p.stamps != nullptr && blockIdx.x == 0;
    const unsigned long long t_kernel = stamping ? eig_clock() : 0ull;
#if PCSF_EIGEN_PHASE_PRINT
    unsigned long long phase_clock = eig_clock();
#endif
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

    // (the job's fields are read where they are used -- uniform scalar loads; its twelve settings go through LDS: a local copy of the
    // struct is indexed at run time by f3x4_weight and lives in scratch memory)
    const EigenJob& job_ref = p.jobs[blockIdx.x];
    struct { uint32_t eig, table; int32_t warm; } job{job_ref.eig, job_ref.table, job_ref.warm};
    __shared__ double qs_s[12];
    const double* qs = qs_s;
    const int tid = threadIdx.x;
    if (tid < 12) qs_s[tid] = job_ref.qs[tid];
    if (tid == 0) flag = 0;
    // a warm start's eigenbasis is on its way from HBM while the rate matrix is built (it used to be fetched behind it: 7,600 cycles)
    constexpr int V0_PER_THREAD = (4096 + NT - 1) / NT;
    const bool warm_start = job.warm >= 0 && p.eig_basis != nullptr;
    double v0_pre[V0_PER_THREAD];
    if (warm_start) {
        const double* __restrict__ V0 = p.eig_basis + static_cast<size_t>(job.warm) * 4096;
#pragma unroll
        for (int k = 0; k < V0_PER_THREAD; ++k) v0_pre[k] = tid + k * NT < 4096 ? V0[tid + k * NT] : 0.0;
    }
    __syncthreads();
    if (tid < 64) {
        const double stop_mass = f3x4_weight(qs, 48) + f3x4_weight(qs, 50) + f3x4_weight(qs, 56);  // TAA, TAG, TGA
        const double denom = 1.0 - ((1.0 - qs[2]) * stop_mass);
        pi[tid] = f3x4_weight(qs, tid) / denom;
        aa_s[tid] = amino_acid_dev(tid);
    }
    __syncthreads();
    EIG_PHASE("pi");
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
    EIG_PHASE("rate fill");
    if (flag) {  // pi has a non-positive entry: the model is invalid (host: eigen_reversible returns false)
        if (tid == 0) p.fail[job.table] = 1u;
        return;
    }
    if (tid < 64) {
        double diag = 0.0;   // (in index order like the host; the row's own place holds 0 and x - 0 = x: no branch between the loads)
#pragma unroll 16
        for (int j = 0; j < 64; ++j) diag -= j != tid ? A[tid * LD + j] : 0.0;
        A[tid * LD + tid] = diag;
    }
    __syncthreads();
    EIG_PHASE("diagonal");
    if (tid == 0) {
        double norm = 0.0;
        for (int i = 0; i < 64; ++i) norm -= pi[i] * A[i * LD + i];
        norm_s = norm;
    }
    __syncthreads();
    EIG_PHASE("normaliser");
    for (int idx = tid; idx < 4096; idx += NT) {
        const int i = idx >> 6, j = idx & 63;
        A[i * LD + j] = root[i] * (A[i * LD + j] / norm_s) * iroot[j];
    }
    __syncthreads();
    EIG_PHASE("scaling");
    for (int idx = tid; idx < 4096; idx += NT) {
        const int i = idx >> 6, j = idx & 63;
        if (i < j) {
            const double m = 0.5 * (A[i * LD + j] + A[j * LD + i]);
            A[i * LD + j] = m;
            A[j * LD + i] = m;
        }
    }
    __syncthreads();
    EIG_PHASE("symmetric form");

    const int lane = tid & 63, wave = tid >> 6;
    // Warm start: an optimiser moves kappa in small steps, so the eigenbasis V0 of an earlier matrix nearly diagonalises
    // this one.  B <- V0^T B V0 (two 64^3 products, each thread 16 entries), V <- V0, and the sweeps below finish the
    // job in 2-4 rounds instead of ~9.  V0 is exactly as orthogonal as the iteration left it, so the result is an
    // eigen-decomposition of B to the same accuracy as a cold start.
    if (warm_start) {
#pragma unroll
        for (int k = 0; k < V0_PER_THREAD; ++k) {
            const int idx = tid + k * NT;
            if (idx < 4096) V[(idx >> 6) * LD + (idx & 63)] = v0_pre[k];
        }
        __syncthreads();
    EIG_PHASE("V0 from HBM");
        // Both 64^3 products on the matrix cores (v_mfma_f64_16x16x4_f64: A operand lane (s, g) = A[s][g], B operand = B[g][s], result
        // register r = row 4r + g, column s): wave w owns the output tiles of rows 16 TI w .. 16 TI (w + 1) - 1, 16 k-steps per tile, the
        // operands read from the two LDS matrices (row length 65: two lanes per bank at worst).  As scalar loops -- 32 entries and 4,096
        // LDS reads per thread and product -- they were 50 us of a 0.31 ms warm solve (profiles/r03/eigen_kernel.txt, variant R0).
        typedef double double4_loc __attribute__((ext_vector_type(4)));
        constexpr int TI = NT >= 256 ? 1 : 4 / (NT / 64);   // row tiles per wave
        const int mw = wave * TI < 4 ? wave : 4 / TI - 1;     // (waves beyond the fourth repeat its tile: the same values to the same places)
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
            for (int a = 0; a < TI; ++a) fa[a] = A[(16 * (mw * TI + a) + ms) * LD + 4 * kk + mg];
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
                for (int r = 0; r < 4; ++r) A[(16 * (mw * TI + a) + 4 * r + mg) * LD + 16 * b + ms] = acc[a][b][r];
        __syncthreads();
    EIG_PHASE("product 1");
        // C = V0^T T: A operand V0^T[i][k] = V0[k][i]
#pragma unroll
        for (int a = 0; a < TI; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = double4_loc{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
        for (int kk = 0; kk < 16; ++kk) {
            double fa[TI], fb[4];
#pragma unroll
            for (int a = 0; a < TI; ++a) fa[a] = V[(4 * kk + mg) * LD + 16 * (mw * TI + a) + ms];
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
                for (int r = 0; r < 4; ++r) A[(16 * (mw * TI + a) + 4 * r + mg) * LD + 16 * b + ms] = acc[a][b][r];
        __syncthreads();
    EIG_PHASE("product 2");
        for (int idx = tid; idx < 4096; idx += NT) {  // exact symmetry again
            const int i = idx >> 6, j = idx & 63;
            if (i < j) {
                const double m = 0.5 * (A[i * LD + j] + A[j * LD + i]);
                A[i * LD + j] = m;
                A[j * LD + i] = m;
            }
        }
        __syncthreads();
    EIG_PHASE("symmetry again");
    }
    const unsigned long long t_sweeps = stamping ? eig_clock() : 0ull;
    EIG_PHASE("(before the sweeps)");   // (the rate matrix, its symmetric form and a warm start's two products are behind)
    if constexpr (FORM == 0) {
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
    } else if constexpr (FORM == 1) {
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
    } else {
    // ================= seven waves: a rotation wave, four waves of B (a quad of lanes per 8x8 block), two waves of rows of V =================
    // Why (profiles/r05/eigen_forms/README.txt): a step of FORM 1 is ONE wave's serial instruction stream -- 256 fp64 instructions of B
    // update, the next pairs' operands to LDS and back, the rotation chain, the rotations to LDS, a barrier: ~2,450 cycles.  Here
    //  * the 8x8 block (R, C) of B lives in a QUAD of lanes: lane j holds the block's column j and ONE column of the second group, in
    //    step t column 4 + ((j + t) & 3) -- the two columns of the lane's rotation pair -- with the rows in the lane's own order
    //    (position m = row (j + m) & 3 of either group), so that every operand of a step sits in a register known at compile time: 32
    //    FMAs per lane and step, and between two steps the moving column travels one lane round the quad (DPP quad_perm, no LDS);
    //  * wave 0 keeps COPIES of the eight diagonal blocks (updated with the same instructions as the originals: same bits) and makes
    //    each step's 32 rotations from them, one per lane, all three operands in the lane's own registers; it owns D;
    //  * waves 5-6, lane = row of V, 32 columns each in seat order.
    // Scaled ("fast") rotations: B = D Bs D, V = Vs D with a diagonal D that absorbs every cosine, so a rotation by tan t is
    // x <- x - alpha y, y <- y + beta x (alpha = t d_b / d_a, beta = t d_a / d_b; d <- d cos): two FMAs per element pair, and t, which
    // only decides how well the pair is annihilated, never orthogonality, comes from an fp32 chain.
    // One LDS hand-over (the rotations) and one barrier per step; every fourth step the blocks of B pass through LDS to their next seats.
    // The three steps of round 0 for the pairs INSIDE the groups (seat e with seat e ^ X of the same group) run in place, in the t = 0 layout:
    // the partner column is in lane j ^ X of the quad, the partner row in the lane's own registers -- at a position that depends on the
    // lane, so these steps select between two registers where a cross step names one (~2x a cross step's instructions, three steps of 63;
    // their first version gave each of them a seating of its own in which it WAS a cross step: three more re-seatings per sweep, 2,000
    // cycles each against ~700 for the selects).  Their rotations travel as one coefficient per SEAT: new[s] = old[s] + coef[s] old[s ^ X].
    double2_loc* rt = reinterpret_cast<double2_loc*>(rot);         // [buf][super-group * 4 + pair] = (alpha, beta)
    __shared__ __attribute__((aligned(16))) double dtab[64 * 2];   // (d, 1/d) per physical seat: written for a re-seating
    __shared__ double dtab2[64];                                   // d per physical seat: written for a stop test
    __shared__ int zfl[2 * 64];                                    // [buf][super-group * 4 + pair] (inside steps: [buf][seat]): the pair is set to zero behind its rotation
    __shared__ int tflag;                                          // wave 0 -> all: run the stop test after this re-seating
    double* coef = rot;                                            // inside steps: [buf][seat], the same 1 KB as rt
    auto rsqrt_nr = [](double x) {
        double y = __builtin_amdgcn_rsq(x);
        const double hx = 0.5 * x;
#pragma unroll
        for (int it = 0; it < 3; ++it) y = y * __builtin_fma(-hx, y * y, 1.5);
        return y;
    };
    // tangent of the annihilating rotation from the SCALED entries and rho = d_b / d_a, irho = d_a / d_b: B[b][b] - B[a][a] and 2 B[a][b], both
    // divided by d_a d_b > 0, are rho Bs[b][b] - irho Bs[a][a] and 2 Bs[a][b]; t = h sgn(d) / (|d| + sqrt(d^2 + h^2)) as in FORM 1's rotation_of,
    // on operands scaled by a power of two, in fp32.  Returns (alpha, beta).
    auto rotation_scaled = [&](double baa, double bbb, double bab, double rho, double irho) {
        const double d = __builtin_fma(rho, bbb, -(irho * baa));
        const double h = bab + bab;
        const int e = __builtin_amdgcn_frexp_exp(fabs(d) + fabs(h));
        const float df = static_cast<float>(__builtin_amdgcn_ldexp(d, -e)), hf = static_cast<float>(__builtin_amdgcn_ldexp(h, -e));
        const float x = __builtin_fmaf(df, df, hf * hf);                      // in [1/8, 1) unless both vanish
        const float w = __builtin_fabsf(df) + __builtin_amdgcn_sqrtf(x);
        float t = (df < 0.0f ? -hf : hf) * __builtin_amdgcn_rcpf(w);         // sgn(0) = +1: 45 degrees
        t = (hf == 0.0f) ? hf : t;                                            // no rotation (0 / 0 otherwise); a NaN is not == 0 and stays
        const double td = static_cast<double>(t);
        return double2_loc{td * rho, td * irho};
    };
    using std::integral_constant;
    // The roles run the SAME sequence of barriers from specialisations of one body (ROLE 0: the rotation wave, 1: blocks of B, 2 / 3: the
    // first / second four super-groups of V's columns), so that a wave's registers hold only what its role needs.
    auto body = [&](auto role_tag) {
    constexpr int ROLE = decltype(role_tag)::value;
    constexpr bool MK = ROLE == 0, BW = ROLE == 1, VW = ROLE >= 2;
    constexpr int W = VW ? ROLE - 2 : 0;
    const int j = lane & 3;
    const int blk = MK ? 9 * ((lane >> 2) & 7) : ((tid - 64) >> 2);   // 8 R + C (wave 0: the diagonal blocks, twice)
    const int R = blk >> 3, C = blk & 7;
    const bool diag = R == C;
    auto rowseat = [&](int pos) { return (pos & 4) + ((j + pos) & 3); };   // the row (physical seat within the super-group) at a lane's position
    double c0[8], c1[8];   // column j / the moving column of the block, by position
    double v[32];          // V waves: row `lane`, seats 8 Gl + q of the super-groups 4 W + Gl
    if constexpr (!VW) {
#pragma unroll
        for (int pos = 0; pos < 8; ++pos) {
            const int rc = rq::canon(0, 8 * R + rowseat(pos));
            c0[pos] = A[rc * LD + rq::canon(0, 8 * C + j)];
            c1[pos] = A[rc * LD + rq::canon(0, 8 * C + 4 + j)];
        }
    } else {
#pragma unroll
        for (int s = 0; s < 32; ++s) v[s] = V[lane * LD + rq::canon(0, 32 * W + s)];
    }
    __syncthreads();   // everything is in registers: A becomes the staging matrix of the re-seatings (by physical seat), V the exchange buffer
    double* S = A;
    const int lds_base_S = static_cast<int>(reinterpret_cast<size_t>(S));   // (LDS byte address, for the hand-written loads of a round's end)
    double2_loc* xch = reinterpret_cast<double2_loc*>(V);   // [V wave][row][2]: the group of four columns that changes waves at a round's end
    // Staging for a round's end is LANE-MAJOR -- the B lane k stores position pos of its column slot at (2 pos + slot) * 256 + k, as it stands
    // after the round's last step (t = 3: the moving column of lane j is 4 + ((j + 3) & 3)) -- so stores and loads are free of bank conflicts
    // (consecutive lanes, consecutive double-words; a load's 32-lane half covers the eight blocks of a block row) and all addresses are a
    // register plus a constant.  (Row-major by seat, the first version: 16 lane-dependent addresses, seven of them spilled to scratch and
    // reloaded behind every barrier, and the blocks' 8 R' and 8 C' left one bit each in the bank number.)
    // A round's end keeps the within-group index: the new half x of the rows comes from the half h'x of super-group R'x, the new column
    // slot y from the group g'y of super-group C'y.  A first group's column e is column slot 0 of lane e, same positions.  A second group's
    // column e was the moving column of lane (e + 1) & 3, whose position m - 1 is this lane's m: the lane's position 0 comes from smapP, the
    // positions 1..3 from smapQ + 512 (m - 1) -- two registers per (x, y) instead of a turn of the moving column before the staging.
    int smapP[2][2] = {{0, 0}, {0, 0}}, smapQ[2][2] = {{0, 0}, {0, 0}};
    int dseat0[2] = {0, 0};   // wave 0: where the lane's two seats' (d, 1/d) come from
    if constexpr (!VW) {
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
            for (int y = 0; y < 2; ++y) {
                const int sr = rr::shift_source(R, x), sc = rr::shift_source(C, y);
                const int second = sc >> 3;
                const int base = 2048 * (sr >> 3) + 32 * (sr & 7) + 256 * second + 4 * (sc & 7) + (second ? (j + 1) & 3 : j);
                smapP[x][y] = second ? base + 1536 : base;
                smapQ[x][y] = second ? base : base + 512;
                if (x == 0) dseat0[y] = 8 * (sc & 7) + 4 * second + j;
            }
    }
    double dA = 1.0, idA = 1.0, dB = 1.0, idB = 1.0;   // wave 0: (d, 1/d) of the seats j and 4 + j of super-group R
    double2_loc ab = double2_loc{0.0, 0.0};             // ... its (alpha, beta)
    int zn = 0;                                         // ... and whether the pair is small enough to be set to zero behind it
    double hmax = 0.0, fro2 = 0.0;
    int buf = 0;
    // B <- J^T B J of the cross step T on the scaled entries
    auto update = [&](auto t_tag) {
        constexpr int T = decltype(t_tag)::value;
        double ac, bc, al[4], bl[4];
        int zl;
        if constexpr (MK) {
            ac = ab.x;
            bc = ab.y;
            zl = zn;
            al[0] = ab.x;
            bl[0] = ab.y;
            al[1] = rq::quad_perm<rq::quad_rot(1)>(ab.x);
            bl[1] = rq::quad_perm<rq::quad_rot(1)>(ab.y);
            al[2] = rq::quad_perm<rq::quad_rot(2)>(ab.x);
            bl[2] = rq::quad_perm<rq::quad_rot(2)>(ab.y);
            al[3] = rq::quad_perm<rq::quad_rot(3)>(ab.x);
            bl[3] = rq::quad_perm<rq::quad_rot(3)>(ab.y);
            // D <- D cos: seat j is this lane's pair's first seat; seat 4 + j is the second seat of the pair of lane (j - T) & 3.  (Here, not
            // where the rotation is made: the other waves wait for the hand-over, and nothing needs the cosine before the next rotation.)
            const double x = __builtin_fma(ab.x, ab.y, 1.0);   // 1 + tan^2
            const double pc = rsqrt_nr(x), pic = x * pc;
            dA *= pc;
            idA *= pic;
            if constexpr (T == 0) {
                dB *= pc;
                idB *= pic;
            } else {
                dB *= rq::quad_perm<rq::quad_rot((4 - T) & 3)>(pc);
                idB *= rq::quad_perm<rq::quad_rot((4 - T) & 3)>(pic);
            }
        } else {
            const double2_loc cc = rt[32 * buf + 4 * C + j];
            ac = cc.x;
            bc = cc.y;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const double2_loc rw = rt[32 * buf + 4 * R + ((j + m) & 3)];
                al[m] = rw.x;
                bl[m] = rw.y;
            }
            zl = zfl[64 * buf + 4 * R + j];
        }
#pragma unroll
        for (int pos = 0; pos < 8; ++pos) {   // columns (j, 4 + ((j + T) & 3)): the lane's pair
            const double x = c0[pos], y = c1[pos];
            c0[pos] = __builtin_fma(-ac, y, x);
            c1[pos] = __builtin_fma(bc, x, y);
        }
#pragma unroll
        for (int m = 0; m < 4; ++m) {         // rows: position m of the first group with position (m + T) & 3 of the second = pair (j + m) & 3
            const int pb = 4 + ((m + T) & 3);
            const double x0 = c0[m], y0 = c0[pb], x1 = c1[m], y1 = c1[pb];
            c0[m] = __builtin_fma(-al[m], y0, x0);
            c0[pb] = __builtin_fma(bl[m], x0, y0);
            c1[m] = __builtin_fma(-al[m], y1, x1);
            c1[pb] = __builtin_fma(bl[m], x1, y1);
        }
        // the rotated pair itself, once it is small (see FORM 1 / zn below): (row j, moving column) and (row of the moving column, column j)
        const bool z = diag && zl != 0;
        c1[0] = z ? 0.0 : c1[0];
        c0[4 + T] = z ? 0.0 : c0[4 + T];
        // (no pins like v_step's here: the B and rotation waves do not spill, and pinned they were 7 % slower)
    };
    // the moving column goes one lane round the quad: lane j takes over from lane (j + 1) & 3, whose position m - 1 is this lane's m
    auto shift = [&]() {
        double n[8];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            n[m] = rq::quad_perm<rq::quad_rot(1)>(c1[(m + 3) & 3]);
            n[4 + m] = rq::quad_perm<rq::quad_rot(1)>(c1[4 + ((m + 3) & 3)]);
        }
#pragma unroll
        for (int pos = 0; pos < 8; ++pos) c1[pos] = n[pos];
    };
    // wave 0: the rotation of the lane's pair (j, 4 + ((j + T) & 3)) of the coming step T from its copy of the diagonal block
    auto rotations = [&](auto t_tag, int to_buf) {
        constexpr int T = decltype(t_tag)::value;
        const double db = T == 0 ? dB : rq::quad_perm<rq::quad_rot(T)>(dB), idb = T == 0 ? idB : rq::quad_perm<rq::quad_rot(T)>(idB);
        const double baa = c0[0], bab = c1[0], bbb = c1[4 + T];
        const double h_true = fabs((bab + bab) * (dA * db));   // |2 B[a][b]| of the unscaled matrix
        hmax = fmax(hmax, h_true);
        // t is an fp32 value: the two FMAs leave ~6e-8 of the entry, and setting THAT to zero is harmless once the entry itself is below
        // 1e-12 |B|_F (without it the entries inside a degenerate cluster are re-made at rounding level in every sweep and the stop rule is never met)
        zn = h_true * h_true <= 1e-24 * fro2;
        ab = rotation_scaled(baa, bbb, bab, db * idA, dA * idb);
        rt[32 * to_buf + 4 * R + j] = ab;
        zfl[64 * to_buf + 4 * R + j] = zn;
        // (the rotation's cosine, for D, is made behind the hand-over's barrier: see update)
    };
    // ---- the steps for the pairs inside the groups: seat e of a group with seat e ^ X (X = 1, 2, 3 = rr's patterns 0, 1, 2), t = 0 layout ----
    // new[s] = old[s] + coef[s] old[s ^ X] for columns, then for rows (coef = -alpha on a pair's first seat, +beta on its second).
    // (delta1: which of two registers holds the partner's entry depends on the lane for X = 1 and 3; for X = 2 it is position + 2.)
    auto update_inside = [&](auto x_tag) {
        constexpr int X = decltype(x_tag)::value;
        const int L = j ^ X;
        if constexpr (MK) {   // D <- D cos: this lane's rotation belongs to its seat j (L > j) or 4 + j (L < j), the partner lane's to the other one
            const double x1 = __builtin_fma(ab.x, ab.y, 1.0);   // 1 + tan^2
            const double pc = rsqrt_nr(x1), pic = x1 * pc;
            const double qc = rq::quad_perm<rq::quad_xor(X)>(pc), qic = rq::quad_perm<rq::quad_xor(X)>(pic);
            const bool first = j < L;
            dA *= first ? pc : qc;
            idA *= first ? pic : qic;
            dB *= first ? qc : pc;
            idB *= first ? qic : pic;
        }
        const double cc0 = coef[64 * buf + 8 * C + j], cc1 = coef[64 * buf + 8 * C + 4 + j];
        double cr[8];
#pragma unroll
        for (int pos = 0; pos < 8; ++pos) cr[pos] = coef[64 * buf + 8 * R + rowseat(pos)];
        const int z0 = zfl[64 * buf + 8 * R + j], z1 = zfl[64 * buf + 8 * R + 4 + j];
        // columns: the partner column sits in lane L, the same row at L's position (pos + j - L) & 3: the SENDER turns its column for its
        // receiver (the lane it receives from), one DPP move per entry
        {
            const bool send1 = ((L - j) & 3) == 1;   // the receiver's (pos + delta): delta = (receiver - sender) & 3 = (L - j) & 3, 1 or 3 (X = 2: 2)
            double n0[8], n1[8];
#pragma unroll
            for (int pos = 0; pos < 8; ++pos) {
                const int h = pos & 4, m = pos & 3;
                const double s0 = X == 2 ? c0[h + ((m + 2) & 3)] : rq::pick(send1, c0[h + ((m + 1) & 3)], c0[h + ((m + 3) & 3)]);
                const double s1 = X == 2 ? c1[h + ((m + 2) & 3)] : rq::pick(send1, c1[h + ((m + 1) & 3)], c1[h + ((m + 3) & 3)]);
                n0[pos] = __builtin_fma(cc0, rq::quad_perm<rq::quad_xor(X)>(s0), c0[pos]);
                n1[pos] = __builtin_fma(cc1, rq::quad_perm<rq::quad_xor(X)>(s1), c1[pos]);
            }
#pragma unroll
            for (int pos = 0; pos < 8; ++pos) {
                c0[pos] = n0[pos];
                c1[pos] = n1[pos];
            }
        }
        // rows: row e = (j + pos) & 3 of a group with row e ^ X of the same group, at position ((e ^ X) - j) & 3
        {
            const bool up = (j & 1) == 0;   // even lanes take the first of the two candidate positions below
            double n0[8], n1[8];
#pragma unroll
            for (int pos = 0; pos < 8; ++pos) {
                const int h = pos & 4, m = pos & 3;
                // X = 1: e even -> m + 1, e odd -> m + 3, e = (j + m) & 3: for even j that is m's parity, for odd j the opposite
                // X = 3: (3 - 2 j - m) & 3: even j -> 3 - m, odd j -> (1 - m) & 3
                const int pa = X == 1 ? ((m & 1) == 0 ? (m + 1) & 3 : (m + 3) & 3) : X == 2 ? (m + 2) & 3 : (3 - m) & 3;   // even j
                const int pb = X == 1 ? ((m & 1) == 0 ? (m + 3) & 3 : (m + 1) & 3) : X == 2 ? (m + 2) & 3 : (5 - m) & 3;   // odd j
                const double y0 = X == 2 ? c0[h + pa] : rq::pick(up, c0[h + pa], c0[h + pb]);
                const double y1 = X == 2 ? c1[h + pa] : rq::pick(up, c1[h + pa], c1[h + pb]);
                n0[pos] = __builtin_fma(cr[pos], y0, c0[pos]);
                n1[pos] = __builtin_fma(cr[pos], y1, c1[pos]);
            }
#pragma unroll
            for (int pos = 0; pos < 8; ++pos) {
                c0[pos] = n0[pos];
                c1[pos] = n1[pos];
            }
        }
        // the rotated pairs themselves, once they are small: in this lane's columns the entry of the partner's row, position (L - j) & 3
        {
            const int pz = (L - j) & 3;
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                c0[m] = (diag && z0 != 0 && m == pz) ? 0.0 : c0[m];
                c1[4 + m] = (diag && z1 != 0 && m == pz) ? 0.0 : c1[4 + m];
            }
        }
    };
    // wave 0: the lane with the smaller index of a quad's lane pair (j, j ^ X) makes the rotation of the first group's pair (seats j, j ^ X),
    // the other one the second group's (4 + (j ^ X), 4 + j); both publish the two seats' coefficients and flags
    auto rotations_inside = [&](auto x_tag, int to_buf) {
        constexpr int X = decltype(x_tag)::value;
        const int L = j ^ X;
        const bool first = j < L;
        const int pm = (L - j) & 3;   // where the partner's row sits in this lane's columns
        const double dg0 = c0[0], dg1 = c1[4];
        const double other = rq::quad_perm<rq::quad_xor(X)>(first ? dg1 : dg0);   // the partner sends the diagonal entry this lane's pair needs
        const double baa = first ? dg0 : other, bbb = first ? other : dg1;
        const double e0 = X == 2 ? c0[2] : rq::pick(pm == 1, c0[1], c0[3]), e1 = X == 2 ? c1[6] : rq::pick(pm == 1, c1[5], c1[7]);
        const double bab = first ? e0 : e1;
        const double od = rq::quad_perm<rq::quad_xor(X)>(first ? dB : dA), oid = rq::quad_perm<rq::quad_xor(X)>(first ? idB : idA);
        const double da = first ? dA : od, ida = first ? idA : oid, db = first ? od : dB, idb = first ? oid : idB;
        const double h_true = fabs((bab + bab) * (da * db));
        hmax = fmax(hmax, h_true);
        zn = h_true * h_true <= 1e-24 * fro2;
        ab = rotation_scaled(baa, bbb, bab, db * ida, da * idb);
        const int sa = 8 * R + (first ? j : 4 + L), sb = 8 * R + (first ? L : 4 + j);   // the pair's first and second seat
        coef[64 * to_buf + sa] = -ab.x;
        coef[64 * to_buf + sb] = ab.y;
        zfl[64 * to_buf + sa] = zn;
        zfl[64 * to_buf + sb] = zn;
    };
    // V: every seat with its partner of the same group
    auto v_inside = [&](auto x_tag) {
        constexpr int X = decltype(x_tag)::value;
        double cf[2][8];
#pragma unroll
        for (int q = 0; q < 8; ++q) cf[0][q] = coef[64 * buf + 32 * W + q];
#pragma unroll
        for (int Gl = 0; Gl < 4; ++Gl) {
            __builtin_amdgcn_sched_barrier(0);
            if (Gl + 1 < 4) {
#pragma unroll
                for (int q = 0; q < 8; ++q) cf[(Gl + 1) & 1][q] = coef[64 * buf + 32 * W + 8 * (Gl + 1) + q];
            }
            __builtin_amdgcn_sched_barrier(0);
            double n[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) n[q] = __builtin_fma(cf[Gl & 1][q], v[8 * Gl + (q ^ X)], v[8 * Gl + q]);
#pragma unroll
            for (int q = 0; q < 8; ++q) v[8 * Gl + q] = n[q];
        }
#pragma unroll
        for (int s = 0; s < 32; ++s) asm volatile("" : "+v"(v[s]));   // (see v_step)
    };
    // V <- V J of the cross step T: the rotations are broadcast reads, fetched a super-group ahead of the arithmetic
    auto v_step = [&](auto t_tag) {
        constexpr int T = decltype(t_tag)::value;
        double2_loc rn[2][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) rn[0][k] = rt[32 * buf + 16 * W + k];
#pragma unroll
        for (int Gl = 0; Gl < 4; ++Gl) {
            __builtin_amdgcn_sched_barrier(0);
            if (Gl + 1 < 4) {
#pragma unroll
                for (int k = 0; k < 4; ++k) rn[(Gl + 1) & 1][k] = rt[32 * buf + 16 * W + 4 * (Gl + 1) + k];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double2_loc r2 = rn[Gl & 1][k];
                const int a = 8 * Gl + k, bq = 8 * Gl + 4 + ((k + T) & 3);
                const double x = v[a], y = v[bq];
                v[a] = __builtin_fma(-r2.x, y, x);
                v[bq] = __builtin_fma(r2.y, x, y);
            }
        }
        // (the step's results are "used" here: left alone, the compiler sinks the FMAs of several steps into a later basic block and keeps
        // the steps' rotations alive until then -- 200 spilled registers)
#pragma unroll
        for (int s = 0; s < 32; ++s) asm volatile("" : "+v"(v[s]));
    };
    // V: the end of a round (rq::shift_groups): the group that changes waves goes through LDS
    auto v_shift_out = [&]() {
        constexpr int g = rq::leaving_group<W>();
        xch[(W * 64 + lane) * 2 + 0] = double2_loc{v[g], v[g + 1]};
        xch[(W * 64 + lane) * 2 + 1] = double2_loc{v[g + 2], v[g + 3]};
    };
    auto v_shift_in = [&]() {
        const double2_loc i0 = xch[((1 - W) * 64 + lane) * 2 + 0], i1 = xch[((1 - W) * 64 + lane) * 2 + 1];
        const double arriving[4] = {i0.x, i0.y, i1.x, i1.y};
        rq::shift_groups<W>(v, arriving);
    };
    // Stop test (every wave takes part: two barriers): squared off-diagonal mass of the UNSCALED matrix against its squared diagonal mass
    auto stop_test = [&]() -> bool {
        if constexpr (MK) {
            dtab2[8 * R + j] = dA;
            dtab2[8 * R + 4 + j] = dB;
        }
        __syncthreads();
        if constexpr (BW) {
            double off = 0.0, total = 0.0;
            const double dc0 = dtab2[8 * C + j], dc1 = dtab2[8 * C + 4 + j];
#pragma unroll
            for (int pos = 0; pos < 8; ++pos) {
                const double dr = dtab2[8 * R + rowseat(pos)];
                const double e0 = c0[pos] * (dr * dc0), e1 = c1[pos] * (dr * dc1);
                const double x0 = e0 * e0, x1 = e1 * e1;
                // (moving column in its t = 0 place: 4 + j.)  Diagonal entries: (row j, column j) at position 0, (row 4 + j, column 4 + j) at position 4
                if (pos == 0) {
                    total += diag ? x0 : 0.0;
                    off += diag ? x1 : x0 + x1;
                } else if (pos == 4) {
                    total += diag ? x1 : 0.0;
                    off += diag ? x0 : x0 + x1;
                } else {
                    off += x0 + x1;
                }
            }
            for (int sft = 32; sft > 0; sft >>= 1) {
                off += __shfl_xor(off, sft);
                total += __shfl_xor(total, sft);
            }
            if (lane == 0) {
                red[wave - 1] = off;
                red[4 + wave - 1] = total;
            }
        }
        __syncthreads();
        const double off = red[0] + red[1] + red[2] + red[3], total = red[4] + red[5] + red[6] + red[7];
        fro2 = off + total;
#ifdef PCSF_EIG_FIXED_ROUNDS
        return false;
#endif
        return off <= 1e-36 * total;   // FORM 1's stop rule
    };
    // A round's end: everybody moves to the next seating.  The B waves stage their entries, wave 0 its seats' (d, 1/d); behind the barrier everybody collects what sits on its seats next (wave 0: its new copies),
    // a stop test runs if the round's rotations were all small, wave 0 makes the first rotations of the new round (NEXT_INSIDE: round 0
    // begins with the pairs inside the groups).  Returns whether the iteration has converged.
    auto reseat = [&](auto next_inside_tag) -> bool {
        constexpr bool NEXT_INSIDE = decltype(next_inside_tag)::value;
        bool quiet = false;
        const unsigned long long t0 = EIG_STEP_CLOCK();   // (wave 0's re-seatings as a whole -> the "V update" slot of the stamps)
        if constexpr (BW) {
#pragma unroll
            for (int pos = 0; pos < 8; ++pos) {
                S[(2 * pos) * 256 + (tid - 64)] = c0[pos];
                S[(2 * pos + 1) * 256 + (tid - 64)] = c1[pos];
            }
        } else if constexpr (MK) {
            *reinterpret_cast<double2_loc*>(&dtab[2 * (8 * R + j)]) = double2_loc{dA, idA};
            *reinterpret_cast<double2_loc*>(&dtab[2 * (8 * R + 4 + j)]) = double2_loc{dB, idB};
            // the stop test proper is only worth running once every rotation of a round stayed below 1e-12 |B|_F (see FORM 1)
            quiet = __builtin_amdgcn_ballot_w64(hmax * hmax > 1e-24 * fro2) == 0ull;
            hmax = 0.0;
            if (lane == 0) tflag = quiet ? 1 : 0;
        } else {
            v_shift_out();
        }
        const unsigned long long q1 = EIG_STEP_CLOCK();
        __syncthreads();
        const unsigned long long q2 = EIG_STEP_CLOCK();
        if constexpr (MK) {   // what the next rotation is made of first (LDS returns in order): D, then B[a][a], B[a][b], B[b][b]
            const double2_loc e0 = *reinterpret_cast<const double2_loc*>(&dtab[2 * dseat0[0]]), e1 = *reinterpret_cast<const double2_loc*>(&dtab[2 * dseat0[1]]);
            dA = e0.x;
            idA = e0.y;
            dB = e1.x;
            idB = e1.y;
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int x = 0; x < 2; ++x) {
                    c0[4 * x + m] = m == 0 ? S[smapP[x][0]] : S[smapQ[x][0] + 512 * (m - 1)];
                    c1[4 * x + m] = m == 0 ? S[smapP[x][1]] : S[smapQ[x][1] + 512 * (m - 1)];
                }
        } else if constexpr (BW) {
            // sixteen ds_read_b64 as written: the compiler pairs loads a constant apart into ds_read2st64_b64, which the LDS serves
            // at half the rate (MI355X_MICROARCH.md: 8 cycles against 2 x 2) -- with four waves collecting at once, 160 cycles a round
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int m = 0; m < 4; ++m) {
                    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(c0[4 * x + m]) : "v"(8 * (m == 0 ? smapP[x][0] : smapQ[x][0]) + lds_base_S), "n"(m == 0 ? 0 : 4096 * (m - 1)));
                    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(c1[4 * x + m]) : "v"(8 * (m == 0 ? smapP[x][1] : smapQ[x][1]) + lds_base_S), "n"(m == 0 ? 0 : 4096 * (m - 1)));
                }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
            v_shift_in();
        }
        unsigned long long q3 = 0;
        if constexpr (MK && PCSF_F2_RESEAT_STAMPS != 0) {
            if (EIG_STEP_STAMPS && stamping) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                q3 = eig_clock();
            }
        }
        if (MK ? quiet : tflag != 0) {   // (wave 0 knows without the trip through LDS)
            if (stop_test()) return true;
        }
        if constexpr (MK) {
            if constexpr (NEXT_INSIDE) rotations_inside(integral_constant<int, 1>{}, buf ^ 1);
            else rotations(integral_constant<int, 0>{}, buf ^ 1);
        }
        __syncthreads();
        buf ^= 1;
        if constexpr (MK) {
            if (EIG_STEP_STAMPS && stamping) {
                if (PCSF_F2_RESEAT_STAMPS) {   // staging / barrier / collecting / rotations and hand-over
                    acc_a += q1 - t0;
                    acc_rot += q2 - q1;
                    acc_b += q3 - q2;
                    acc_v += eig_clock() - q3;
                } else {
                    acc_v += eig_clock() - t0;
                }
            }
        }
        return false;
    };
    // one step: apply the rotations in the registers; unless a re-seating follows, move on to the next pattern and hand its rotations over
    auto step = [&](auto t_tag, auto last_tag) {
        constexpr int T = decltype(t_tag)::value;
        constexpr bool LAST = decltype(last_tag)::value;
        const unsigned long long t0 = EIG_STEP_CLOCK();   // diagnostics (-DPCSF_EIGEN_STEP_STAMPS=1), wave 0: update + shift / rotations / barrier
        if constexpr (VW) v_step(t_tag);
        else update(t_tag);
        if constexpr (!LAST) {
            if constexpr (!VW) shift();
            const unsigned long long t1 = EIG_STEP_CLOCK();
            if constexpr (MK) rotations(integral_constant<int, (T + 1) & 3>{}, buf ^ 1);
            const unsigned long long t2 = EIG_STEP_CLOCK();
            __syncthreads();
            buf ^= 1;
            if constexpr (MK) {
                if (EIG_STEP_STAMPS && !PCSF_F2_RESEAT_STAMPS && stamping) {
                    acc_a += t1 - t0;
                    acc_rot += t2 - t1;
                    acc_b += eig_clock() - t2;
                }
            }
        } else {
            if constexpr (MK) {
                if (EIG_STEP_STAMPS && !PCSF_F2_RESEAT_STAMPS && stamping) acc_a += eig_clock() - t0;
            }
        }
    };
    // a step for the pairs inside the groups (no turn of the moving column: these steps and the first cross step share the t = 0 layout)
    auto step_inside = [&](auto x_tag) {
        constexpr int X = decltype(x_tag)::value;
        const unsigned long long t0 = EIG_STEP_CLOCK();
        if constexpr (VW) v_inside(x_tag);
        else update_inside(x_tag);
        const unsigned long long t1 = EIG_STEP_CLOCK();
        if constexpr (MK) {
            if constexpr (X < 3) rotations_inside(integral_constant<int, X + 1>{}, buf ^ 1);
            else rotations(integral_constant<int, 0>{}, buf ^ 1);
        }
        const unsigned long long t2 = EIG_STEP_CLOCK();
        __syncthreads();
        buf ^= 1;
        if constexpr (MK) {
            if (EIG_STEP_STAMPS && !PCSF_F2_RESEAT_STAMPS && stamping) {
                acc_a += t1 - t0;
                acc_rot += t2 - t1;
                acc_b += eig_clock() - t2;
            }
        }
    };
    int r = 0;
    bool converged = stop_test();   // (also |B|_F^2 for the rotations' zero flags)
    if (!converged) {
        if constexpr (MK) rotations_inside(integral_constant<int, 1>{}, 0);
        __syncthreads();
        using T0 = integral_constant<int, 0>;
        using T1 = integral_constant<int, 1>;
        using T2 = integral_constant<int, 2>;
        using T3 = integral_constant<int, 3>;
        for (int round = 0; round < 60 * 15; ++round) {
            if (r == 0) {   // the pairs inside the groups
                if constexpr (MK) n_sweeps += 1;
                step_inside(T1{});
                step_inside(T2{});
                step_inside(T3{});
            }
            step(T0{}, std::false_type{});
            step(T1{}, std::false_type{});
            step(T2{}, std::false_type{});
            step(T3{}, std::true_type{});
            const bool done = r == 14 ? reseat(std::true_type{}) : reseat(std::false_type{});
            if constexpr (MK) n_steps += r == 0 ? 7 : 4;
            r = r == 14 ? 0 : r + 1;
            if (done) {
                converged = true;
                break;
            }
        }
    }
    if (!converged && tid == 0) p.fail[job.table] = 1u;   // rounds used up (a NaN makes the stop test false for ever): flagged, see FORM 1
    __syncthreads();   // nobody reads the staging buffers any more
    // eigenvalues and (scaled) eigenvectors back into the canonical matrices, in the seating the iteration stopped in
    if constexpr (MK) {
        if (lane < 32) {
            const int ca = rq::canon(r, 8 * R + j), cb = rq::canon(r, 8 * R + 4 + j);
            A[ca * LD + ca] = c0[0] * (dA * dA);
            A[cb * LD + cb] = c1[4] * (dB * dB);
        }
    } else if constexpr (VW) {
#pragma unroll
        for (int s = 0; s < 32; ++s) V[lane * LD + rq::canon(r, 32 * W + s)] = v[s];
    }
    };   // body
    if (wave == 0) body(integral_constant<int, 0>{});
    else if (wave <= 4) body(integral_constant<int, 1>{});
    else if (wave == 5) body(integral_constant<int, 2>{});
    else body(integral_constant<int, 3>{});
    __syncthreads();
    {   // V = Vs D: the columns of Vs are orthogonal by construction (every step an orthogonal transformation times a diagonal one), so D is
        // their norms' reciprocals -- taken from the columns themselves rather than from the tracked products of cosines (~500 factors each)
        if (tid < 64) {
            double n2 = 0.0;
            for (int i = 0; i < 64; ++i) n2 = __builtin_fma(V[i * LD + tid], V[i * LD + tid], n2);
            pi[tid] = 1.0 / sqrt(n2);   // (pi is free: root / iroot hold what the outputs need)
        }
        __syncthreads();
        for (int idx = tid; idx < 4096; idx += NT) V[(idx >> 6) * LD + (idx & 63)] *= pi[idx & 63];
        __syncthreads();
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
            atomicAdd(p.stamps + 7, t_sweeps - t_kernel);
        }
        if (tid == (FORM == 0 ? 64 : 0)) atomicAdd(p.stamps + 2, acc_v);
    }

    EIG_PHASE("sweeps, back to the canonical matrices");
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
    EIG_PHASE("right / left / basis out");
    if (p.eig_right_frag) {   // ... and in prune_factored's A-operand order (kernels.h FactoredParams): coalesced loads for every tile that uses the system
        double* __restrict__ Rf = p.eig_right_frag + static_cast<size_t>(job.eig) * 4096;
        double* __restrict__ Lf = p.eig_left_frag + static_cast<size_t>(job.eig) * 4096;
        for (int idx = tid; idx < 4096; idx += NT) {
            const int lane = idx & 63, r = 16 * (idx >> 10) + (lane & 15), c = 4 * ((idx >> 6) & 15) + (lane >> 4);
            Rf[idx] = iroot[r] * V[r * LD + c];
            Lf[idx] = V[c * LD + r] * root[c];
        }
    }
    EIG_PHASE("fragment-ordered copies out");
    if (tid < 64) lam[tid] = A[tid * LD + tid];
    // equilibrium: left eigenvector of the eigenvalue closest to zero, scaled to sum 1 (index order)
    // (z = the FIRST index of the smallest |lambda|, as a scan from index 0 finds it -- one lane per eigenvalue and a wave reduction; the
    // mass is the sum of the 64 products in index order, the products made by 64 lanes.  As two loops of one thread: 10,000 cycles.)
    if (tid < 64) {
        double key = fabs(A[tid * LD + tid]);
        key = key == key ? key : __builtin_huge_val();   // a NaN never wins a scan's comparison ...
        int idx = tid;
        for (int sft = 32; sft > 0; sft >>= 1) {
            const double ok = __shfl_xor(key, sft);
            const int oi = __shfl_xor(idx, sft);
            const bool take = ok < key || (ok == key && oi < idx);
            key = take ? ok : key;
            idx = take ? oi : idx;
        }
        const double a0 = A[0];
        const int z = a0 == a0 ? idx : 0;                // ... and one at index 0 never loses it
        pi[tid] = V[tid * LD + z] * root[tid];           // (pi is free: root / iroot hold what the outputs need)
        if (tid == 0) pp[0] = z;
    }
    __syncthreads();
    if (tid == 0) {
        double mass = 0.0;
#pragma unroll 16
        for (int j = 0; j < 64; ++j) mass += pi[j];
        norm_s = mass;
    }
    __syncthreads();
    EIG_PHASE("smallest eigenvalue, its vector's mass");
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
    EIG_PHASE("prior, weights");
}

hipError_t launch_eigen_omega(const EigenParams& p, hipStream_t stream)
{
    if (p.n_jobs == 0) return hipSuccess;
    constexpr size_t lds = 2 * 64 * LD * sizeof(double);
    static std::atomic<uint64_t> attr_set{0};  // per device (kernels.hip launch_impl)
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    if (!(attr_set.load(std::memory_order_acquire) >> (dev & 63) & 1)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&eigen_omega_kernel<0>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&eigen_omega_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(&eigen_omega_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
        if (e != hipSuccess) return e;
        attr_set.fetch_or(uint64_t(1) << (dev & 63), std::memory_order_release);
    }
    // Three forms of the same iteration (one form per process setting, whatever a launch's size: a block's result must not depend on the
    // list it is scored in).  Default since round 5: FORM 2, seven waves -- 0.31 ms per cold launch of 200 jobs against FORM 1's 0.56 ms,
    // 1.9 against 2.3 ms at 2,048 jobs, OMEGA end to end 0.140 / 0.223 / 0.703 s against 0.179 / 0.267 / 0.797 s at 516 / 2,064 / 8,256
    // blocks (profiles/r05/eigen_forms).  PCSF_EIGEN_FORM=1: rounds 2-4's form (128 threads: wave 0 holds B, wave 1 V);
    // PCSF_EIGEN_FORM=0 or PCSF_EIGEN_LDS=1: the first version (matrices in LDS, 256 threads).  Both kept for A/B runs and tests.
    const char* lds_env = getenv("PCSF_EIGEN_LDS");   // read at every launch: tests switch forms inside one process
    const bool in_lds = lds_env && lds_env[0] == '1';
    const char* form_env = getenv("PCSF_EIGEN_FORM");
    const int form = in_lds ? 0 : (form_env && form_env[0] >= '0' && form_env[0] <= '2') ? form_env[0] - '0' : 2;
    if (form == 0) hipLaunchKernelGGL(eigen_omega_kernel<0>, dim3(p.n_jobs), dim3(256), lds, stream, p);
    else if (form == 1) hipLaunchKernelGGL(eigen_omega_kernel<1>, dim3(p.n_jobs), dim3(128), lds, stream, p);
    else hipLaunchKernelGGL(eigen_omega_kernel<2>, dim3(p.n_jobs), dim3(448), lds, stream, p);
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


// tests: FORM 2's bookkeeping replayed on indices -- the 63 x 32 pairs of one sweep in step order as the kernel's seats give them (inside
// steps: seat e of a group with seat e ^ X; cross steps: seat k with seat 4 + ((k + t) & 3)), and whether every round's end (the staged
// look-up of B: rr::shift_source as smapP / smapQ use it; the register moves and the four-column exchange of V) lands where rq::canon says:
// returns the number of disagreements (0), over two sweeps.
int eigen_schedule_quads_for_tests(int* pairs /* [63][32][2] */)
{
    int arr[64], vv[2][32], bad = 0, step = 0;
    for (int s = 0; s < 64; ++s) arr[s] = vv[s >> 5][s & 31] = rq::canon(0, s);
    auto emit = [&](int a, int b, bool record, int slot) {
        if (vv[a >> 5][a & 31] != arr[a] || vv[b >> 5][b & 31] != arr[b]) ++bad;
        if (record) {
            pairs[(step * 32 + slot) * 2 + 0] = arr[a];
            pairs[(step * 32 + slot) * 2 + 1] = arr[b];
        }
    };
    for (int sweep = 0; sweep < 2; ++sweep)
        for (int r = 0; r < 15; ++r) {
            if (r == 0)
                for (int X = 1; X <= 3; ++X) {
                    int slot = 0;
                    for (int G = 0; G < 8; ++G)
                        for (int q = 0; q < 8; ++q)
                            if (q < (q ^ X)) emit(8 * G + q, 8 * G + (q ^ X), sweep == 0, slot++);
                    if (sweep == 0) ++step;
                }
            for (int t = 0; t < 4; ++t) {
                for (int G = 0; G < 8; ++G)
                    for (int k = 0; k < 4; ++k) emit(8 * G + k, 8 * G + 4 + ((k + t) & 3), sweep == 0, 4 * G + k);
                if (sweep == 0) ++step;
            }
            // the round's end
            int next[64];
            for (int s = 0; s < 64; ++s) {
                const int src = rr::shift_source(s >> 3, (s >> 2) & 1);
                next[s] = arr[8 * (src & 7) + 4 * (src >> 3) + (s & 3)];
            }
            for (int s = 0; s < 64; ++s) arr[s] = next[s];
            int out0[4], out1[4];
            for (int q = 0; q < 4; ++q) {
                out0[q] = vv[0][rq::leaving_group<0>() + q];
                out1[q] = vv[1][rq::leaving_group<1>() + q];
            }
            rq::shift_groups<0>(vv[0], out1);
            rq::shift_groups<1>(vv[1], out0);
            const int r1 = r == 14 ? 0 : r + 1;
            for (int s = 0; s < 64; ++s)
                if (arr[s] != rq::canon(r1, s) || vv[s >> 5][s & 31] != arr[s]) ++bad;
        }
    return bad;
}

}  // namespace pcsf
