// tiles.hip -- block-mode pruning in eigen coordinates, ONE WAVE PER TILE (MLE / OMEGA evaluations of whole blocks).
//
// One objective evaluation of the whole-block strategies (reference: lpr_leaves under a freshly instantiated model,
// src/fixed_lik.hpp:362-449 + src/instance.hpp:449-646) applies P(t_b) = R diag(e^{lambda t_b}) L of every branch to the block's
// columns.  The factors are applied directly (factored.hip has the algebra and what is not reproduced entry by entry):
//     internal edge   m = R (e_b o (L v))          two 64x64 x (64x16) products
//     leaf edge       m = R (e_b o L[:, codon])    one product; nothing when the leaf is a gap in all 16 columns
//     all-gap subtree m = 1                        nothing
//     edge whose sibling is an all-gap subtree     y <- e_b o y, no product (L R = I)
// What is different here is the mapping.  Every product of an evaluation multiplies by one of only TWO matrices, R and L, whatever
// the branch, and for the reversible models of this path the two are one: the eigen-systems come from the symmetrised rate matrix
// (R = D^-1/2 U, L = U^T D^1/2 with U orthogonal, D = diag(pi): hostmath.hpp eigen_reversible, eigen.hip), so L = R^T diag(w) with
// w = root / iroot (= pi to rounding).  A work-group stages R ONCE -- 32 KiB of LDS, row-major with an XOR swizzle of the columns that
// makes both the A operands of R x (lane (s, g) reads R[16t + s][4q + g]) and those of R^T x (R[4q + g][16t + s]) conflict-free
// ds_read_b64 -- and from then on its waves never synchronise again: each wave owns a 16-column tile, keeps the tile's message in
// the B-operand layout of v_mfma_f64_16x16x4_f64 (16 registers per lane), and a product is 64 MFMAs.  The result registers are the
// B operands of the next product (kernels.hip, same layout), so a tile stays in registers from the leaves to the root; no matrix is
// formed, no table is written to or read from HBM, no barrier or LDS-DMA sits between products (prune_fixed in block mode re-stages
// a 32 KiB branch matrix per product; prune_factored exchanges the tile through LDS with a barrier per product and pays its prologue
// -- 64 KiB of eigenvectors into registers -- per 16 columns).  ~40 KiB of LDS per work-group: four work-groups per CU, so a group
// whose evaluation has fewer tiles than waves (an OMEGA kappa probe has an eigen-system of its own and only its block's tiles) leaves
// no SIMD idle -- waves without a tile exit at once and free their registers.  A work-group takes up to TILE_JOBS_PER_GROUP tiles that
// share an eigen-system: the tiles of one evaluation of a block, of several evaluations of the same block (rho stages), or of any
// blocks under the two ECM models (MLE).
//
// Leaf columns of L are rows of R (16 x ds_read_b64 per lane and leaf, scaled by w[codon]); edge scales e_b come from
// edge_scale_kernel (factored.hip), 128 bytes per lane group and branch from L2.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>

#include "kernels.h"

namespace pcsf {

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

namespace {

struct V16 {
    double v[16];
};

// R is kept row-major with its columns XOR-swizzled: element (i, j) at i * 64 + (j ^ swz(i)), swz(i) = i & 15.  The compiler reads the
// four tiles of a k-step as two ds_read2_b64, which the LDS serves in groups of 16 consecutive lanes (s = 0..15 at one g): for R x
// they hit 8-byte slots ((4q + g) ^ s) mod 16 -- sixteen different ones --, for R^T x (s ^ g ^ 4 (q & 3)) mod 16: sixteen again.
// (The first swizzle, ((i & 1) << 4) | (i & 14), was derived for 32-lane ds_read_b64 passes and cost a 2-way conflict on every
// fragment read of R x: SQ_LDS_BANK_CONFLICT 5.8 cycles per LDS instruction.)
__device__ __forceinline__ int swz(int i) { return i & 15; }

// wave-uniform values the compiler cannot prove uniform (they pass through structs and loop-carried state): forcing them into
// scalar registers turns the op loop's branches into scalar branches instead of EXEC-masked regions that evaluate both sides
__device__ __forceinline__ uint32_t uni(uint32_t x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ bool uni(bool x) { return __builtin_amdgcn_readfirstlane(static_cast<int>(x)) != 0; }
// max(x, 0) in one instruction (fmax() first canonicalises its operand: a second v_max_f64 per value)
__device__ __forceinline__ double clamp0(double x)
{
    double r;
    asm("v_max_f64 %0, %1, 0" : "=v"(r) : "v"(x));
    return r;
}

#ifdef PCSF_TILES_STAMPS   // diagnostic build only (tools/tile_stamps.sh): cycles inside products vs the whole job
__device__ unsigned long long g_tile_stamps[8];
__device__ __forceinline__ unsigned long long tile_clock()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#define PCSF_TILE_T0() const unsigned long long t_in_ = tile_clock();
#define PCSF_TILE_T1(o) { asm volatile("" ::"v"((o).v[0]), "v"((o).v[15])); t_prod += tile_clock() - t_in_; ++n_prod; }
#define PCSF_TILE_SEG0() const unsigned long long t_seg_ = tile_clock();
#define PCSF_TILE_SEG1(acc, o) { asm volatile("" ::"v"((o).v[0]), "v"((o).v[15])); acc += tile_clock() - t_seg_; }
#else
#define PCSF_TILE_T0()
#define PCSF_TILE_T1(o)
#define PCSF_TILE_SEG0()
#define PCSF_TILE_SEG1(acc, o)
#endif

// o <- R x b for the 64x16 tile b (B-operand layout: b[q] of lane l = row 4q + (l >> 4), column l & 15): 64 MFMAs in four accumulator
// chains (output tile t = states 16t..16t+15), the four ds_read_b64 of the next k-step issued behind the first MFMA of this one.
// o comes out in the same layout: o[4t + r] = acc[t][r].
__device__ __forceinline__ void product_R(const double* __restrict__ Rs, int s, int g, const V16& b, V16& o)
{
    double4_t acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = double4_t{0.0, 0.0, 0.0, 0.0};
    // lane (s, g), tile t, k-step q reads R[16t + s][4q + g]: column (4q + g) ^ swz(s) = 4 (q ^ ms) + (g ^ (s & 3)), ms = s >> 2.  ms only
    // touches the low two bits of q: four per-lane addresses (q & 3 = 0..3) and compile-time offsets for the rest -- no address
    // arithmetic between the MFMAs (integer VALU instructions take the matrix pipe's time like any other)
    const int ms = (s >> 2) & 3;
    const double* __restrict__ base = Rs + s * 64 + (g ^ (s & 3));
    const double* __restrict__ bj[4] = {base + 4 * (0 ^ ms), base + 4 * (1 ^ ms), base + 4 * (2 ^ ms), base + 4 * (3 ^ ms)};
    double n0 = bj[0][0], n1 = bj[0][1024], n2 = bj[0][2048], n3 = bj[0][3072];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const double f0 = n0, f1 = n1, f2 = n2, f3 = n3;
        __builtin_amdgcn_sched_barrier(0);
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(f0, b.v[q], acc[0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#ifndef PCSF_TILES_NO_FRAGMENT_READS
        if (q + 1 < 16) {
            const double* __restrict__ nx = bj[(q + 1) & 3] + 16 * ((q + 1) >> 2);
            n0 = nx[0];
            n1 = nx[1024];
            n2 = nx[2048];
            n3 = nx[3072];
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f1, b.v[q], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(f2, b.v[q], acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(f3, b.v[q], acc[3], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) o.v[4 * t + r] = acc[t][r];
}

// o <- R^T x b: lane (s, g), tile t, k-step q reads R[4q + g][16t + s], stored at column (16t + s) ^ swz(4q + g)
// = 16 t + (s ^ g ^ 4 (q & 3))
__device__ __forceinline__ void product_Rt(const double* __restrict__ Rs, int s, int g, const V16& b, V16& o)
{
    double4_t acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = double4_t{0.0, 0.0, 0.0, 0.0};
    const int x = s ^ g;
    const double* __restrict__ row0 = Rs + g * 64;   // tile t at + 16 t
    const double* __restrict__ bj[4] = {row0 + (x ^ 0), row0 + (x ^ 4), row0 + (x ^ 8), row0 + (x ^ 12)};   // by q & 3; the rest are constants
    double n0 = bj[0][0], n1 = bj[0][16], n2 = bj[0][32], n3 = bj[0][48];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const double f0 = n0, f1 = n1, f2 = n2, f3 = n3;
        __builtin_amdgcn_sched_barrier(0);
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(f0, b.v[q], acc[0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#ifndef PCSF_TILES_NO_FRAGMENT_READS
        if (q + 1 < 16) {
            const double* __restrict__ nx = bj[(q + 1) & 3] + (q + 1) * 256;
            n0 = nx[0];
            n1 = nx[16];
            n2 = nx[32];
            n3 = nx[48];
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f1, b.v[q], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(f2, b.v[q], acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(f3, b.v[q], acc[3], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) o.v[4 * t + r] = acc[t][r];
}

}  // namespace

constexpr int TILE_WAVES = 4;

// OWN: the kernel makes its edge scales itself and validates the evaluation (p.eval_x); otherwise it reads edge_scale_kernel's rows
template <bool ANC, bool OWN>
__global__ __launch_bounds__(TILE_WAVES * 64, 2) void prune_tiles(const FactoredParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_tiles[];
    double* Rs = reinterpret_cast<double*>(lds_tiles);                 // [64][64] R, columns swizzled
    double* wL = Rs + 4096;                                            // [4][16] w = root / iroot (L = R^T diag(w)) in lane order: [g][q] = w[4q + g]
    uint32_t* lprog = reinterpret_cast<uint32_t*>(wL + 64);            // [n_ops rounded up to 64]
    int32_t* lslot = reinterpret_cast<int32_t*>(lprog + ((p.n_ops + 63) & ~63u));   // [n_edges] slot -> branch
    // per wave: [n_leaves][16] codons (+ [n_leaves] "has a codon" for trees of more than 128 leaves)
    const int cod_bytes = (p.n_leaves * (p.n_leaves > 128 ? 17 : 16) + 15) & ~15;
    unsigned char* lcod_all = reinterpret_cast<unsigned char*>(lslot + ((p.n_edges + 3) & ~3u));
    // in-kernel edge scales (p.eval_x): the branch lengths and, per wave, 64 doubles through which the lanes hand their exponentials round
    double* lsc_all = reinterpret_cast<double*>(lcod_all + TILE_WAVES * cod_bytes);
    float* lblen = reinterpret_cast<float*>(lsc_all + TILE_WAVES * 64);
    constexpr bool own_scales = OWN;

    TileGroup grp = p.groups[blockIdx.x];
    if (grp.job0 + grp.n_jobs > p.n_jobs) grp.n_jobs = grp.job0 < p.n_jobs ? p.n_jobs - grp.job0 : 0u;   // (a list of fixed groups of four: the last one of a launch may be short)
    if (grp.n_jobs == 0) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_threads = blockDim.x, n_waves = n_threads >> 6;   // a work-group has as many waves as its group has jobs (at most TILE_WAVES)
    const int sj = lane & 15, g = lane >> 4;
    const uint32_t eig = p.jobs[grp.job0].eig;

    // ---- R of the group's eigen-system into LDS: thread t takes row i = t >> 2, columns 16 (t & 3) .. +15 (128 contiguous bytes);
    // a pair of columns (2u, 2u + 1) stays a pair under the swizzle -- swapped when swz(i) is odd: eight 16-byte stores
    for (int vt = tid; vt < 256; vt += n_threads) {
        const int i = vt >> 2, cc = vt & 3;
        const double2_t* __restrict__ Rg = reinterpret_cast<const double2_t*>(p.eig_right + static_cast<size_t>(eig) * 4096 + i * 64 + 16 * cc);
        double2_t r[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) r[u] = Rg[u];
        const int sw = swz(i);
#pragma unroll
        for (int u = 0; u < 8; ++u)
            *reinterpret_cast<double2_t*>(Rs + i * 64 + (((16 * cc + 2 * u) ^ sw) & ~1)) = (sw & 1) ? double2_t{r[u].y, r[u].x} : r[u];
    }
    if (tid < 64) wL[tid] = p.eig_w[static_cast<size_t>(eig) * 64 + tid];
    for (int i = tid; i < static_cast<int>((p.n_ops + 63) & ~63u); i += n_threads) lprog[i] = p.program[i < p.n_ops ? i : p.n_ops - 1];
    for (int i = tid; i < p.n_edges; i += n_threads) lslot[i] = p.slot_branch[i];
    if (own_scales)
        for (uint32_t i = tid; i < p.n_branches; i += n_threads) lblen[i] = p.branch_len[i];
    __syncthreads();   // the only barrier: from here on the waves are independent

    unsigned char* lcod = lcod_all + wave * cod_bytes;
    unsigned char* lany = lcod + p.n_leaves * 16;
    const int n_ops = p.n_ops;

    for (uint32_t jx = wave; jx < grp.n_jobs; jx += n_waves) {
        const FactoredJob job = p.jobs[grp.job0 + jx];
#ifdef PCSF_TILES_STAMPS
        const unsigned long long t_job = tile_clock();
        unsigned long long t_prod = 0, n_prod = 0, t_leaf = 0, t_setup = 0;
#endif
        uint32_t products = 0;   // wave-uniform: the products this tile runs (p.product_count)
        // the tile's codons (16 consecutive lanes hold one leaf's columns)
        for (int idx = lane; idx < p.n_leaves * 16; idx += 64) {
            const int leaf = idx >> 4, s = idx & 15;
            unsigned c = 64u;
            if (static_cast<uint32_t>(s) < job.n_sites) {
                c = p.codons[static_cast<size_t>(leaf) * p.leaf_stride + job.site0 + s];
                c = c > 64u ? 64u : c;
            }
            lcod[idx] = static_cast<unsigned char>(c);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // per leaf: does it have a codon in the tile?  leaves 0..127 as two wave-uniform bit masks
        auto row_has = [&](int leaf) __attribute__((always_inline)) -> bool {
            const uint4 w = *reinterpret_cast<const uint4*>(lcod + leaf * 16);
            return (w.x != 0x40404040u) | (w.y != 0x40404040u) | (w.z != 0x40404040u) | (w.w != 0x40404040u);
        };
        const unsigned long long has0 = __ballot(lane < p.n_leaves && row_has(lane < p.n_leaves ? lane : 0));
        const unsigned long long has1 = __ballot(64 + lane < p.n_leaves && row_has(64 + lane < p.n_leaves ? 64 + lane : 0));
        if (p.n_leaves > 128) {
            for (int l = lane; l < p.n_leaves; l += 64) lany[l] = row_has(l);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        auto leaf_has = [&](int a) __attribute__((always_inline)) -> bool {
            if (a < 64) return (has0 >> a) & 1ull;
            if (a < 128) return (has1 >> (a - 64)) & 1ull;
            return lany[a] != 0;
        };
        const double* __restrict__ scale_eval = own_scales ? nullptr : p.edge_scale + static_cast<size_t>(job.eval) * p.n_branches * 64;
        const double* __restrict__ scale = scale_eval + 4 * g;
        // Edge scales made here (p.eval_x): e_b[k] = exp(lambda_k t_b), t_b = (double)(float)(len_b x) like the reference's float tree
        // (src/instance.hpp:299-307,498) -- lane k computes the exponential of eigenvalue k (the same exp() on the same operand as
        // edge_scale_kernel: the same bits) and the lanes hand them round through 512 bytes of LDS, laid out so that lane (s, g) reads its
        // sixteen values 4q + g as four ds_read_b128 of one address per g (broadcast).  Replaces a kernel per round and set, 29-51 KB written
        // and read per evaluation, and the ~1,150 cycles a leaf waited for its row of scales.
        double* lsc = lsc_all + wave * 64;
        const double lam = own_scales ? p.eig_lambda[static_cast<size_t>(job.eig) * 64 + lane] : 0.0;
        const double x_eval = own_scales ? p.eval_x[job.eval] : 0.0;
        if (own_scales && (job.pad & FACTORED_JOB_VALIDATES)) {
            // The reference rejects a model whose raw row sums of P(t) are off by more than 1e-6 (src/instance.hpp:632-636): sum_j P_ij(t) =
            // sum_k R_ik e^{lambda_k t} c_k with c = L 1 (edge_scale_kernel has the argument why one branch, the longest, is enough).
            // Row i = lane, R from the work-group's LDS image.
            const double t_long = static_cast<double>(static_cast<float>(static_cast<double>(p.longest_branch) * x_eval));
            lsc[lane] = exp(lam * t_long) * p.eig_c[static_cast<size_t>(job.eig) * 64 + lane];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            double sum = 0.0;
            const double* __restrict__ row = Rs + lane * 64;
            const int sw = swz(lane);
            for (int k = 0; k < 64; ++k) sum += row[k ^ sw] * lsc[k];
            if (!(fabs(sum - 1.0) <= 1e-6)) p.fail[job.eval] = 1u;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }

        // e_b in this lane's rows: e[q] = exp(lambda_k t_b), k = 4q + g; row-split position of k is 16(q >> 2) + 4g + (q & 3)
        // own scales in two halves, so that the sixteen values are not live across a product: scale_begin(branch) computes this lane's
        // exponential and stores it, scale_read() fetches the lane's sixteen (no other scale_begin may come between the two)
        auto scale_begin = [&](int branch) __attribute__((always_inline)) {
            const double t = static_cast<double>(static_cast<float>(static_cast<double>(lblen[branch]) * x_eval));
            lsc[(lane & 3) * 16 + (lane >> 2)] = exp(lam * t);   // state k = lane at [g = k & 3][q = k >> 2]
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        };
        auto scale_read = [&]() __attribute__((always_inline)) -> V16 {
            V16 e;
            const double4_t* s4 = reinterpret_cast<const double4_t*>(lsc + g * 16);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const double4_t v = s4[i];
                e.v[4 * i] = v.x;
                e.v[4 * i + 1] = v.y;
                e.v[4 * i + 2] = v.z;
                e.v[4 * i + 3] = v.w;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the next scale_begin's store comes behind these reads
            __builtin_amdgcn_wave_barrier();
            return e;
        };
        auto load_scale = [&](int branch) __attribute__((always_inline)) -> V16 {
            V16 e;
            if (own_scales) {
                scale_begin(branch);
                return scale_read();
            }
            const double2_t* s2 = reinterpret_cast<const double2_t*>(scale + static_cast<size_t>(branch) * 64);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const double2_t lo = s2[8 * i], hi = s2[8 * i + 1];
                e.v[4 * i] = lo.x;
                e.v[4 * i + 1] = lo.y;
                e.v[4 * i + 2] = hi.x;
                e.v[4 * i + 3] = hi.y;
            }
            return e;
        };

        // A message is kept in one of two forms (factored.hip): real -- the 64x16 tile itself -- or eigen: y with message = max(R y, 0)
        // still to be formed.  `mask`: columns in which the whole subtree is gaps (their message is exactly 1); 0xffff = the all-ones
        // tile, whose registers hold nothing.  Two working messages, X (the walk's current one) and Y (the other operand of a
        // product of two children); every step works in place, so the kernel has three product sites in all.
        struct Msg {
            V16 v;
            uint32_t mask;
            bool eig;
        };
        Msg X, Y;
        X.mask = Y.mask = 0xffffu;
        X.eig = Y.eig = false;
        // leaf a, eigen form: L[:, codon] o e_a, column `codon` of L being row `codon` of R times w[codon]:
        // R[c][4q + g] sits at column (4q + g) ^ swz(c) = 4 (q ^ mc) + (g ^ (c & 3)).  A gap column's entries never matter: its message
        // is forced to exactly 1 when the real form is made.
        auto leaf = [&](int a) __attribute__((always_inline)) -> Msg {
            Msg T;
            T.mask = 0xffffu;
            T.eig = false;
            if (!leaf_has(a)) return T;
            const unsigned code = lcod[a * 16 + sj];
            const uint32_t mask = uni(static_cast<uint32_t>(__ballot(code == 64u)) & 0xffffu);
            if (mask == 0xffffu) return T;
            PCSF_TILE_SEG0()
            const V16 e = load_scale(a);
            const unsigned col = code == 64u ? 0u : code;
            const int mc = (col >> 2) & 3;
            const double* __restrict__ base = Rs + col * 64 + (g ^ (col & 3));
            const double wc = wL[(col & 3) * 16 + (col >> 2)];
#pragma unroll
            for (int q = 0; q < 16; ++q) T.v.v[q] = base[4 * (q ^ mc)] * wc * e.v[q];
            PCSF_TILE_SEG1(t_leaf, T.v)
            T.mask = mask;
            T.eig = true;
            return T;
        };
        // T (not the all-ones tile) in real form: max(R y, 0), gap columns exactly 1
        auto make_real = [&](const Msg& T) __attribute__((always_inline)) -> Msg {
            if (!uni(T.eig)) return T;
            Msg out;
            PCSF_TILE_T0()
            product_R(Rs, sj, g, T.v, out.v);
            ++products;
            PCSF_TILE_T1(out.v)
            const uint32_t tmask = uni(T.mask);
            if (tmask == 0u) {
#pragma unroll
                for (int q = 0; q < 16; ++q) out.v.v[q] = clamp0(out.v.v[q]);
            } else {
                const bool gap_col = (tmask >> sj) & 1u;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const double c = clamp0(out.v.v[q]);   // unconditionally: inside the select the compiler would branch around the asm
                    out.v.v[q] = gap_col ? 1.0 : c;
                }
            }
            out.mask = tmask;
            out.eig = false;
            return out;
        };
        // the edge of program slot slot_index applied to T: y <- e_b o (L x), or e_b o y when T is in eigen form already (L R = I)
        auto edge = [&](int slot_index, const Msg& T) __attribute__((always_inline)) -> Msg {
            if (uni(T.mask) == 0xffffu) return T;
            V16 e;
            if (own_scales) scale_begin(lslot[slot_index]);   // the exponential now, its sixteen values after the product
            else e = load_scale(lslot[slot_index]);
            Msg out;
            if (!uni(T.eig)) {   // L x = R^T (w o x)
                const double* __restrict__ wl = wL + g * 16;
                V16 wv;
#pragma unroll
                for (int q = 0; q < 16; ++q) wv.v[q] = T.v.v[q] * wl[q];
                PCSF_TILE_T0()
                product_Rt(Rs, sj, g, wv, out.v);
                ++products;
                PCSF_TILE_T1(out.v)
            } else {
                out.v = T.v;
            }
            if (own_scales) e = scale_read();
#pragma unroll
            for (int q = 0; q < 16; ++q) out.v.v[q] *= e.v[q];
            out.mask = T.mask;
            out.eig = true;
            return out;
        };
        // element-wise product in real form; an all-ones operand costs nothing and leaves the other in whatever form it has
        auto mul = [&](const Msg& A, const Msg& B) __attribute__((always_inline)) -> Msg {
            if (uni(B.mask) == 0xffffu) return A;
            if (uni(A.mask) == 0xffffu) return B;
            const Msg a = make_real(A);
            const Msg b = make_real(B);
            Msg out;
#pragma unroll
            for (int q = 0; q < 16; ++q) out.v.v[q] = a.v.v[q] * b.v.v[q];
            out.mask = uni(a.mask & b.mask);
            out.eig = false;
            return out;
        };

        // parked messages: pushes and pops nest; an all-ones tile is never stored or loaded
        // Depths 0 and 1 in registers; deeper ones -- one or two pushes per tile in the reference's trees, any depth newick_parse
        // accepts otherwise -- in a dynamically indexed array, i.e. scratch memory: a third register set would push the products'
        // operands out of the 256 registers a wave has at two waves per SIMD.
        V16 s0, s1, stk[MAX_STACK_DEPTH_DEEP_DEV];
        uint32_t stk_mask[MAX_STACK_DEPTH_DEEP_DEV + 2];
        uint32_t stk_eig = 0;                  // bit d: parked message d is in eigen form
        auto park = [&](int d, const Msg& T) __attribute__((always_inline)) {
            stk_mask[d] = T.mask;
            stk_eig = (stk_eig & ~(1u << d)) | (T.eig ? 1u << d : 0u);
            if (uni(T.mask) == 0xffffu) return;
            if (d == 0) s0 = T.v;
            else if (d == 1) s1 = T.v;
            else stk[d - 2] = T.v;
        };
        auto unpark = [&](int d) __attribute__((always_inline)) -> Msg {
            Msg T;
            T.mask = uni(stk_mask[d]);
            T.eig = uni(((stk_eig >> d) & 1u) != 0);
            if (T.mask == 0xffffu) return T;
            if (d == 0) T.v = s0;
            else if (d == 1) T.v = s1;
            else T.v = stk[d - 2];
            return T;
        };
        // one op: `slot` and `depth` are the program slot and the number of parked tiles the op starts with; parked_ones: the tile a
        // pop takes back is known to be all ones whatever the registers hold (its parking op was skipped)
        auto run_op = [&](uint32_t op, int slot, int depth, bool parked_ones) __attribute__((always_inline)) {
            const uint32_t code = op >> 28;
            const int a = (op >> 14) & 0x3fff, b = op & 0x3fff;
            // Edge scales are read once per evaluation and branch: every load of them misses L2 (2,048 evaluations in flight x 29 KB),
            // and a leaf needs its scales at once -- ~2,500 cycles of memory latency per leaf, a quarter of a lone wave's time.  One
            // wave-wide load touches the four 128-byte lines of this op's leaves and of the next two edges while the op's first
            // product runs; the value is consumed at the end of the op (one register), so the demand loads behind it hit.
            float touched = 0.0f;
            if (!own_scales && code < F_ROOT_DEV) {
                const int which = (lane >> 2) & 3;
                const int nxt = slot + (which - 1) < p.n_edges ? slot + (which - 1) : p.n_edges - 1;   // which = 2, 3: the next two edges
                const int br = which == 0 ? a : (which == 1 ? b : lslot[nxt > 0 ? nxt : 0]);
                touched = *reinterpret_cast<const float*>(scale_eval + static_cast<size_t>(br) * 64 + (lane & 3) * 16);
            }
            if (code >= F_ROOT_DEV) {   // z = prior . alpha_root, ln z; the ancestral term sum_x ln(prior_x) prior_x alpha_x / z
                if (uni(X.mask) == 0xffffu) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) X.v.v[q] = 1.0;
                } else {
                    X = make_real(X);
                }
                const double* __restrict__ pr = p.prior + static_cast<size_t>(job.eig) * 64 + g * 16;   // lane order [g][q]
                double z = 0.0, t = 0.0;
#pragma unroll
                for (int q = 0; q < 16; ++q) z += pr[q] * X.v.v[q];
                z += __shfl_xor(z, 16);
                z += __shfl_xor(z, 32);
                if (ANC) {
                    const double* __restrict__ aw = p.anc_weight + static_cast<size_t>(job.eig) * 64 + g * 16;
#pragma unroll
                    for (int q = 0; q < 16; ++q) t += aw[q] * X.v.v[q];
                    t += __shfl_xor(t, 16);
                    t += __shfl_xor(t, 32);
                }
                if (g == 0 && static_cast<uint32_t>(sj) < job.n_sites) {
                    p.out_lnz[job.out0 + sj] = log(z);
                    if (ANC) p.out_anc[job.out0 + sj] = z == 0.0 ? 0.0 : t / z;
                }
                return;
            }
            if (code != F_CHERRY_DEV) X = edge(slot, X);   // every op but the first cherry starts with the edge above the walk's current node
            if (code >= F_MM_POP0_DEV) {                    // pop: the parked sibling
                if (parked_ones) return;
                Y = unpark(depth - 1);
            } else {
                if (code >= F_MM_PUSH_CHERRY0_DEV) park(depth, X);
                if (code != F_MM_LEAF_DEV) X = leaf(a);     // a cherry: two leaves
                Y = leaf(code == F_MM_LEAF_DEV ? a : b);
            }
            X = mul(X, Y);
            asm volatile("" ::"v"(touched));
        };

        // The ops of all-gap subtrees are skipped (factored.hip): schedule.hpp `program_facts` knows every op's leaf set, start slot,
        // stack depth and, for a pop, the parking op; with the tile's leaf bits a lane-parallel pass marks the ops whose result is the
        // all-ones tile, and the loop visits only the others (trees of at most 128 leaves: n_ops <= 128).  Larger trees run every op,
        // slot and depth counted along the way.
        const bool skipping = p.op_meta != nullptr;
        uint32_t opw[2] = {0, 0}, mt[2] = {0, 0};
        unsigned long long ones_after[2] = {0, 0}, active[2] = {0, 0};
        if (skipping) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int j = 64 * k + lane;
                const bool valid = j < n_ops;
                opw[k] = valid ? lprog[j] : (static_cast<uint32_t>(F_ROOT_DEV) << 28);
                mt[k] = valid ? p.op_meta[j] : 0u;
                const unsigned long long l0 = valid ? p.op_leaves[2 * j] : 0ull, l1 = valid ? p.op_leaves[2 * j + 1] : 0ull;
                ones_after[k] = __ballot(valid && ((l0 & has0) | (l1 & has1)) == 0ull);
            }
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int j = 64 * k + lane;
                const uint32_t code = opw[k] >> 28;
                const bool is_push = code >= F_MM_PUSH_CHERRY0_DEV && code < F_MM_POP0_DEV, is_root = code >= F_ROOT_DEV;
                const bool after = (ones_after[k] >> lane) & 1ull;
                const bool before = lane > 0 ? ((ones_after[k] >> (lane - 1)) & 1ull) : (k == 1 && (ones_after[0] >> 63));
                const bool trivial = j < n_ops && after && !is_root && (!is_push || before);
                active[k] = __ballot(j < n_ops && !trivial);
            }
        }
        int k = 0, ip = 0, slot_ctr = 0, sp = 0;
        unsigned long long todo = active[0];
#ifdef PCSF_TILES_STAMPS
        t_setup = tile_clock() - t_job;
#endif
        for (;;) {
            uint32_t op;
            int slot, depth;
            bool parked_ones = false;
            if (skipping) {
                if (!todo) {
                    if (k == 1) break;
                    k = 1;
                    todo = active[1];
                    if (!todo) break;
                }
                const int l = __builtin_ctzll(todo);
                todo &= todo - 1;
                const int i = 64 * k + l;
                op = __builtin_amdgcn_readlane(k ? opw[1] : opw[0], l);
                const uint32_t meta = __builtin_amdgcn_readlane(k ? mt[1] : mt[0], l);
                const unsigned long long oa = k ? ones_after[1] : ones_after[0];
                const bool ones_before = i > 0 && ((l > 0 ? oa >> (l - 1) : ones_after[0] >> 63) & 1ull);
                if (ones_before) {   // whatever X holds is from before a run of skipped ops
                    X.mask = 0xffffu;
                    X.eig = false;
                }
                const int parking = static_cast<int>(meta >> 18);   // >= 1 for a pop
                parked_ones = parking > 0 && ((ones_after[(parking - 1) >> 6] >> ((parking - 1) & 63)) & 1ull);
                slot = static_cast<int>(meta & 0x3fffu);
                depth = static_cast<int>((meta >> 14) & 0xfu);
            } else {
                if (ip >= n_ops) break;
                op = __builtin_amdgcn_readfirstlane(lprog[ip]);
                ++ip;
                const uint32_t code = op >> 28;
                slot = slot_ctr;
                depth = sp;
                if (code != F_CHERRY_DEV && code < F_ROOT_DEV) ++slot_ctr;
                if (code >= F_MM_PUSH_CHERRY0_DEV && code < F_MM_POP0_DEV) ++sp;
                else if (code >= F_MM_POP0_DEV && code < F_ROOT_DEV) --sp;
            }
            run_op(op, slot, depth, parked_ones);
        }
#ifdef PCSF_TILES_STAMPS
        if (lane == 0) {
            atomicAdd(&g_tile_stamps[0], tile_clock() - t_job);
            atomicAdd(&g_tile_stamps[1], t_prod);
            atomicAdd(&g_tile_stamps[2], n_prod);
            atomicAdd(&g_tile_stamps[3], 1ull);
            atomicAdd(&g_tile_stamps[4], t_leaf);
            atomicAdd(&g_tile_stamps[5], t_setup);
        }
#endif
        if (p.product_count && lane == 0) atomicAdd(p.product_count, static_cast<unsigned long long>(products));
        __builtin_amdgcn_wave_barrier();   // the next job rewrites this wave's codon rows
    }
}

size_t prune_tiles_lds_bytes(const FactoredParams& p)
{
    return 4096 * sizeof(double) + 64 * sizeof(double) + ((static_cast<size_t>(p.n_ops) + 63) & ~size_t(63)) * 4 +
           ((static_cast<size_t>(p.n_edges) + 3) & ~size_t(3)) * 4 +
           TILE_WAVES * ((static_cast<size_t>(p.n_leaves) * (p.n_leaves > 128 ? 17 : 16) + 15) & ~size_t(15)) +
           (p.eval_x ? TILE_WAVES * 64 * sizeof(double) + ((static_cast<size_t>(p.n_branches) + 3) & ~size_t(3)) * sizeof(float) : 0);
}

// groups come sorted by their number of jobs; one launch per work-group size, so that no wave is launched without a tile: the
// registers of an idle wave would be held until its whole work-group can be placed
template <bool ANC, bool OWN>
static hipError_t launch_tiles_impl(const FactoredParams& p, size_t lds, hipStream_t stream)
{
    static std::atomic<uint64_t> attr_set{0};   // per (instantiation, device), like kernels.hip launch_impl
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    if (!(attr_set.load(std::memory_order_acquire) >> (dev & 63) & 1)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&prune_tiles<ANC, OWN>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set.fetch_or(uint64_t(1) << (dev & 63), std::memory_order_release);
    }
    // A launch that does not fill the chip a few times over is bound by the latency of one tile, and four launches in a row would
    // pay it four times (the reference's 516-block file: 2,218 launches of ~170 us instead of 850): one launch of full work-groups.
    if (p.n_groups < 2048) {
        hipLaunchKernelGGL((prune_tiles<ANC, OWN>), dim3(p.n_groups), dim3(TILE_WAVES * 64), lds, stream, p);
        return hipGetLastError();
    }
    // Largest work-groups first: a 4-wave work-group needs four free wave slots on one CU at once, and finds them while the chip is
    // empty; the small ones fill in behind (configs[2], OMEGA: 1.45 -> 1.38 s against smallest first; the launches on streams of
    // their own, so that none waits for another's last work-groups: 1.5 s).
    for (int waves = TILE_WAVES; waves >= 1; --waves) {
        uint32_t first = 0;
        for (int w = 1; w < waves; ++w) first += p.group_counts[w - 1];
        // the 4-wave launch takes every group that is left over (groups of more than TILE_WAVES jobs loop over their jobs)
        const uint32_t cnt = waves < TILE_WAVES ? p.group_counts[waves - 1] : p.n_groups - first;
        if (cnt == 0) continue;
        FactoredParams q = p;
        q.groups = p.groups + first;
        q.n_groups = cnt;
        hipLaunchKernelGGL((prune_tiles<ANC, OWN>), dim3(cnt), dim3(waves * 64), lds, stream, q);
        if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    }
    return hipSuccess;
}

template <bool ANC, bool OWN>
static hipError_t launch_tiles_sized_impl(const FactoredParams& p, int waves, size_t lds, hipStream_t stream)
{
    static std::atomic<uint64_t> attr_set{0};
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    if (!(attr_set.load(std::memory_order_acquire) >> (dev & 63) & 1)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&prune_tiles<ANC, OWN>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set.fetch_or(uint64_t(1) << (dev & 63), std::memory_order_release);
    }
    hipLaunchKernelGGL((prune_tiles<ANC, OWN>), dim3(p.n_groups), dim3(waves * 64), lds, stream, p);
    return hipGetLastError();
}

hipError_t launch_prune_tiles_sized(const FactoredParams& p, int waves, hipStream_t stream)
{
    if (p.n_groups == 0) return hipSuccess;
    if (!p.groups || waves < 1 || waves > TILE_WAVES) return hipErrorInvalidValue;
    const size_t lds = prune_tiles_lds_bytes(p);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (p.eval_x) return p.out_anc ? launch_tiles_sized_impl<true, true>(p, waves, lds, stream) : launch_tiles_sized_impl<false, true>(p, waves, lds, stream);
    return p.out_anc ? launch_tiles_sized_impl<true, false>(p, waves, lds, stream) : launch_tiles_sized_impl<false, false>(p, waves, lds, stream);
}

hipError_t launch_prune_tiles(const FactoredParams& p, hipStream_t stream)
{
#ifdef PCSF_TILES_STAMPS
    {
        static bool registered = false;
        if (!registered) {
            registered = true;
            atexit([] {
                unsigned long long h[8] = {0};
                if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_tile_stamps), sizeof h) == hipSuccess && h[3])
                    fprintf(stderr, "prune_tiles stamps: %llu jobs, %.0f cycles per job, %.1f products per job taking %.0f cycles each (%.3f of the job); leaf operands %.0f cycles per job, job set-up %.0f\n", h[3],
                            double(h[0]) / h[3], double(h[2]) / h[3], double(h[1]) / double(h[2] ? h[2] : 1), double(h[1]) / double(h[0] ? h[0] : 1), double(h[4]) / h[3], double(h[5]) / h[3]);
            });
        }
    }
#endif
    if (p.n_groups == 0) return hipSuccess;
    if (!p.groups) return hipErrorInvalidValue;
    const size_t lds = prune_tiles_lds_bytes(p);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (p.eval_x) return p.out_anc ? launch_tiles_impl<true, true>(p, lds, stream) : launch_tiles_impl<false, true>(p, lds, stream);
    return p.out_anc ? launch_tiles_impl<true, false>(p, lds, stream) : launch_tiles_impl<false, false>(p, lds, stream);
}

}  // namespace pcsf
