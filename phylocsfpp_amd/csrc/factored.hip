// factored.hip -- block-mode pruning WITHOUT explicit branch matrices (MLE / OMEGA evaluations of short blocks).
//
// One objective evaluation of the whole-block strategies (reference: lpr_leaves under a freshly instantiated model,
// src/fixed_lik.hpp:362-449 + src/instance.hpp:449-646) needs all n-1 matrices P(t_b) = R diag(e^{lambda t_b}) L only to
// apply them to the block's columns -- and the blocks score-msa sees are short (median 6 codons = ONE 16-column tile)
// and sparse (median 4 of 100 species present).  Forming P explicitly (expm.hip: four tile-products per branch, 32 KiB
// written and re-read per branch) is then almost all of the work.  This kernel applies the factors instead:
//     internal edge   m = R (e_b o (L v))          two 64x64 x (64x16) products
//     leaf edge       m = R (e_b o L[:, codon])    one product; nothing at all when the leaf is a gap in all 16 columns
//     all-gap subtree m = P 1 = 1                  nothing (the reference gets 1 +- 1 ulp from its row sums)
//     edge above an edge whose sibling is an all-gap subtree:  y <- e_b o y, no product (L R = I; the message stays in eigen
//                                                  coordinates until two children meet -- see `struct Msg` in the kernel)
// with e_b[k] = exp(lambda_k t_b) from edge_scale_kernel (one exponential per eigenvalue, branch and evaluation).
//
// Mapping: ONE WORK-GROUP PER TILE (16 columns of one evaluation), its four waves split every product by OUTPUT ROWS:
// wave w computes states 16w..16w+15 of the result with 16 x v_mfma_f64_16x16x4_f64 whose A operands -- the wave's
// 16 rows of R and of L -- stay in registers for the whole tile (no LDS staging of matrices at all).  The B operand is
// the full 64x16 tile: the MFMA result layout (lane l, register r = state 16w + (l>>4) + 4r, column l&15) is exactly the
// B layout of k-step q = 4w + r in the SAME lane, so the all-gather between two products is a same-lane exchange
// through LDS: two ds_write_b128 per wave, one barrier, eight ds_read_b128 (double-buffered, 16 KiB per work-group).
// Element-wise work (scaling by e_b, message products, clamps) touches 4 registers per lane instead of 16.
//
// What is NOT reproduced entry by entry: the reference's clamp of P's negative noise and its diagonal fix
// (src/instance.hpp:602-640) act on matrix entries that are never formed here.  Messages are clamped at zero instead
// (identical for leaf messages, which ARE columns of P); the raw row sums are validated per evaluation by
// edge_scale_kernel (they do not depend on the branch); the entry-wise lower bound cannot fail for eigen-systems from the symmetric Jacobi solvers
// (P = D^-1/2 U e^{Lt} U^T D^1/2 with U orthonormal to 1e-15).  Caller-supplied eigen-systems and long blocks take the
// explicit path (expm.hip + prune_fixed), which keeps the entry-wise post-processing.  DESIGN.md section 3.6.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdlib>
#include <type_traits>

#include "kernels.h"

namespace pcsf {

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

namespace {

// position of state k in the "row-split" order [wave][lane group g][register r], k = 16 wave + g + 4 r
__device__ __forceinline__ int row_split_pos(int k) { return ((k >> 4) * 4 + (k & 3)) * 4 + ((k >> 2) & 3); }

struct Vec4 {
    double v[4];
};

}  // namespace

// ---- e_b[k] = exp(lambda_k * t_b) for every branch of every evaluation ----------------------------------------------
// t_b = (double)(float)(len_b * x) like the reference's float tree (src/instance.hpp:299-307,498).  One work-group per
// evaluation (a thread computes ~50 exponentials; one launch per round, thousands of evaluations); states are stored in
// row-split order.
//
// Validation (the reference rejects a model whose raw row sums of P(t) are off by more than 1e-6,
// src/instance.hpp:632-636): sum_j P_ij(t) = sum_k R_ik e^{lambda_k t} c_k with c = L 1.  Q 1 = 0 makes 1 the right
// eigenvector of eigenvalue 0, so c_k vanishes (to rounding) for every other k and the row sums do not depend on t:
// they are checked ONCE per evaluation, for the longest branch (smallest e_k: the largest possible deviation), by the
// first wave of the evaluation's work-group.
__global__ __launch_bounds__(256) void edge_scale_kernel(const ExpmEval* __restrict__ evals, uint32_t n_branches, const float* __restrict__ branch_len,
                                                         const double* __restrict__ eig_right, const double* __restrict__ eig_left,
                                                         const double* __restrict__ eig_lambda, double* __restrict__ scale_out,
                                                         uint32_t* __restrict__ fail, uint32_t dense_rows)
{
    // one work-group per evaluation; wave w takes branches w, w+4, ...; lane k the eigenvalue k.  The branch lengths go through LDS
    // first (one coalesced load; reading them one by one in the loops below was ~250 dependent round trips to L2 per thread -- most
    // of this kernel's 0.17 ms, which every round of the lock-step optimiser waits for).
    __shared__ double ec[64];
    __shared__ float lens[512];
    const ExpmEval ev = evals[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const double lam = eig_lambda[static_cast<size_t>(ev.eig) * 64 + lane];
    const int pos = row_split_pos(lane);
    double* __restrict__ dst = scale_out + static_cast<size_t>(dense_rows ? blockIdx.x : ev.table) * n_branches * 64;   // (the failure flag stays at ev.table)
    const bool staged = n_branches <= 512;
    if (staged) {
        for (uint32_t b = tid; b < n_branches; b += 256) lens[b] = branch_len[b];
        __syncthreads();
    }
    float longest = 0.0f;
    for (uint32_t b = wave; b < n_branches; b += 4) {
        const float len = staged ? lens[b] : branch_len[b];
        const double t = static_cast<double>(static_cast<float>(static_cast<double>(len) * ev.x));
        dst[static_cast<size_t>(b) * 64 + pos] = exp(lam * t);
    }
    // The evaluation's row-sum check, by all four waves: thread t owns 16 consecutive entries of row t >> 2 of L and of R (128
    // contiguous bytes each: the first version had ONE wave walk both matrices with a row per lane -- 128 uncoalesced wave-loads,
    // ~150 us of a ~185 us launch that sits on every round's critical path).  c = L 1 first, then sum_k R_ik e_k c_k at the longest branch.
    for (uint32_t i = lane; i < n_branches; i += 64) longest = fmaxf(longest, staged ? lens[i] : branch_len[i]);
    for (int sft = 32; sft > 0; sft >>= 1) longest = fmaxf(longest, __shfl_xor(longest, sft));
    const int row = tid >> 2, part = tid & 3;
    {
        const double2_t* __restrict__ L2 = reinterpret_cast<const double2_t*>(eig_left + static_cast<size_t>(ev.eig) * 4096 + row * 64 + 16 * part);
        double c = 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double2_t v = L2[u];
            c += v.x;
            c += v.y;
        }
        c += __shfl_xor(c, 1);
        c += __shfl_xor(c, 2);
        if (part == 0) {
            const double t = static_cast<double>(static_cast<float>(static_cast<double>(longest) * ev.x));
            ec[row] = exp(eig_lambda[static_cast<size_t>(ev.eig) * 64 + row] * t) * c;
        }
    }
    __syncthreads();
    {
        const double2_t* __restrict__ R2 = reinterpret_cast<const double2_t*>(eig_right + static_cast<size_t>(ev.eig) * 4096 + row * 64 + 16 * part);
        double sum = 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const double2_t v = R2[u];
            sum += v.x * ec[16 * part + 2 * u];
            sum += v.y * ec[16 * part + 2 * u + 1];
        }
        sum += __shfl_xor(sum, 1);
        sum += __shfl_xor(sum, 2);
        if (part == 0 && !(fabs(sum - 1.0) <= 1e-6)) fail[ev.table] = 1u;
    }
}

__device__ __forceinline__ unsigned long long fact_clock()
{
    unsigned long long t;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

// ---- a tile's op list, compiled once per call -------------------------------------------------------------------------
// What prune_factored's work-groups derived from the op facts at the start of EVERY evaluation of a tile (rounds 2-4: a block of the
// reference's test file is evaluated ~400 times under OMEGA, always with the same species present): which ops leave the all-ones tile
// (skipped), which pushes only carry a message across an all-gap subtree (with their pops: one scaling), which ops only scale a message
// in eigen form, the branch of every remaining op's edge by the op's rank, and per op whether the message before it / the tile its pop
// takes back is all ones.  One wave per tile, the passes of the kernel below moved here word for word; the work-group of an evaluation
// then loads 1.5 KB and walks the ops by rank ("compile the op list once per (block, tile)").
__global__ __launch_bounds__(64) void compile_tiles_kernel(const FactoredParams p, const TileDesc* __restrict__ tiles, uint32_t n_tiles, uint32_t* __restrict__ programs,
                                                           uint32_t* __restrict__ site_tile)
{
    __shared__ unsigned char lpush[128];
    __shared__ uint32_t lout[3 * 128];
    const uint32_t tile = blockIdx.x;
    if (tile >= n_tiles) return;
    const TileDesc td = tiles[tile];
    const int lane = threadIdx.x;
    const int n_ops = p.n_ops;
    // the leaves with a codon in the tile: lane = leaf (two passes for 128)
    unsigned long long has[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int leaf = 64 * k + lane;
        bool any = false;
        if (leaf < p.n_leaves)
            for (uint32_t s2 = 0; s2 < td.n_sites; ++s2) any |= p.codons[static_cast<size_t>(leaf) * p.leaf_stride + td.site0 + s2] < 64u;
        has[k] = __ballot(any);
    }
    const unsigned long long has0 = has[0], has1 = has[1];
    uint32_t opw[2], mt[2];
    unsigned long long ol0[2], ol1[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int j = 64 * k + lane;
        const bool valid = j < n_ops;
        opw[k] = valid ? p.program[j] : (static_cast<uint32_t>(F_ROOT_DEV) << 28);
        mt[k] = valid ? p.op_meta[j] : 0u;
        ol0[k] = valid ? p.op_leaves[2 * j] : 0ull;
        ol1[k] = valid ? p.op_leaves[2 * j + 1] : 0ull;
    }
    lpush[lane] = 0;
    lpush[64 + lane] = 0;
    for (int i = lane; i < 3 * 128; i += 64) lout[i] = i < 128 ? (static_cast<uint32_t>(F_ROOT_DEV) << 28) : 0u;
    __syncthreads();
    unsigned long long ones_after[2], active[2], carried[2], push_carries[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int j = 64 * k + lane;
        ones_after[k] = __ballot(j < n_ops && ((ol0[k] & has0) | (ol1[k] & has1)) == 0ull);
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int j = 64 * k + lane;
        const uint32_t code = opw[k] >> 28;
        const bool before = lane > 0 ? ((ones_after[k] >> (lane - 1)) & 1ull) : (k == 1 && (ones_after[0] >> 63));   // op j - 1 left all ones
        const int parking = static_cast<int>(mt[k] >> 18);   // the parking op of a pop (>= 1)
        const bool parked_ones = parking > 0 && ((((parking - 1) >> 6) ? ones_after[1] : ones_after[0]) >> ((parking - 1) & 63) & 1ull);
        const bool c = j < n_ops && code >= F_MM_POP0_DEV && code < F_ROOT_DEV && before && !parked_ones && parking < 128;
        carried[k] = __ballot(c);
        if (c) lpush[parking] = 1;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 2; ++k) push_carries[k] = __ballot(64 * k + lane < n_ops && lpush[64 * k + lane] != 0);
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int j = 64 * k + lane;
        const uint32_t code = opw[k] >> 28;
        const bool is_push = code >= F_MM_PUSH_CHERRY0_DEV && code < F_MM_POP0_DEV, is_root = code >= F_ROOT_DEV;
        const bool after = (ones_after[k] >> lane) & 1ull;
        const bool before = lane > 0 ? ((ones_after[k] >> (lane - 1)) & 1ull) : (k == 1 && (ones_after[0] >> 63));
        const bool trivial = j < n_ops && ((after && !is_root && (!is_push || before)) || ((carried[k] >> lane) & 1ull));
        active[k] = __ballot(j < n_ops && !trivial);
    }
    const int n_act0 = __popcll(active[0]);
    const int n_active = n_act0 + __popcll(active[1]);
#pragma unroll
    for (int k = 0; k < 2; ++k)
        if ((active[k] >> lane) & 1ull) {
            const int j = 64 * k + lane;
            const int at = (k ? n_act0 : 0) + __popcll(active[k] & ((1ull << lane) - 1ull));
            const uint32_t code = opw[k] >> 28;
            const bool has_edge = code != F_CHERRY_DEV && code < F_ROOT_DEV;
            const int a = static_cast<int>((opw[k] >> 14) & 0x3fffu);
            const bool a_has = ((a < 64 ? has0 >> a : has1 >> (a - 64)) & 1ull) != 0;
            const int parking = static_cast<int>(mt[k] >> 18);
            const bool parked_ones = parking > 0 && ((((parking - 1) >> 6) ? ones_after[1] : ones_after[0]) >> ((parking - 1) & 63) & 1ull);
            const bool before = lane > 0 ? ((ones_after[k] >> (lane - 1)) & 1ull) : (k == 1 && (ones_after[0] >> 63));
            const bool scales_only = !before && ((code == F_MM_LEAF_DEV && !a_has) || (code >= F_MM_POP0_DEV && code < F_ROOT_DEV && parked_ones) || ((push_carries[k] >> lane) & 1ull));
            const bool ones_before = j > 0 && before;
            lout[at] = opw[k];
            lout[128 + at] = (has_edge ? static_cast<uint32_t>(p.slot_branch[mt[k] & 0x3fffu]) : 0u) | (scales_only ? 0x80000000u : 0u);
            lout[256 + at] = (mt[k] & 0x3fffu) | (((mt[k] >> 14) & 0xfu) << 14) | (ones_before ? 1u << 18 : 0u) | (parked_ones ? 1u << 19 : 0u);
        }
    __syncthreads();
    uint32_t* __restrict__ out = programs + static_cast<size_t>(tile) * TILE_PROGRAM_WORDS;
    if (lane == 0) {
        out[0] = static_cast<uint32_t>(n_active);
        out[1] = out[2] = out[3] = 0u;
        out[4] = static_cast<uint32_t>(has0);
        out[5] = static_cast<uint32_t>(has0 >> 32);
        out[6] = static_cast<uint32_t>(has1);
        out[7] = static_cast<uint32_t>(has1 >> 32);
        site_tile[td.site0] = tile;
    }
    for (int i = lane; i < 3 * 128; i += 64) out[8 + i] = lout[i];
}

// ---- the pruning kernel ----------------------------------------------------------------------------------------------
// DEEP: the tree parks more than MAX_STACK_DEPTH_DEV tiles (none of the reference's eleven trees does): the parked tiles live in
// scratch memory, dynamically indexed; otherwise in three register sets written and read with value selects (no scratch at all).
// OWN: the kernel makes its edge scales itself (p.eval_x: e_b[k] = exp(lambda_k t_b) for the lane's four states) and checks the evaluation's
// row sums in the tile flagged FACTORED_JOB_VALIDATES -- no edge_scale_kernel, no rows of scales in HBM (99 KB per evaluation of a round for
// 100 leaves); otherwise it reads edge_scale_kernel's rows.  Same exp() on the same operands: the same bits either way.
template <bool ANC, bool DEEP, bool OWN>
__global__ __launch_bounds__(256, 3) void prune_factored(const FactoredParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_fact[];
    double2_t* xbuf = reinterpret_cast<double2_t*>(lds_fact);                      // [2][4 waves][2 pairs][64 lanes]
    double* red = reinterpret_cast<double*>(lds_fact + 2 * 4 * 2 * 64 * 16);       // [2][4][16]
    double* lcgap = red + 2 * 4 * 16;                                              // [64] c = L 1 (eigen coordinates of a gap column's all-ones message)
    uint32_t* lprog = reinterpret_cast<uint32_t*>(lcgap + 64);                     // [n_ops rounded up to 64]
    int32_t* lslot = reinterpret_cast<int32_t*>(lprog + ((p.n_ops + 63) & ~63u));  // [n_edges] slot -> branch
    uint32_t* lrow_edge = reinterpret_cast<uint32_t*>(lslot + ((p.n_edges + 3) & ~3u));      // [128] by rank among the ops that run: branch of the op's edge (row of edge_scale)
    unsigned char* lcod = reinterpret_cast<unsigned char*>(lrow_edge + 128);                  // [n_leaves][16]
    unsigned char* lany = lcod + static_cast<size_t>(p.n_leaves) * 16;             // [n_leaves]: the leaf has a codon in the tile
    unsigned char* lpush = lany + p.n_leaves;                                      // [128]: the push op only carries its message across an all-gap subtree
    float* lblen = reinterpret_cast<float*>(lds_fact + ((static_cast<size_t>(lpush + 128 - lds_fact) + 15) & ~size_t(15)));   // OWN: [n_branches rounded up to 4] branch lengths
    double* lec = reinterpret_cast<double*>(lblen + ((p.n_branches + 3) & ~3u));    // OWN: [4 waves][64] e^{lambda_k t_long} c_k of the row-sum check

    const FactoredJob job = p.jobs[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sj = lane & 15, g = lane >> 4;
    const bool stamping = p.stamps != nullptr;
    const unsigned long long t_begin = stamping ? fact_clock() : 0ull;
    unsigned long long t_load = 0, t_prod = 0, n_load = 0, n_prod = 0, t_class[4] = {0, 0, 0, 0}, n_class[4] = {0, 0, 0, 0};

    // Everything the prologue needs from global memory is REQUESTED before anything waits for it (the first version ran seven dependent
    // round trips for the codon bytes alone -- each iteration's ballot waited for its own load -- then the program, then the operand
    // rows, then the op facts: 16 k cycles before the first op of a tile whose ten products take 14 k).
    // (a) this wave's 16 rows of R and L as MFMA A operands: k-step q needs M[16 wave + (lane & 15)][4q + (lane >> 4)]
    const double* __restrict__ Rg = p.eig_right ? p.eig_right + static_cast<size_t>(job.eig) * 4096 : nullptr;   // (the row-major images may be absent when the
    const double* __restrict__ Lg = p.eig_left ? p.eig_left + static_cast<size_t>(job.eig) * 4096 : nullptr;     //  fragment-ordered ones are there)
    double RA[16], LA[16];
    if (p.eig_right_frag) {   // the same values from the images stored in this order: 512 contiguous bytes per wave and k-step
        const double* __restrict__ Rf = p.eig_right_frag + static_cast<size_t>(job.eig) * 4096 + wave * 1024 + lane;
        const double* __restrict__ Lf = p.eig_left_frag + static_cast<size_t>(job.eig) * 4096 + wave * 1024 + lane;
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            RA[q] = Rf[64 * q];
            LA[q] = Lf[64 * q];
        }
    } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            RA[q] = Rg[(16 * wave + sj) * 64 + 4 * q + g];
            LA[q] = Lg[(16 * wave + sj) * 64 + 4 * q + g];
        }
    }
    // (b) the static facts of ops 64 k + lane (schedule.hpp `program_facts`; trees of at most 128 leaves)
    const int n_ops = static_cast<int>(p.n_ops);
    const bool facts = p.op_meta != nullptr;
    uint32_t opw[2], mt[2];
    unsigned long long ol0[2], ol1[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int j = 64 * k + lane;
        const bool valid = facts && j < n_ops && p.tile_programs == nullptr;
        opw[k] = valid ? p.program[j] : (static_cast<uint32_t>(F_ROOT_DEV) << 28);
        mt[k] = valid ? p.op_meta[j] : 0u;
        ol0[k] = valid ? p.op_leaves[2 * j] : 0ull;
        ol1[k] = valid ? p.op_leaves[2 * j + 1] : 0ull;
    }
    // (b') ... or the tile's compiled op list (round 5): by rank, 64 k + lane
    const bool compiled = p.tile_programs != nullptr;
    const uint32_t* __restrict__ tprog = compiled ? p.tile_programs + static_cast<size_t>(p.site_tile[job.site0]) * TILE_PROGRAM_WORDS : nullptr;
    uint32_t cop[2] = {0, 0}, ci0[2] = {0, 0}, ci1[2] = {0, 0};
    if (compiled) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            cop[k] = tprog[8 + 64 * k + lane];
            ci0[k] = tprog[8 + 128 + 64 * k + lane];
            ci1[k] = tprog[8 + 256 + 64 * k + lane];
        }
    }
    // (c) the tile's codons (16 consecutive lanes hold one leaf's columns): the first 128 leaves in registers, ...
    unsigned cc[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int idx = it * 256 + tid, leaf = idx >> 4, s = idx & 15;
        cc[it] = 64u;
        if (leaf < p.n_leaves && static_cast<uint32_t>(s) < job.n_sites) cc[it] = p.codons[static_cast<size_t>(leaf) * p.leaf_stride + job.site0 + s];
    }
    for (int i = tid; i < static_cast<int>((p.n_ops + 63) & ~63u); i += 256) lprog[i] = p.program[i < static_cast<int>(p.n_ops) ? i : static_cast<int>(p.n_ops) - 1];
    for (int i = tid; i < p.n_edges; i += 256) lslot[i] = p.slot_branch[i];
    if (tid < 128) lpush[tid] = 0;
    // ... then into LDS, with per leaf whether it has a codon in any of the 16 columns
#pragma unroll
    for (int it = 0; it < 8; ++it) {
        const int idx = it * 256 + tid, leaf = idx >> 4;
        const unsigned c = cc[it] > 64u ? 64u : cc[it];
        const unsigned long long has = __ballot(c != 64u);
        if (leaf < p.n_leaves) {
            lcod[idx] = static_cast<unsigned char>(c);
            if ((idx & 15) == 0) lany[leaf] = ((has >> (16 * (lane >> 4))) & 0xffffull) != 0;
        }
    }
    for (int base = 2048; base < p.n_leaves * 16; base += 256) {   // trees of more than 128 leaves
        const int idx = base + tid, leaf = idx >> 4, s = idx & 15;
        unsigned c = 64u;
        if (leaf < p.n_leaves && static_cast<uint32_t>(s) < job.n_sites) {
            c = p.codons[static_cast<size_t>(leaf) * p.leaf_stride + job.site0 + s];
            c = c > 64u ? 64u : c;
        }
        const unsigned long long has = __ballot(c != 64u);
        if (leaf < p.n_leaves) {
            lcod[idx] = static_cast<unsigned char>(c);
            if (s == 0) lany[leaf] = ((has >> (16 * (lane >> 4))) & 0xffffull) != 0;
        }
    }
    const double* __restrict__ scale = OWN ? nullptr : p.edge_scale + static_cast<size_t>(job.eval) * p.n_branches * 64 + (wave * 4 + g) * 4;
    // OWN: this lane's four eigenvalues (states 16 wave + g + 4r), the evaluation's tree scale, the branch lengths
    double lam4[4] = {0.0, 0.0, 0.0, 0.0};
    double x_eval = 0.0;
    if (OWN) {
#pragma unroll
        for (int r = 0; r < 4; ++r) lam4[r] = p.eig_lambda[static_cast<size_t>(job.eig) * 64 + 16 * wave + g + 4 * r];
        x_eval = p.eval_x[job.eval];
        for (uint32_t b = tid; b < p.n_branches; b += 256) lblen[b] = p.branch_len[b];
    }
    // e_b for this lane's rows: t_b = (double)(float)(len_b x) like the reference's float tree (src/instance.hpp:299-307,498)
    auto own_scale = [&](int branch) __attribute__((always_inline)) -> Vec4 {
        const double t = static_cast<double>(static_cast<float>(static_cast<double>(lblen[branch]) * x_eval));
        Vec4 e;
#pragma unroll
        for (int r = 0; r < 4; ++r) e.v[r] = exp(lam4[r] * t);
        return e;
    };
    // a leaf's column of L for this lane's rows 16 wave + g + 4r: L[row][codon], from the row-major image (+ 4r * 64 + codon) or from the
    // fragment-ordered one, where element [row][col] sits at ((row >> 4) * 16 + (col >> 2)) * 64 + (col & 3) * 16 + (row & 15)
    const bool l_frag = p.eig_left_frag != nullptr;
    const double* __restrict__ Lrows = l_frag ? p.eig_left_frag + static_cast<size_t>(job.eig) * 4096 + wave * 1024 + g : Lg + static_cast<size_t>(16 * wave + g) * 64;
    auto l_entry = [&](int r, unsigned col) __attribute__((always_inline)) -> double {
        return l_frag ? Lrows[(col >> 2) * 64 + (col & 3) * 16 + 4 * r] : Lrows[r * 256 + col];
    };
    __syncthreads();
    if (OWN && (job.pad & FACTORED_JOB_VALIDATES)) {
        // The reference rejects a model whose raw row sums of P(t) are off by more than 1e-6 (src/instance.hpp:632-636): sum_j P_ij(t) =
        // sum_k R_ik e^{lambda_k t} c_k with c = L 1, at the longest branch (edge_scale_kernel has the argument).  Every wave makes the 64
        // products e^{lambda_k t} c_k for itself and checks its sixteen rows of R, which it holds as MFMA operands.
        const double t_long = static_cast<double>(static_cast<float>(static_cast<double>(p.longest_branch) * x_eval));
        double* ec = lec + wave * 64;
        ec[lane] = exp(p.eig_lambda[static_cast<size_t>(job.eig) * 64 + lane] * t_long) * p.eig_c[static_cast<size_t>(job.eig) * 64 + lane];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        double sum = 0.0;   // row 16 wave + sj, this lane's k = 4q + g
#pragma unroll
        for (int q = 0; q < 16; ++q) sum += RA[q] * ec[4 * q + g];
        sum += __shfl_xor(sum, 16);
        sum += __shfl_xor(sum, 32);
        if (!(fabs(sum - 1.0) <= 1e-6)) p.fail[job.eval] = 1u;
    }
    // leaves 0..127 as two wave-uniform bit masks: most ops of a sparse tile then cost a few scalar instructions (the op loop's LDS
    // look-ups, not the products, were 70 % of a tile's time: PCSF_FACTORED_STAMPS)
    const unsigned long long t_prologue = stamping ? fact_clock() - t_begin : 0ull;
    const unsigned long long has0 = compiled ? (static_cast<unsigned long long>(tprog[4]) | (static_cast<unsigned long long>(tprog[5]) << 32))
                                             : __ballot(lane < p.n_leaves && lany[lane] != 0);
    const unsigned long long has1 = compiled ? (static_cast<unsigned long long>(tprog[6]) | (static_cast<unsigned long long>(tprog[7]) << 32))
                                             : __ballot(64 + lane < p.n_leaves && lany[64 + lane < p.n_leaves ? 64 + lane : 0] != 0);
    auto leaf_has = [&](int a) __attribute__((always_inline)) -> bool {
        if (a < 64) return (has0 >> a) & 1ull;
        if (a < 128) return (has1 >> (a - 64)) & 1ull;
        return lany[a] != 0;
    };

    int nprod = 0;
    // out <- M v for the 64x16 tile whose rows 16 wave + g + 4r this lane holds in v[r]: all-gather through LDS, then
    // 16 MFMAs in two accumulator chains
    // (two copies with the operand array named, not passed: a reference parameter made the compiler merge the two call paths and
    // keep R's and L's rows in scratch memory)
    auto matvecR = [&](const Vec4& v) __attribute__((always_inline)) -> Vec4 {
        const unsigned long long t_in = stamping ? fact_clock() : 0ull;
        double2_t* xb = xbuf + (nprod & 1) * (4 * 2 * 64);
        ++nprod;
        xb[(wave * 2 + 0) * 64 + lane] = double2_t{v.v[0], v.v[1]};
        xb[(wave * 2 + 1) * 64 + lane] = double2_t{v.v[2], v.v[3]};
        __syncthreads();
        double2_t b[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) b[i] = xb[i * 64 + lane];  // k-steps q = 2i, 2i + 1
        double4_t a0 = double4_t{0.0, 0.0, 0.0, 0.0}, a1 = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(RA[2 * i], b[i].x, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(RA[2 * i + 1], b[i].y, a1, 0, 0, 0);
        }
        Vec4 out;
#pragma unroll
        for (int r = 0; r < 4; ++r) out.v[r] = a0[r] + a1[r];
        if (stamping) {
            asm volatile("" ::"v"(out.v[0]), "v"(out.v[3]));
            t_prod += fact_clock() - t_in;
            ++n_prod;
        }
        return out;
    };
    auto matvecL = [&](const Vec4& v) __attribute__((always_inline)) -> Vec4 {
        const unsigned long long t_in = stamping ? fact_clock() : 0ull;
        double2_t* xb = xbuf + (nprod & 1) * (4 * 2 * 64);
        ++nprod;
        xb[(wave * 2 + 0) * 64 + lane] = double2_t{v.v[0], v.v[1]};
        xb[(wave * 2 + 1) * 64 + lane] = double2_t{v.v[2], v.v[3]};
        __syncthreads();
        double2_t b[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) b[i] = xb[i * 64 + lane];  // k-steps q = 2i, 2i + 1
        double4_t a0 = double4_t{0.0, 0.0, 0.0, 0.0}, a1 = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(LA[2 * i], b[i].x, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(LA[2 * i + 1], b[i].y, a1, 0, 0, 0);
        }
        Vec4 out;
#pragma unroll
        for (int r = 0; r < 4; ++r) out.v[r] = a0[r] + a1[r];
        if (stamping) {
            asm volatile("" ::"v"(out.v[0]), "v"(out.v[3]));
            t_prod += fact_clock() - t_in;
            ++n_prod;
        }
        return out;
    };
    // Rows of edge_scale.  Read one by one they were 31 dependent round trips (~620 cycles each) in a tile with four species present --
    // a fifth of its time -- nearly all of them for the edges of chains (messages in eigen form: the row is all an edge needs).  With
    // the op facts the edges of the next SIXTEEN ops that run are known ahead: lane (sj, g) fetches its 32 bytes of the row of op
    // `rank0 + sj`, one batch for all sixteen, and an op takes its row from the lanes of its rank with ds_bpermute (no memory access).
    // A row fetched and not used costs 32 bytes.  (The rows of an op's leaves are still read where they are used: their wait is the
    // wait for the leaf's column of L, issued beside them; holding them too took the kernel from three waves per SIMD to two.)
    double pre[4] = {1.0, 1.0, 1.0, 1.0};
    int n_active = 0;
    uint32_t chain16 = 0;   // bit j: the op of rank rank0 + j only scales a message in eigen form (see the op loop)
    auto load_batch = [&](int rank0) __attribute__((always_inline)) {
        const unsigned long long t_in = stamping ? fact_clock() : 0ull;
        const int idx = rank0 + sj < n_active ? rank0 + sj : n_active - 1;
        const uint32_t row = compiled ? (idx < 64 ? __shfl(ci0[0], idx) : __shfl(ci0[1], idx - 64)) : lrow_edge[idx];
        chain16 = static_cast<uint32_t>(__ballot((row >> 31) != 0u && rank0 + sj < n_active)) & 0xffffu;   // lanes 0..15: sj = 0..15 of lane group 0
        if (OWN) {
            const Vec4 e = own_scale(static_cast<int>(row & 0x7fffffffu));
#pragma unroll
            for (int r = 0; r < 4; ++r) pre[r] = e.v[r];
        } else {
            const double2_t* s0 = reinterpret_cast<const double2_t*>(scale + static_cast<size_t>(row & 0x7fffffffu) * 64);
            const double2_t a0 = s0[0], a1 = s0[1];
            pre[0] = a0.x; pre[1] = a0.y; pre[2] = a1.x; pre[3] = a1.y;
        }
        if (stamping) {
            t_load += fact_clock() - t_in;   // fact_clock waits for the loads
            ++n_load;
        }
    };
    int rank = 0;   // of the op being run among the ops that run (PF: with the op facts; otherwise rows are read from memory)
    auto load_scale = [&](int branch, int kind, auto pf) __attribute__((always_inline)) -> Vec4 {
        if constexpr (decltype(pf)::value) {
            if (kind == 0) {
                const int src = (lane & 48) | (rank & 15);
                Vec4 e;
#pragma unroll
                for (int r = 0; r < 4; ++r) e.v[r] = __shfl(pre[r], src);
                return e;
            }
        }
        if (OWN) return own_scale(branch);
        const unsigned long long t_in = stamping ? fact_clock() : 0ull;
        const double2_t* s2 = reinterpret_cast<const double2_t*>(scale + static_cast<size_t>(branch) * 64);
        const double2_t lo = s2[0], hi = s2[1];
        if (stamping) {
            t_load += fact_clock() - t_in;   // fact_clock waits for the loads
            ++n_load;
        }
        return Vec4{{lo.x, lo.y, hi.x, hi.y}};
    };
    // c = L 1: the eigen-coordinates of the all-ones message (a gap column of a leaf that has codons elsewhere in the tile); kept in LDS --
    // it is read by the few leaves with such columns, and eight registers less keep the kernel at three waves per SIMD
    {
        double rowsum = 0.0;   // row 16 wave + sj of L, this lane's 16 entries, then over the four lane groups
#pragma unroll
        for (int q = 0; q < 16; ++q) rowsum += LA[q];
        rowsum += __shfl_xor(rowsum, 16);
        rowsum += __shfl_xor(rowsum, 32);
        if (g == 0) lcgap[16 * wave + sj] = rowsum;
    }
    const double* my_cgap = lcgap + 16 * wave + g;   // + 4r: row 16 wave + g + 4r (written by this wave: no barrier needed, only the wave's own order)

    // A message is kept in one of two forms.  Real: the 64x16 tile itself.  Eigen: y with message = max(R y, 0) still to be formed.
    // P(t) = R e^{Lambda t} L and L R = I, so an edge applied to a message in eigen form is just y <- e_b o y -- no product -- and
    // a chain of edges whose other children are all-gap subtrees (the rule in a sparse block: 4 of 100 species present) costs one
    // L-product at its start and one R-product where the chain meets another message, instead of two products per edge.  Real form
    // is only needed for the element-wise product of two children and at the root.  `mask`: columns in which the whole subtree is
    // gaps -- their message is exactly 1 whatever the form (bit s = site s); all 16 bits set = the all-ones tile.  Dense tiles do the
    // same products in the same order as before.
    struct Msg {
        Vec4 v;
        uint32_t mask;
        bool eig;
    };
    auto to_real = [&](const Msg& m) __attribute__((always_inline)) -> Msg {
        if (m.mask == 0xffffu) return Msg{Vec4{{1.0, 1.0, 1.0, 1.0}}, 0xffffu, false};
        if (!m.eig) return m;
        Msg out{matvecR(m.v), m.mask, false};
        const bool gap_col = (m.mask >> sj) & 1u;
#pragma unroll
        for (int r = 0; r < 4; ++r) out.v.v[r] = gap_col ? 1.0 : fmax(out.v.v[r], 0.0);
        return out;
    };
    auto apply_edge = [&](int slot_index, const Msg& m, auto pf) __attribute__((always_inline)) -> Msg {   // the edge of program slot slot_index
        if (m.mask == 0xffffu) return m;
        const Vec4 e = load_scale(decltype(pf)::value ? 0 : lslot[slot_index], 0, pf);
        Msg out{m.eig ? m.v : matvecL(m.v), m.mask, true};
#pragma unroll
        for (int r = 0; r < 4; ++r) out.v.v[r] *= e.v[r];
        return out;
    };
    auto leaf = [&](int a, int kind, auto pf) __attribute__((always_inline)) -> Msg {
        if (!leaf_has(a)) return Msg{Vec4{{1.0, 1.0, 1.0, 1.0}}, 0xffffu, false};
        const unsigned code = lcod[a * 16 + sj];
        const uint32_t mask = static_cast<uint32_t>(__ballot(code == 64u)) & 0xffffu;   // lanes 0..15 hold the 16 sites
        if (mask == 0xffffu) return Msg{Vec4{{1.0, 1.0, 1.0, 1.0}}, mask, false};
        const Vec4 e = load_scale(a, kind, pf);
        const unsigned col = code == 64u ? 0u : code;
        Msg out{Vec4{}, mask, true};
#pragma unroll
        for (int r = 0; r < 4; ++r) out.v.v[r] = e.v[r] * (code == 64u ? my_cgap[4 * r] : l_entry(r, col));
        return out;
    };
    auto mul = [&](const Msg& x, const Msg& y) __attribute__((always_inline)) -> Msg {
        if (x.mask == 0xffffu) return y;
        if (y.mask == 0xffffu) return x;
        const Msg a = to_real(x), b = to_real(y);
        Msg out{Vec4{}, a.mask & b.mask, false};
#pragma unroll
        for (int r = 0; r < 4; ++r) out.v.v[r] = a.v.v[r] * b.v.v[r];
        return out;
    };

    // parked messages: pushes and pops nest, so a stack pointer replaces the slot numbers of the op codes (which deep
    // trees do not carry, schedule.hpp).  An all-ones tile is never stored or loaded.
    Vec4 stk[DEEP ? MAX_STACK_DEPTH_DEEP_DEV : 1], s0{{1.0, 1.0, 1.0, 1.0}}, s1 = s0, s2 = s0;
    uint32_t stk_mask[DEEP ? MAX_STACK_DEPTH_DEEP_DEV : 1];
    unsigned long long packed_masks = 0;   // shallow trees: 16 bits per parked message, no indexed array (it would live in scratch)
    uint32_t stk_eig = 0;  // bit d: parked message d is in eigen form
    auto park = [&](int d, const Msg& m) __attribute__((always_inline)) {
        if (DEEP) stk_mask[d] = m.mask;
        else packed_masks = (packed_masks & ~(0xffffull << (16 * d))) | (static_cast<unsigned long long>(m.mask) << (16 * d));
        stk_eig = (stk_eig & ~(1u << d)) | (m.eig ? 1u << d : 0u);
        if (m.mask == 0xffffu) return;
        if (DEEP) {
            stk[d] = m.v;
        } else {   // value selects, component by component: selecting between the three ADDRESSES sends all of them to scratch memory
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s0.v[r] = d == 0 ? m.v.v[r] : s0.v[r];
                s1.v[r] = d == 1 ? m.v.v[r] : s1.v[r];
                s2.v[r] = d == 2 ? m.v.v[r] : s2.v[r];
            }
        }
    };
    auto parked = [&](int d) __attribute__((always_inline)) -> Msg {
        Msg m{Vec4{{1.0, 1.0, 1.0, 1.0}}, DEEP ? stk_mask[d] : static_cast<uint32_t>((packed_masks >> (16 * d)) & 0xffffull), ((stk_eig >> d) & 1u) != 0};
        if (m.mask == 0xffffu) return m;
        if (DEEP) {
            m.v = stk[d];
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) m.v.v[r] = d == 0 ? s0.v[r] : (d == 1 ? s1.v[r] : s2.v[r]);
        }
        return m;
    };
    Msg cur{Vec4{{1.0, 1.0, 1.0, 1.0}}, 0xffffu, false};
    // one op: `slot` and `depth` are the program slot and the number of parked tiles the op starts with; parked_ones: the tile a
    // pop takes back is known to be all ones whatever the registers hold (its parking op was skipped)
    auto run_op = [&](uint32_t op, int slot, int depth, bool parked_ones, auto pf) __attribute__((always_inline)) {
        const uint32_t code = op >> 28;
        const int a = (op >> 14) & 0x3fff, b = op & 0x3fff;
        if (code == F_CHERRY_DEV) {
            cur = mul(leaf(a, 1, pf), leaf(b, 2, pf));
        } else if (code == F_MM_LEAF_DEV) {
            cur = mul(apply_edge(slot, cur, pf), leaf(a, 1, pf));
        } else if (code >= F_MM_PUSH_CHERRY0_DEV && code < F_MM_POP0_DEV) {
            park(depth, apply_edge(slot, cur, pf));
            cur = mul(leaf(a, 1, pf), leaf(b, 2, pf));
        } else if (code >= F_MM_POP0_DEV && code < F_ROOT_DEV) {
            const Msg m = apply_edge(slot, cur, pf);
            if (parked_ones) cur = m;
            else cur = mul(parked(depth - 1), m);
        }   // F_ROOT, always the program's last op, runs after the loop (below): inside it the constants of log() were live across every op
    };

    using std::false_type;
    using std::true_type;
    if (compiled) {
        // the tile's compiled list: the ops that run, by rank (compile_tiles_kernel made what the passes of the next branch make)
        n_active = static_cast<int>(__builtin_amdgcn_readfirstlane(tprog[0]));
        while (rank < n_active) {
            if ((rank & 15) == 0) load_batch(rank);
            if ((chain16 >> (rank & 15)) & 1u) {   // an op that only applies its edge (see the next branch)
                if (cur.mask == 0xffffu) {
                    ++rank;
                    continue;
                }
                if (!cur.eig) {   // real form: the L-product, then the scale
                    const uint32_t i1 = rank < 64 ? __builtin_amdgcn_readlane(ci1[0], rank & 63) : __builtin_amdgcn_readlane(ci1[1], rank & 63);
                    cur = apply_edge(static_cast<int>(i1 & 0x3fffu), cur, true_type{});
                    ++rank;
                    continue;
                }
                const uint32_t run_bits = ~(chain16 >> (rank & 15));
                int n_run = run_bits ? __builtin_ctz(run_bits) : 16;
                n_run = n_run < 16 - (rank & 15) ? n_run : 16 - (rank & 15);
                n_run = n_run < n_active - rank ? n_run : n_active - rank;
                for (int j = 0; j < n_run; ++j) {
                    const int src = (lane & 48) | ((rank + j) & 15);
#pragma unroll
                    for (int r = 0; r < 4; ++r) cur.v.v[r] *= __shfl(pre[r], src);
                }
                rank += n_run;
                continue;
            }
            const uint32_t op = rank < 64 ? __builtin_amdgcn_readlane(cop[0], rank & 63) : __builtin_amdgcn_readlane(cop[1], rank & 63);
            const uint32_t i1 = rank < 64 ? __builtin_amdgcn_readlane(ci1[0], rank & 63) : __builtin_amdgcn_readlane(ci1[1], rank & 63);
            if (i1 & (1u << 18)) {   // whatever cur holds is from before a run of skipped ops
                cur.mask = 0xffffu;
                cur.eig = false;
            }
            const bool stamp_op = stamping && p.stamp_classes != 0;
            const unsigned long long t_op = stamp_op ? fact_clock() : 0ull;
            run_op(op, static_cast<int>(i1 & 0x3fffu), static_cast<int>((i1 >> 14) & 0xfu), (i1 & (1u << 19)) != 0u, true_type{});
            if (stamp_op) {
                const uint32_t cls = (op >> 28) == F_CHERRY_DEV ? 0u : ((op >> 28) == F_MM_LEAF_DEV ? 1u : ((op >> 28) < F_MM_POP0_DEV ? 2u : 3u));
                if ((op >> 28) < F_ROOT_DEV) {
                    const unsigned long long dt = fact_clock() - t_op;
#pragma unroll
                    for (uint32_t k = 0; k < 4; ++k) {
                        t_class[k] += cls == k ? dt : 0ull;
                        n_class[k] += cls == k ? 1ull : 0ull;
                    }
                }
            }
            ++rank;
        }
    } else if (facts) {
        // Skipping the ops of all-gap subtrees.  A trivial op still costs ~430 cycles of branchy scalar code, and 70 of the 100 ops of
        // a real tile (4 of 100 species present) are trivial.  schedule.hpp `program_facts` knows every op's leaf set, start slot,
        // stack depth and, for a pop, the parking op; with the tile's leaf bits a lane-parallel pass marks the ops whose result is
        // the all-ones tile, and the loop visits only the others (trees of at most 128 leaves: n_ops <= 128, two 64-bit sets).
        unsigned long long ones_after[2], active[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int j = 64 * k + lane;
            ones_after[k] = __ballot(j < n_ops && ((ol0[k] & has0) | (ol1[k] & has1)) == 0ull);
        }
        // A push whose second subtree is all gaps, and the pop that takes its tile back, only carry a message across that subtree: the
        // push applies its edge and parks, everything up to the pop is skipped, the pop multiplies by ones.  Such a pair becomes ONE
        // scaling -- the push applies its edge and leaves the message where it is, the pop is skipped like the ops between them (most
        // of the pushes and pops of a sparse tile: ~1,000 cycles each for two selects' worth of work).
        unsigned long long carried[2], push_carries[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int j = 64 * k + lane;
            const uint32_t code = opw[k] >> 28;
            const bool before = lane > 0 ? ((ones_after[k] >> (lane - 1)) & 1ull) : (k == 1 && (ones_after[0] >> 63));   // op j - 1 left all ones
            const int parking = static_cast<int>(mt[k] >> 18);   // the parking op of a pop (>= 1)
            const bool parked_ones = parking > 0 && ((((parking - 1) >> 6) ? ones_after[1] : ones_after[0]) >> ((parking - 1) & 63) & 1ull);
            const bool c = j < n_ops && code >= F_MM_POP0_DEV && code < F_ROOT_DEV && before && !parked_ones && parking < 128;
            carried[k] = __ballot(c);
            if (c) lpush[parking] = 1;   // all four waves write the same bytes
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2; ++k) push_carries[k] = __ballot(64 * k + lane < n_ops && lpush[64 * k + lane] != 0);
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int j = 64 * k + lane;
            const uint32_t code = opw[k] >> 28;
            const bool is_push = code >= F_MM_PUSH_CHERRY0_DEV && code < F_MM_POP0_DEV, is_root = code >= F_ROOT_DEV;
            const bool after = (ones_after[k] >> lane) & 1ull;
            const bool before = lane > 0 ? ((ones_after[k] >> (lane - 1)) & 1ull) : (k == 1 && (ones_after[0] >> 63));   // op j - 1 left all ones
            const bool trivial = j < n_ops && ((after && !is_root && (!is_push || before)) || ((carried[k] >> lane) & 1ull));
            active[k] = __ballot(j < n_ops && !trivial);
        }
        // the row of edge_scale the edge of every op that runs reads, by the op's rank among them (all four waves write the same values)
        const int n_act0 = __popcll(active[0]);
        n_active = n_act0 + __popcll(active[1]);
#pragma unroll
        for (int k = 0; k < 2; ++k)
            if ((active[k] >> lane) & 1ull) {
                const int at = (k ? n_act0 : 0) + __popcll(active[k] & ((1ull << lane) - 1ull));
                const uint32_t code = opw[k] >> 28;
                const bool has_edge = code != F_CHERRY_DEV && code < F_ROOT_DEV;
                // an op that, on a message in eigen form, is nothing but y <- e_b o y: the other operand of its product of messages is all
                // ones -- a leaf edge whose leaf has no codon in the tile, a pop whose parked tile is all ones
                const int a = static_cast<int>((opw[k] >> 14) & 0x3fffu);
                const bool a_has = ((a < 64 ? has0 >> a : has1 >> (a - 64)) & 1ull) != 0;
                const int parking = static_cast<int>(mt[k] >> 18);
                const bool parked_ones = parking > 0 && ((((parking - 1) >> 6) ? ones_after[1] : ones_after[0]) >> ((parking - 1) & 63) & 1ull);
                const bool before = lane > 0 ? ((ones_after[k] >> (lane - 1)) & 1ull) : (k == 1 && (ones_after[0] >> 63));
                const bool scales_only = !before && ((code == F_MM_LEAF_DEV && !a_has) || (code >= F_MM_POP0_DEV && code < F_ROOT_DEV && parked_ones) || ((push_carries[k] >> lane) & 1ull));
                lrow_edge[at] = (has_edge ? static_cast<uint32_t>(lslot[mt[k] & 0x3fffu]) : 0u) | (scales_only ? 0x80000000u : 0u);
            }
        __syncthreads();
        for (int k = 0; k < 2; ++k) {
            unsigned long long todo = k ? active[1] : active[0];
            while (todo) {
                if ((rank & 15) == 0) load_batch(rank);
                // A chain of edges above a lone species is a run of ops that only scale: y <- e_b o y, one after the other in the order the
                // ops would do it (same roundings), without an op's dispatch each -- 21 of the 38 ops of a tile with four species present.
                if ((chain16 >> (rank & 15)) & 1u) {   // an op that only applies its edge: never through run_op (a carrying push must not park)
                    if (cur.mask == 0xffffu) {   // nothing to scale (the op facts say the message has content: cannot happen)
                        todo &= todo - 1;
                        ++rank;
                        continue;
                    }
                    if (!cur.eig) {   // real form: the L-product, then the scale
                        const int l1 = __builtin_ctzll(todo);
                        todo &= todo - 1;
                        const uint32_t meta1 = __builtin_amdgcn_readlane(k ? mt[1] : mt[0], l1);
                        cur = apply_edge(static_cast<int>(meta1 & 0x3fffu), cur, true_type{});
                        ++rank;
                        continue;
                    }
                    const uint32_t run_bits = ~(chain16 >> (rank & 15));
                    int n_run = run_bits ? __builtin_ctz(run_bits) : 16;
                    n_run = n_run < 16 - (rank & 15) ? n_run : 16 - (rank & 15);
                    const int left = __popcll(todo);
                    n_run = n_run < left ? n_run : left;
                    for (int j = 0; j < n_run; ++j) {
                        const int src = (lane & 48) | ((rank + j) & 15);
#pragma unroll
                        for (int r = 0; r < 4; ++r) cur.v.v[r] *= __shfl(pre[r], src);
                        todo &= todo - 1;
                    }
                    rank += n_run;
                    continue;
                }
                const int l = __builtin_ctzll(todo);
                todo &= todo - 1;
                const int i = 64 * k + l;
                const uint32_t op = __builtin_amdgcn_readlane(k ? opw[1] : opw[0], l), meta = __builtin_amdgcn_readlane(k ? mt[1] : mt[0], l);   // selects: an index would send both arrays to scratch
                const unsigned long long ones_k = k ? ones_after[1] : ones_after[0];
                const bool ones_before = i > 0 && ((l > 0 ? ones_k >> (l - 1) : ones_after[0] >> 63) & 1ull);
                if (ones_before) {   // whatever cur holds is from before a run of skipped ops
                    cur.mask = 0xffffu;
                    cur.eig = false;
                }
                const int parking = static_cast<int>(meta >> 18);   // >= 1 for a pop
                const bool parked_ones = parking > 0 && (((parking - 1) >> 6 ? ones_after[1] : ones_after[0]) >> ((parking - 1) & 63) & 1ull);
                const bool stamp_op = stamping && p.stamp_classes != 0;
                const unsigned long long t_op = stamp_op ? fact_clock() : 0ull;
                run_op(op, static_cast<int>(meta & 0x3fffu), static_cast<int>((meta >> 14) & 0xfu), parked_ones, true_type{});
                if (stamp_op) {
                    const uint32_t cls = (op >> 28) == F_CHERRY_DEV ? 0u : ((op >> 28) == F_MM_LEAF_DEV ? 1u : ((op >> 28) < F_MM_POP0_DEV ? 2u : 3u));
                    if ((op >> 28) < F_ROOT_DEV) {   // selects, not an index (an indexed array would live in scratch memory in every build)
                        const unsigned long long dt = fact_clock() - t_op;
#pragma unroll
                        for (uint32_t k = 0; k < 4; ++k) {
                            t_class[k] += cls == k ? dt : 0ull;
                            n_class[k] += cls == k ? 1ull : 0ull;
                        }
                    }
                }
                ++rank;
            }
        }
    } else {
        // trees of more than 128 leaves: every op, slot and depth counted along the way
        int slot = 0, sp = 0;
        uint32_t op_lanes = 0;   // 64 ops at a time, one per lane: one LDS read per 64 ops, the op itself through v_readlane
        for (int ip = 0; ip < n_ops; ++ip) {
            if ((ip & 63) == 0) op_lanes = lprog[ip + lane];
            const uint32_t op = __builtin_amdgcn_readlane(op_lanes, ip & 63);
            const uint32_t code = op >> 28;
            const int slot_here = slot, depth_here = sp;
            if (code != F_CHERRY_DEV && code < F_ROOT_DEV) ++slot;
            if (code >= F_MM_PUSH_CHERRY0_DEV && code < F_MM_POP0_DEV) ++sp;
            else if (code >= F_MM_POP0_DEV && code < F_ROOT_DEV) --sp;
            run_op(op, slot_here, depth_here, false, false_type{});
        }
    }
    {   // F_ROOT: z = prior . alpha_root, ln z; the ancestral term sum_x ln(prior_x) prior_x alpha_x / z
        if (stamping && tid == 0) atomicAdd(p.stamps + 7, fact_clock() - t_begin);   // cycles until the root op
        const Vec4 cur_v = to_real(cur).v;
        const double* __restrict__ pr = p.prior + static_cast<size_t>(job.eig) * 64 + g * 16 + 4 * wave;  // lane order [g][q], q = 4 wave + r
        double z = 0.0, t = 0.0;
#pragma unroll
        for (int r = 0; r < 4; ++r) z += pr[r] * cur_v.v[r];
        z += __shfl_xor(z, 16);
        z += __shfl_xor(z, 32);
        if (ANC) {
            const double* __restrict__ aw = p.anc_weight + static_cast<size_t>(job.eig) * 64 + g * 16 + 4 * wave;
#pragma unroll
            for (int r = 0; r < 4; ++r) t += aw[r] * cur_v.v[r];
            t += __shfl_xor(t, 16);
            t += __shfl_xor(t, 32);
        }
        if (g == 0) {
            red[wave * 16 + sj] = z;
            if (ANC) red[64 + wave * 16 + sj] = t;
        }
        __syncthreads();
        if (wave == 0 && g == 0 && static_cast<uint32_t>(sj) < job.n_sites) {
            const double zz = ((red[sj] + red[16 + sj]) + red[32 + sj]) + red[48 + sj];
            p.out_lnz[job.out0 + sj] = log(zz);
            if (ANC) {
                const double tt = ((red[64 + sj] + red[80 + sj]) + red[96 + sj]) + red[112 + sj];
                p.out_anc[job.out0 + sj] = zz == 0.0 ? 0.0 : tt / zz;
            }
        }
    }
    if (p.product_count && tid == 0) atomicAdd(p.product_count, static_cast<unsigned long long>(nprod));
    if (stamping && tid == 0) {
        atomicAdd(p.stamps + 0, fact_clock() - t_begin);
        atomicAdd(p.stamps + 1, t_load);
        atomicAdd(p.stamps + 2, t_prod);
        atomicAdd(p.stamps + 3, 1ull);
        atomicAdd(p.stamps + 4, n_load);
        atomicAdd(p.stamps + 5, n_prod);
        atomicAdd(p.stamps + 6, t_prologue);
        for (int k = 0; k < 4; ++k) {
            atomicAdd(p.stamps + 8 + k, t_class[k]);
            atomicAdd(p.stamps + 12 + k, n_class[k]);
        }
    }
}

hipError_t launch_edge_scale(const ExpmEval* d_evals, uint32_t n_evals, uint32_t n_branches, const float* d_branch_len, const double* d_eig_right,
                             const double* d_eig_left, const double* d_eig_lambda, double* d_scale_out, uint32_t* d_fail, hipStream_t stream, bool dense_rows)
{
    if (n_evals == 0) return hipSuccess;
    hipLaunchKernelGGL(edge_scale_kernel, dim3(n_evals), dim3(256), 0, stream, d_evals, n_branches, d_branch_len, d_eig_right, d_eig_left,
                       d_eig_lambda, d_scale_out, d_fail, dense_rows ? 1u : 0u);
    return hipGetLastError();
}

hipError_t launch_compile_tiles(const FactoredParams& p, const TileDesc* d_tiles, uint32_t n_tiles, uint32_t* d_programs, uint32_t* d_site_tile, hipStream_t stream)
{
    if (n_tiles == 0) return hipSuccess;
    if (!p.op_meta || !p.op_leaves || p.n_ops > 128 || p.n_leaves > 128) return hipErrorInvalidValue;
    hipLaunchKernelGGL(compile_tiles_kernel, dim3(n_tiles), dim3(64), 0, stream, p, d_tiles, n_tiles, d_programs, d_site_tile);
    return hipGetLastError();
}

hipError_t launch_prune_factored(const FactoredParams& params, hipStream_t stream)
{
    if (params.n_jobs == 0) return hipSuccess;
    FactoredParams p = params;
    static const bool strided = [] { const char* v = getenv("PCSF_EIGEN_FRAG"); return v && v[0] == '0'; }();   // A/B: the operands from the row-major images
    if (strided && p.eig_right && p.eig_left) p.eig_right_frag = p.eig_left_frag = nullptr;   // only where the row-major images exist (the device engine's OMEGA lists hold none)
    if (!p.eig_right_frag && !p.eig_right) return hipErrorInvalidValue;
    const bool own = p.eval_x != nullptr;   // the kernel makes its own edge scales (needs eig_lambda, eig_c, branch_len, longest_branch, fail)
    const size_t lds = 2 * 4 * 2 * 64 * 16 + 2 * 4 * 16 * 8 + 64 * 8 + ((static_cast<size_t>(p.n_ops) + 63) & ~size_t(63)) * 4 +
                       ((static_cast<size_t>(p.n_edges) + 3) & ~size_t(3)) * 4 + 128 * 4 + static_cast<size_t>(p.n_leaves) * 17 + 128 +
                       16 + ((static_cast<size_t>(p.n_branches) + 3) & ~size_t(3)) * 4 + 4 * 64 * 8;
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    if (own && (!p.eig_lambda || !p.eig_c || !p.branch_len || !p.fail)) return hipErrorInvalidValue;
#define PCSF_LAUNCH_FACTORED(ANC, DEEP)                                                                                         \
    do {                                                                                                                        \
        if (own) hipLaunchKernelGGL((prune_factored<ANC, DEEP, true>), dim3(p.n_jobs), dim3(256), lds, stream, p);             \
        else hipLaunchKernelGGL((prune_factored<ANC, DEEP, false>), dim3(p.n_jobs), dim3(256), lds, stream, p);                \
    } while (0)
    if (p.deep) {
        if (p.out_anc) PCSF_LAUNCH_FACTORED(true, true);
        else PCSF_LAUNCH_FACTORED(false, true);
    } else {
        if (p.out_anc) PCSF_LAUNCH_FACTORED(true, false);
        else PCSF_LAUNCH_FACTORED(false, false);
    }
#undef PCSF_LAUNCH_FACTORED
    return hipGetLastError();
}

}  // namespace pcsf
