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
// the branch -- so a work-group stages both ONCE, as MFMA fragment images in LDS (64 KiB), and from then on its waves never
// synchronise again: each wave owns a 16-column tile, keeps the tile's message in the B-operand layout of v_mfma_f64_16x16x4_f64
// (16 registers per lane), and a product is 64 MFMAs whose A operands are conflict-free ds_read_b128 of the resident image.  The
// result registers are the B operands of the next product (kernels.hip, same layout), so a tile stays in registers from the
// leaves to the root; no matrix is formed, no table is written to or read from HBM, no barrier or LDS-DMA sits between products
// (prune_fixed in block mode re-stages a 32 KiB branch matrix per product; prune_factored exchanges the tile through LDS with a
// barrier per product and pays its prologue -- 64 KiB of eigenvectors into registers -- per 16 columns).  A work-group takes up to
// TILE_JOBS_PER_GROUP tiles that share an eigen-system: the tiles of one evaluation of a long block, or of several evaluations of
// the same block (rho stages), or of any blocks under the two ECM models (MLE).
//
// Leaf columns of L are gathered from the fragment image itself (16 x ds_read_b64 per lane and leaf); edge scales e_b come from
// edge_scale_kernel (factored.hip), 128 bytes per lane group and branch from L2.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "kernels.h"

namespace pcsf {

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

namespace {

struct V16 {
    double v[16];
};

// o <- M x b for the 64x16 tile b (B-operand layout: b[q] of lane l = row 4q + (l >> 4), column l & 15), M read from its LDS
// fragment image: 64 MFMAs in four accumulator chains, the two ds_read_b128 of the next k-step issued behind the first MFMA of
// this one (same issue order as kernels.hip product()).  o comes out in the same layout: o[4t + r] = acc[t][r].
__device__ __forceinline__ void tile_product(const double2_t* __restrict__ buf, int lane, const V16& b, V16& o)
{
    double4_t acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = double4_t{0.0, 0.0, 0.0, 0.0};
    double2_t n01 = buf[lane], n23 = buf[64 + lane];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const double2_t f01 = n01, f23 = n23;
        __builtin_amdgcn_sched_barrier(0);
        acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(f01.x, b.v[q], acc[0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (q + 1 < 16) {
            n01 = buf[((q + 1) * 2 + 0) * 64 + lane];
            n23 = buf[((q + 1) * 2 + 1) * 64 + lane];
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(f01.y, b.v[q], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(f23.x, b.v[q], acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(f23.y, b.v[q], acc[3], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) o.v[4 * t + r] = acc[t][r];
}

}  // namespace

constexpr int TILE_WAVES = 4;

// DEEP: the tree parks more than MAX_STACK_DEPTH_DEV tiles (none of the reference's trees does): parked tiles in scratch memory.
template <bool ANC, bool DEEP>
__global__ __launch_bounds__(TILE_WAVES * 64, 2) void prune_tiles(const FactoredParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_tiles[];
    double* Rf = reinterpret_cast<double*>(lds_tiles);                 // [4096] fragment image of R
    double* Lf = Rf + 4096;                                            // [4096] fragment image of L
    double* cgapL = Lf + 4096;                                         // [4][16] c = L 1 in lane order: [g][q] = c[4q + g]
    uint32_t* lprog = reinterpret_cast<uint32_t*>(cgapL + 64);         // [n_ops rounded up to 64]
    int32_t* lslot = reinterpret_cast<int32_t*>(lprog + ((p.n_ops + 63) & ~63u));   // [n_edges] slot -> branch
    const int cod_bytes = (p.n_leaves * 17 + 15) & ~15;                // per wave: [n_leaves][16] codons, [n_leaves] "has a codon"
    unsigned char* lcod_all = reinterpret_cast<unsigned char*>(lslot + ((p.n_edges + 3) & ~3u));

    const TileGroup grp = p.groups[blockIdx.x];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int sj = lane & 15, g = lane >> 4;
    const uint32_t eig = p.jobs[grp.job0].eig;

    // ---- R and L of the group's eigen-system into fragment order: frag[((q*2+h)*64 + l)*2 + e] = M[16(2h+e) + (l&15)][4q + (l>>4)].
    // Thread t takes row i = t >> 2, columns 16(t & 3) .. +15 (128 contiguous bytes of each matrix); the row sums of L (the eigen
    // coordinates of the all-ones message, a gap leaf) fall out of the same pass.
    {
        const int i = tid >> 2, cc = tid & 3;
        const double2_t* __restrict__ Rg = reinterpret_cast<const double2_t*>(p.eig_right + static_cast<size_t>(eig) * 4096 + i * 64 + 16 * cc);
        const double2_t* __restrict__ Lg = reinterpret_cast<const double2_t*>(p.eig_left + static_cast<size_t>(eig) * 4096 + i * 64 + 16 * cc);
        double2_t r[8], l[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            r[u] = Rg[u];
            l[u] = Lg[u];
        }
        const int row_part = ((i >> 5) * 64 + (i & 15)) * 2 + ((i >> 4) & 1);
        double rowsum = 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                const int j = 2 * u + w, q = 4 * cc + (j >> 2), gg = j & 3;
                const int idx = q * 256 + gg * 32 + row_part;
                Rf[idx] = w ? r[u].y : r[u].x;
                const double lv = w ? l[u].y : l[u].x;
                Lf[idx] = lv;
                rowsum += lv;
            }
        }
        rowsum += __shfl_xor(rowsum, 1);
        rowsum += __shfl_xor(rowsum, 2);
        if (cc == 0) cgapL[(i & 3) * 16 + (i >> 2)] = rowsum;
    }
    for (int i = tid; i < static_cast<int>((p.n_ops + 63) & ~63u); i += TILE_WAVES * 64) lprog[i] = p.program[i < p.n_ops ? i : p.n_ops - 1];
    for (int i = tid; i < p.n_edges; i += TILE_WAVES * 64) lslot[i] = p.slot_branch[i];
    __syncthreads();   // the only barrier: from here on the waves are independent

    const double2_t* __restrict__ Rbuf = reinterpret_cast<const double2_t*>(Rf);
    const double2_t* __restrict__ Lbuf = reinterpret_cast<const double2_t*>(Lf);
    unsigned char* lcod = lcod_all + wave * cod_bytes;
    unsigned char* lany = lcod + p.n_leaves * 16;
    const int n_ops = p.n_ops;

    for (uint32_t jx = wave; jx < grp.n_jobs; jx += TILE_WAVES) {
        const FactoredJob job = p.jobs[grp.job0 + jx];
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
        const double* __restrict__ scale = p.edge_scale + static_cast<size_t>(job.eval) * p.n_branches * 64 + 4 * g;

        // e_b in this lane's rows: e[q] = exp(lambda_k t_b), k = 4q + g; row-split position of k is 16(q >> 2) + 4g + (q & 3)
        auto load_scale = [&](int branch) __attribute__((always_inline)) -> V16 {
            const double2_t* s2 = reinterpret_cast<const double2_t*>(scale + static_cast<size_t>(branch) * 64);
            V16 e;
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
        // still to be formed.  `mask`: columns in which the whole subtree is gaps (their message is exactly 1); 0xffff = the all-ones tile.
        struct Msg {
            V16 v;
            uint32_t mask;
            bool eig;
        };
        auto ones = [&](uint32_t mask) __attribute__((always_inline)) -> Msg {
            Msg m;
#pragma unroll
            for (int q = 0; q < 16; ++q) m.v.v[q] = 1.0;
            m.mask = mask;
            m.eig = false;
            return m;
        };
        auto to_real = [&](const Msg& m) __attribute__((always_inline)) -> Msg {
            if (m.mask == 0xffffu) return ones(0xffffu);
            if (!m.eig) return m;
            Msg out;
            tile_product(Rbuf, lane, m.v, out.v);
            out.mask = m.mask;
            out.eig = false;
            const bool gap_col = (m.mask >> sj) & 1u;
#pragma unroll
            for (int q = 0; q < 16; ++q) out.v.v[q] = gap_col ? 1.0 : fmax(out.v.v[q], 0.0);
            return out;
        };
        auto apply_edge = [&](int slot_index, const Msg& m) __attribute__((always_inline)) -> Msg {
            if (m.mask == 0xffffu) return m;
            const V16 e = load_scale(lslot[slot_index]);
            Msg out;
            if (m.eig) out.v = m.v;
            else tile_product(Lbuf, lane, m.v, out.v);
            out.mask = m.mask;
            out.eig = true;
#pragma unroll
            for (int q = 0; q < 16; ++q) out.v.v[q] *= e.v[q];
            return out;
        };
        auto leaf = [&](int a) __attribute__((always_inline)) -> Msg {
            if (!leaf_has(a)) return ones(0xffffu);
            const unsigned code = lcod[a * 16 + sj];
            const uint32_t mask = static_cast<uint32_t>(__ballot(code == 64u)) & 0xffffu;
            const V16 e = load_scale(a);
            // column `code` of L, rows 4q + g, from the fragment image: element (k, c) sits at
            // (((c>>2)*2 + (k>>5))*64 + (c&3)*16 + (k&15))*2 + ((k>>4)&1); with k = 4q + g the q-dependent part is a constant
            const unsigned col = code == 64u ? 0u : code;
            const double* __restrict__ base = Lf + ((col >> 2) * 128 + (col & 3) * 16 + g) * 2;
            Msg out;
#pragma unroll
            for (int q = 0; q < 16; ++q) out.v.v[q] = base[(q >> 3) * 128 + (q & 3) * 8 + ((q >> 2) & 1)];
            if (mask != 0u) {
                const double* __restrict__ cg = cgapL + g * 16;
#pragma unroll
                for (int q = 0; q < 16; ++q) out.v.v[q] = code == 64u ? cg[q] : out.v.v[q];
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) out.v.v[q] *= e.v[q];
            out.mask = mask;
            out.eig = true;
            return out;
        };
        auto mul = [&](const Msg& x, const Msg& y) __attribute__((always_inline)) -> Msg {
            if (x.mask == 0xffffu) return y;
            if (y.mask == 0xffffu) return x;
            const Msg a = to_real(x);
            const Msg b = to_real(y);
            Msg out;
#pragma unroll
            for (int q = 0; q < 16; ++q) out.v.v[q] = a.v.v[q] * b.v.v[q];
            out.mask = a.mask & b.mask;
            out.eig = false;
            return out;
        };

        // parked messages: pushes and pops nest; an all-ones tile is never stored or loaded
        V16 stk[DEEP ? MAX_STACK_DEPTH_DEEP_DEV : 1], s0, s1, s2;
        uint32_t stk_mask[DEEP ? MAX_STACK_DEPTH_DEEP_DEV : 1];
#pragma unroll
        for (int q = 0; q < 16; ++q) s0.v[q] = s1.v[q] = s2.v[q] = 1.0;
        unsigned long long packed_masks = 0;   // shallow trees: 16 bits per parked message
        uint32_t stk_eig = 0;                  // bit d: parked message d is in eigen form
        auto park = [&](int d, const Msg& m) __attribute__((always_inline)) {
            if (DEEP) stk_mask[d] = m.mask;
            else packed_masks = (packed_masks & ~(0xffffull << (16 * d))) | (static_cast<unsigned long long>(m.mask) << (16 * d));
            stk_eig = (stk_eig & ~(1u << d)) | (m.eig ? 1u << d : 0u);
            if (m.mask == 0xffffu) return;
            if (DEEP) {
                stk[d] = m.v;
            } else {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    s0.v[q] = d == 0 ? m.v.v[q] : s0.v[q];
                    s1.v[q] = d == 1 ? m.v.v[q] : s1.v[q];
                    s2.v[q] = d == 2 ? m.v.v[q] : s2.v[q];
                }
            }
        };
        auto parked = [&](int d) __attribute__((always_inline)) -> Msg {
            Msg m = ones(DEEP ? stk_mask[d] : static_cast<uint32_t>((packed_masks >> (16 * d)) & 0xffffull));
            m.eig = ((stk_eig >> d) & 1u) != 0;
            if (m.mask == 0xffffu) return m;
            if (DEEP) {
                m.v = stk[d];
            } else {
#pragma unroll
                for (int q = 0; q < 16; ++q) m.v.v[q] = d == 0 ? s0.v[q] : (d == 1 ? s1.v[q] : s2.v[q]);
            }
            return m;
        };
        Msg cur = ones(0xffffu);
        auto run_op = [&](uint32_t op, int slot, int depth, bool parked_ones) __attribute__((always_inline)) {
            const uint32_t code = op >> 28;
            const int a = (op >> 14) & 0x3fff, b = op & 0x3fff;
            if (code == F_CHERRY_DEV) {
                cur = mul(leaf(a), leaf(b));
            } else if (code == F_MM_LEAF_DEV) {
                cur = mul(apply_edge(slot, cur), leaf(a));
            } else if (code >= F_MM_PUSH_CHERRY0_DEV && code < F_MM_POP0_DEV) {
                park(depth, apply_edge(slot, cur));
                cur = mul(leaf(a), leaf(b));
            } else if (code >= F_MM_POP0_DEV && code < F_ROOT_DEV) {
                const Msg m = apply_edge(slot, cur);
                if (parked_ones) cur = m;
                else cur = mul(parked(depth - 1), m);
            } else {  // F_ROOT: z = prior . alpha_root, ln z; the ancestral term sum_x ln(prior_x) prior_x alpha_x / z
                const Msg r = to_real(cur);
                const double* __restrict__ pr = p.prior + static_cast<size_t>(job.eig) * 64 + g * 16;   // lane order [g][q]
                double z = 0.0, t = 0.0;
#pragma unroll
                for (int q = 0; q < 16; ++q) z += pr[q] * r.v.v[q];
                z += __shfl_xor(z, 16);
                z += __shfl_xor(z, 32);
                if (ANC) {
                    const double* __restrict__ aw = p.anc_weight + static_cast<size_t>(job.eig) * 64 + g * 16;
#pragma unroll
                    for (int q = 0; q < 16; ++q) t += aw[q] * r.v.v[q];
                    t += __shfl_xor(t, 16);
                    t += __shfl_xor(t, 32);
                }
                if (g == 0 && static_cast<uint32_t>(sj) < job.n_sites) {
                    p.out_lnz[job.out0 + sj] = log(z);
                    if (ANC) p.out_anc[job.out0 + sj] = z == 0.0 ? 0.0 : t / z;
                }
            }
        };

        if (p.op_meta != nullptr) {
            // ops of all-gap subtrees are skipped (factored.hip): schedule.hpp `program_facts` knows every op's leaf set, start slot,
            // stack depth and, for a pop, the parking op; with the tile's leaf bits a lane-parallel pass marks the ops whose result is
            // the all-ones tile, and the loop visits only the others (trees of at most 128 leaves: n_ops <= 128)
            uint32_t opw[2], mt[2];
            unsigned long long ones_after[2], active[2];
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
            for (int k = 0; k < 2; ++k) {
                unsigned long long todo = active[k];
                while (todo) {
                    const int l = __builtin_ctzll(todo);
                    todo &= todo - 1;
                    const int i = 64 * k + l;
                    const uint32_t op = __builtin_amdgcn_readlane(opw[k], l), meta = __builtin_amdgcn_readlane(mt[k], l);
                    const bool ones_before = i > 0 && ((l > 0 ? ones_after[k] >> (l - 1) : ones_after[0] >> 63) & 1ull);
                    if (ones_before) {   // whatever cur holds is from before a run of skipped ops
                        cur.mask = 0xffffu;
                        cur.eig = false;
                    }
                    const int parking = static_cast<int>(meta >> 18);   // >= 1 for a pop
                    const bool parked_ones = parking > 0 && ((ones_after[(parking - 1) >> 6] >> ((parking - 1) & 63)) & 1ull);
                    run_op(op, static_cast<int>(meta & 0x3fffu), static_cast<int>((meta >> 14) & 0xfu), parked_ones);
                }
            }
        } else {
            // trees of more than 128 leaves: every op, slot and depth counted along the way
            int slot = 0, sp = 0;
            uint32_t op_lanes = 0;
            for (int ip = 0; ip < n_ops; ++ip) {
                if ((ip & 63) == 0) op_lanes = lprog[ip + lane];
                const uint32_t op = __builtin_amdgcn_readlane(op_lanes, ip & 63);
                const uint32_t code = op >> 28;
                const int slot_here = slot, depth_here = sp;
                if (code != F_CHERRY_DEV && code < F_ROOT_DEV) ++slot;
                if (code >= F_MM_PUSH_CHERRY0_DEV && code < F_MM_POP0_DEV) ++sp;
                else if (code >= F_MM_POP0_DEV && code < F_ROOT_DEV) --sp;
                run_op(op, slot_here, depth_here, false);
            }
        }
        __builtin_amdgcn_wave_barrier();   // the next job rewrites this wave's codon rows
    }
}

size_t prune_tiles_lds_bytes(const FactoredParams& p)
{
    return 2 * 4096 * sizeof(double) + 64 * sizeof(double) + ((static_cast<size_t>(p.n_ops) + 63) & ~size_t(63)) * 4 +
           ((static_cast<size_t>(p.n_edges) + 3) & ~size_t(3)) * 4 + TILE_WAVES * ((static_cast<size_t>(p.n_leaves) * 17 + 15) & ~size_t(15));
}

template <bool ANC, bool DEEP>
static hipError_t launch_tiles_impl(const FactoredParams& p, size_t lds, hipStream_t stream)
{
    static std::atomic<uint64_t> attr_set{0};   // per (instantiation, device), like kernels.hip launch_impl
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    if (!(attr_set.load(std::memory_order_acquire) >> (dev & 63) & 1)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&prune_tiles<ANC, DEEP>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set.fetch_or(uint64_t(1) << (dev & 63), std::memory_order_release);
    }
    hipLaunchKernelGGL((prune_tiles<ANC, DEEP>), dim3(p.n_groups), dim3(TILE_WAVES * 64), lds, stream, p);
    return hipGetLastError();
}

hipError_t launch_prune_tiles(const FactoredParams& p, hipStream_t stream)
{
    if (p.n_groups == 0) return hipSuccess;
    if (!p.groups) return hipErrorInvalidValue;
    const size_t lds = prune_tiles_lds_bytes(p);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (p.deep) return p.out_anc ? launch_tiles_impl<true, true>(p, lds, stream) : launch_tiles_impl<false, true>(p, lds, stream);
    return p.out_anc ? launch_tiles_impl<true, false>(p, lds, stream) : launch_tiles_impl<false, false>(p, lds, stream);
}

}  // namespace pcsf
