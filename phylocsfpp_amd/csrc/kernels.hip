// kernels.hip -- gfx950 (CDNA4) kernels of the PhyloCSF scoring core.
//
// prune_fixed<NW,G>: Felsenstein pruning + log-likelihood ratio for the FIXED strategy
// (reference: ensure_alpha / lpr_leaves / run_tracks, src/fixed_lik.hpp:125-164,362-449,
// src/run.hpp:35-55).  One wave owns G tiles of 16 alignment columns ("sites"); the partial
// likelihoods of a tile (64 states x 16 sites, fp64) live in 16 registers per lane in the
// operand layout of v_mfma_f64_16x16x4_f64, so a node's result feeds the next product without
// leaving the register file.  The 64x64 branch matrix of the edge being processed is staged
// once per work-group in LDS (fragment order, double-buffered); leaf edges are table look-ups
// in HBM/L2 (one 128-byte row segment per lane).  See DESIGN.md for layouts and the roofline.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace pcsf {

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double4_t mfma_f64(double a, double b, double4_t c)
{
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// One lane's 128-byte segment of a leaf-message row: table[leaf][code][g4][0..15]
struct LeafRow {
    double2_t v[8];
};

__device__ __forceinline__ LeafRow load_leaf_row(const double* __restrict__ leaf_tab, int leaf, uint32_t code, int g4)
{
    const double2_t* r2 = reinterpret_cast<const double2_t*>(leaf_tab + ((static_cast<size_t>(leaf) * 65 + code) * 4 + g4) * 16);
    LeafRow r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = r2[i];
    return r;
}

__device__ __forceinline__ const double2_t* leaf_row_ptr(const double* __restrict__ leaf_tab, int leaf, uint32_t code, int g4)
{
    return reinterpret_cast<const double2_t*>(leaf_tab + ((static_cast<size_t>(leaf) * 65 + code) * 4 + g4) * 16);
}

// acc[t] (t = 0..3, output states 16t..16t+15) <- P x cur, with P read from its LDS fragment image:
// 64 x v_mfma_f64_16x16x4_f64 in four independent accumulator chains.
// Fragments are fetched one k-step ahead of the MFMAs that use them; the scheduling barriers keep
// the compiler from hoisting all 32 LDS reads (128 VGPRs) to the top.
// CHERRY: while the product runs, the next subtree's first tile nxt = leafmsg(a)*leafmsg(b) is
// built a quarter row at a time (loads issued 4 k-steps = 16 MFMAs ahead of their use), so the
// two 128-byte rows never occupy registers in full.
template <bool CHERRY>
__device__ __forceinline__ void product(const double2_t* __restrict__ buf, int lane, const double (&cur)[16], double4_t (&acc)[4],
                                        const double2_t* __restrict__ rowa, const double2_t* __restrict__ rowb, double (&nxt)[16])
{
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = double4_t{0.0, 0.0, 0.0, 0.0};
    double2_t n01 = buf[lane], n23 = buf[64 + lane];
    double2_t xa0, xa1, xb0, xb1;
    if (CHERRY) {
        xa0 = rowa[0]; xa1 = rowa[1];
        xb0 = rowb[0]; xb1 = rowb[1];
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const double2_t f01 = n01, f23 = n23;
        if (q + 1 < 16) {
            n01 = buf[((q + 1) * 2 + 0) * 64 + lane];
            n23 = buf[((q + 1) * 2 + 1) * 64 + lane];
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[0] = mfma_f64(f01.x, cur[q], acc[0]);
        acc[1] = mfma_f64(f01.y, cur[q], acc[1]);
        acc[2] = mfma_f64(f23.x, cur[q], acc[2]);
        acc[3] = mfma_f64(f23.y, cur[q], acc[3]);
        if (CHERRY && (q & 3) == 3) {
            const int k = q >> 2;
            nxt[4 * k + 0] = xa0.x * xb0.x;
            nxt[4 * k + 1] = xa0.y * xb0.y;
            nxt[4 * k + 2] = xa1.x * xb1.x;
            nxt[4 * k + 3] = xa1.y * xb1.y;
            if (k < 3) {
                xa0 = rowa[2 * k + 2]; xa1 = rowa[2 * k + 3];
                xb0 = rowb[2 * k + 2]; xb1 = rowb[2 * k + 3];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

// prune_fixed<NW>: NW waves per work-group, 16 sites per wave.
// LDS (dynamic, one array): [2][4096] doubles fragment double-buffer, then [n_leaves][NW*16] codon bytes.
template <int NW>
__global__ __launch_bounds__(NW * 64, 2) void prune_fixed(const PruneParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    double2_t* lfrag = reinterpret_cast<double2_t*>(lds_raw);
    unsigned char* lcod = lds_raw + 2 * FRAG_DOUBLES_DEV * sizeof(double);

    constexpr int SITES = NW * 16;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int sl = lane & 15, g4 = lane >> 4;
    const size_t block_site0 = static_cast<size_t>(blockIdx.x) * SITES;
    const size_t my_site = block_site0 + wave * 16 + sl;
    const bool valid = my_site < p.n_sites;

    // ---- stage this work-group's codon columns into LDS (clamped to 0..64; 64 beyond the end)
    for (int idx = tid; idx < p.n_leaves * SITES; idx += NW * 64) {
        const int leaf = idx / SITES, s = idx - leaf * SITES;
        const size_t site = block_site0 + s;
        unsigned c = 64u;
        if (site < p.n_sites) {
            c = p.codons[static_cast<size_t>(leaf) * p.leaf_stride + site];
            c = c > 64u ? 64u : c;
        }
        lcod[idx] = static_cast<unsigned char>(c);
    }
    const unsigned char* my_cod = lcod + wave * 16 + sl;  // + leaf*SITES

    // ---- fragment staging: 32 KiB per slot, moved by LDS-DMA in 1 KiB wave-instructions
    constexpr int CHUNKS_PER_WAVE = 32 / NW;
    const int total_slots = 2 * p.n_edges;  // coding slots, then non-coding slots
    auto stage_slot = [&](int slot) {
        const unsigned char* src = reinterpret_cast<const unsigned char*>(p.frag) + static_cast<size_t>(slot) * (FRAG_DOUBLES_DEV * 8);
        unsigned char* dst = lds_raw + (slot & 1) * (FRAG_DOUBLES_DEV * 8);
#pragma unroll
        for (int j = 0; j < CHUNKS_PER_WAVE; ++j) {
            const int chunk = wave * CHUNKS_PER_WAVE + j;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + chunk * 1024 + lane * 16),
                                             (__attribute__((address_space(3))) void*)(dst + chunk * 1024), 16, 0, 0);
        }
    };
    int slot = 0;
    if (total_slots > 0) stage_slot(0);

    double lnz0 = 0.0, lnz1 = 0.0;

    for (int model = 0; model < 2; ++model) {
        const double* __restrict__ leaf_tab = p.leaf_tab + static_cast<size_t>(model) * p.n_leaves * LEAF_TABLE_DOUBLES_DEV;
        const double* __restrict__ prior = p.prior + model * 64 + g4 * 16;

        double cur[16], nxt[16];
        double4_t acc[4];
        double stk[MAX_STACK_DEPTH_DEV][16];

        for (int ip = 0; ip < p.n_ops; ++ip) {
            const uint32_t op = p.program[ip];
            const uint32_t code = op >> 28;
            const int a = (op >> 14) & 0x3fff, b = op & 0x3fff;

            // Every op except the first cherry and the root starts with a product.  Its barrier
            // (a) completes this wave's LDS-DMA of the current slot (vmcnt(0)), (b) publishes all
            // waves' parts, (c) retires all reads of the other buffer, which the next slot's DMA
            // overwrites.  Leaf rows needed after the product are requested before the MFMAs.
#define PCSF_PRODUCT_PROLOGUE()                                                      \
    __syncthreads();                                                                 \
    const double2_t* buf = lfrag + (slot & 1) * (FRAG_DOUBLES_DEV / 2);
#define PCSF_PRODUCT_BODY(CH, RA, RB)                                                \
    if (slot + 1 < total_slots) stage_slot(slot + 1);                                \
    ++slot;                                                                          \
    product<CH>(buf, lane, cur, acc, RA, RB, nxt);
#define PCSF_PUSH_CHERRY(D)                                                          \
    {                                                                                \
        PCSF_PRODUCT_PROLOGUE()                                                      \
        const double2_t* pa = leaf_row_ptr(leaf_tab, a, my_cod[a * SITES], g4);      \
        const double2_t* pb = leaf_row_ptr(leaf_tab, b, my_cod[b * SITES], g4);      \
        PCSF_PRODUCT_BODY(true, pa, pb)                                              \
        _Pragma("unroll") for (int q = 0; q < 16; ++q) {                             \
            stk[D][q] = acc[q >> 2][q & 3];                                          \
            cur[q] = nxt[q];                                                         \
        }                                                                            \
    }
#define PCSF_POP(D)                                                                  \
    {                                                                                \
        PCSF_PRODUCT_PROLOGUE()                                                      \
        PCSF_PRODUCT_BODY(false, nullptr, nullptr)                                   \
        _Pragma("unroll") for (int q = 0; q < 16; ++q) cur[q] = stk[D][q] * acc[q >> 2][q & 3]; \
    }
            switch (code) {
                case F_CHERRY_DEV: {
                    __syncthreads();  // codon bytes staged above are visible
                    const LeafRow ra = load_leaf_row(leaf_tab, a, my_cod[a * SITES], g4);
                    const LeafRow rb = load_leaf_row(leaf_tab, b, my_cod[b * SITES], g4);
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        cur[2 * i] = ra.v[i].x * rb.v[i].x;
                        cur[2 * i + 1] = ra.v[i].y * rb.v[i].y;
                    }
                    break;
                }
                case F_MM_LEAF_DEV: {
                    PCSF_PRODUCT_PROLOGUE()
                    const LeafRow ra = load_leaf_row(leaf_tab, a, my_cod[a * SITES], g4);
                    PCSF_PRODUCT_BODY(false, nullptr, nullptr)
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        cur[2 * i] = acc[(2 * i) >> 2][(2 * i) & 3] * ra.v[i].x;
                        cur[2 * i + 1] = acc[(2 * i + 1) >> 2][(2 * i + 1) & 3] * ra.v[i].y;
                    }
                    break;
                }
                case F_MM_PUSH_CHERRY0_DEV + 0: PCSF_PUSH_CHERRY(0) break;
                case F_MM_PUSH_CHERRY0_DEV + 1: PCSF_PUSH_CHERRY(1) break;
                case F_MM_PUSH_CHERRY0_DEV + 2: PCSF_PUSH_CHERRY(2) break;
                case F_MM_POP0_DEV + 0: PCSF_POP(0) break;
                case F_MM_POP0_DEV + 1: PCSF_POP(1) break;
                case F_MM_POP0_DEV + 2: PCSF_POP(2) break;
                default: {  // F_ROOT
                    double z = 0.0;
#pragma unroll
                    for (int q = 0; q < 16; ++q) z += prior[q] * cur[q];
                    z += __shfl_xor(z, 16);
                    z += __shfl_xor(z, 32);
                    const double lz = log(z);
                    if (model == 0) lnz0 = lz;
                    else lnz1 = lz;
                    break;
                }
            }
#undef PCSF_PRODUCT_PROLOGUE
#undef PCSF_PRODUCT_BODY
#undef PCSF_PUSH_CHERRY
#undef PCSF_POP
        }
    }

    if (g4 == 0 && valid) {
        if (p.out_score) p.out_score[my_site] = 10.0 * (lnz0 - lnz1) / log(10.0);
        if (p.out_lnz) {
            p.out_lnz[my_site] = lnz0;
            p.out_lnz[p.n_sites + my_site] = lnz1;
        }
    }
}

template <int NW>
static hipError_t launch_impl(const PruneParams& p, hipStream_t stream)
{
    const size_t sites_per_block = static_cast<size_t>(NW) * 16;
    const size_t blocks = (p.n_sites + sites_per_block - 1) / sites_per_block;
    if (blocks == 0) return hipSuccess;
    if (blocks > 0x7fffffffULL) return hipErrorInvalidValue;
    const size_t lds = 2 * FRAG_DOUBLES_DEV * sizeof(double) + static_cast<size_t>(p.n_leaves) * sites_per_block;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&prune_fixed<NW>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((prune_fixed<NW>), dim3(static_cast<unsigned>(blocks)), dim3(NW * 64), lds, stream, p);
    return hipGetLastError();
}

hipError_t launch_prune_fixed(const PruneParams& p, int variant, hipStream_t stream)
{
    switch (variant) {
        case 0: return launch_impl<4>(p, stream);
        case 1: return launch_impl<8>(p, stream);
        default: return hipErrorInvalidValue;
    }
}

int prune_variant_count() { return 2; }

}  // namespace pcsf
