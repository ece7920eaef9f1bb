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

// 16 doubles of one lane's leaf-message row segment
__device__ __forceinline__ void load_leaf_row(const double* __restrict__ row, double (&t)[16])
{
    const double2_t* r2 = reinterpret_cast<const double2_t*>(row);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const double2_t v = r2[i];
        t[2 * i] = v.x;
        t[2 * i + 1] = v.y;
    }
}

template <int NW, int G>
__global__ __launch_bounds__(NW * 64) void prune_fixed(const PruneParams p)
{
    __shared__ __attribute__((aligned(16))) double lds[2 * FRAG_DOUBLES_DEV];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int sl = lane & 15, g4 = lane >> 4;
    const size_t tile0 = (static_cast<size_t>(blockIdx.x) * NW + wave) * (16 * G);

    size_t site[G];
    bool valid[G];
#pragma unroll
    for (int gi = 0; gi < G; ++gi) {
        const size_t s = tile0 + gi * 16 + sl;
        valid[gi] = s < p.n_sites;
        site[gi] = valid[gi] ? s : 0;
    }

    // cooperative staging of one 32 KiB fragment image: every thread moves PER_THREAD 16-byte pieces
    constexpr int PIECES = FRAG_DOUBLES_DEV / 2;           // 2048 double2
    constexpr int PER_THREAD = PIECES / (NW * 64);
    const int total_slots = 2 * p.n_edges;                 // coding slots, then non-coding slots
    const double2_t* gfrag = reinterpret_cast<const double2_t*>(p.frag);
    double2_t* lfrag = reinterpret_cast<double2_t*>(lds);
    double2_t stage[PER_THREAD];

    int slot = 0;  // next MATMUL's slot (runs through both models)
    if (total_slots > 0) {
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) stage[i] = gfrag[i * (NW * 64) + tid];
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) lfrag[i * (NW * 64) + tid] = stage[i];
    }

    double lnz0[G], lnz1[G];

    for (int model = 0; model < 2; ++model) {
        const double* __restrict__ leaf_tab = p.leaf_tab + static_cast<size_t>(model) * p.n_leaves * LEAF_TABLE_DOUBLES_DEV;
        const double* __restrict__ prior = p.prior + model * 64 + g4 * 16;

        double cur[G][16];
        double4_t acc[G][4];
        double stk[MAX_STACK_DEPTH_DEV][G][16];

        for (int ip = 0; ip < p.n_ops; ++ip) {
            const uint32_t op = p.program[ip];
            const uint32_t code = op >> 28;
            const int a = (op >> 14) & 0x3fff, b = op & 0x3fff;
            switch (code) {
                case OP_CHERRY_DEV: {
#pragma unroll
                    for (int gi = 0; gi < G; ++gi) {
                        uint32_t ca = valid[gi] ? p.codons[static_cast<size_t>(a) * p.leaf_stride + site[gi]] : 64u;
                        uint32_t cb = valid[gi] ? p.codons[static_cast<size_t>(b) * p.leaf_stride + site[gi]] : 64u;
                        ca = ca > 64u ? 64u : ca;
                        cb = cb > 64u ? 64u : cb;
                        double ta[16], tb[16];
                        load_leaf_row(leaf_tab + ((static_cast<size_t>(a) * 65 + ca) * 4 + g4) * 16, ta);
                        load_leaf_row(leaf_tab + ((static_cast<size_t>(b) * 65 + cb) * 4 + g4) * 16, tb);
#pragma unroll
                        for (int q = 0; q < 16; ++q) cur[gi][q] = ta[q] * tb[q];
                    }
                    break;
                }
                case OP_MATMUL_DEV: {
                    // buffer (slot&1) was filled one product ago; the barrier publishes it and
                    // retires every wave's reads of the other buffer
                    __syncthreads();
                    const bool more = slot + 1 < total_slots;
                    if (more) {
                        const double2_t* src = gfrag + static_cast<size_t>(slot + 1) * PIECES;
#pragma unroll
                        for (int i = 0; i < PER_THREAD; ++i) stage[i] = src[i * (NW * 64) + tid];
                    }
                    const double2_t* buf = lfrag + (slot & 1) * PIECES;
#pragma unroll
                    for (int gi = 0; gi < G; ++gi)
#pragma unroll
                        for (int t = 0; t < 4; ++t) acc[gi][t] = double4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const double2_t f01 = buf[(q * 2 + 0) * 64 + lane];
                        const double2_t f23 = buf[(q * 2 + 1) * 64 + lane];
#pragma unroll
                        for (int gi = 0; gi < G; ++gi) {
                            acc[gi][0] = mfma_f64(f01.x, cur[gi][q], acc[gi][0]);
                            acc[gi][1] = mfma_f64(f01.y, cur[gi][q], acc[gi][1]);
                            acc[gi][2] = mfma_f64(f23.x, cur[gi][q], acc[gi][2]);
                            acc[gi][3] = mfma_f64(f23.y, cur[gi][q], acc[gi][3]);
                        }
                    }
                    if (more) {
                        double2_t* dst = lfrag + ((slot + 1) & 1) * PIECES;
#pragma unroll
                        for (int i = 0; i < PER_THREAD; ++i) dst[i * (NW * 64) + tid] = stage[i];
                    }
                    ++slot;
                    break;
                }
                case OP_MUL_LEAF_DEV: {
#pragma unroll
                    for (int gi = 0; gi < G; ++gi) {
                        uint32_t ca = valid[gi] ? p.codons[static_cast<size_t>(a) * p.leaf_stride + site[gi]] : 64u;
                        ca = ca > 64u ? 64u : ca;
                        double ta[16];
                        load_leaf_row(leaf_tab + ((static_cast<size_t>(a) * 65 + ca) * 4 + g4) * 16, ta);
#pragma unroll
                        for (int t = 0; t < 4; ++t)
#pragma unroll
                            for (int r = 0; r < 4; ++r) cur[gi][4 * t + r] = acc[gi][t][r] * ta[4 * t + r];
                    }
                    break;
                }
                case OP_PUSH_DEV: {
#pragma unroll
                    for (int d = 0; d < MAX_STACK_DEPTH_DEV; ++d)
                        if (a == d) {
#pragma unroll
                            for (int gi = 0; gi < G; ++gi)
#pragma unroll
                                for (int t = 0; t < 4; ++t)
#pragma unroll
                                    for (int r = 0; r < 4; ++r) stk[d][gi][4 * t + r] = acc[gi][t][r];
                        }
                    break;
                }
                case OP_MUL_POP_DEV: {
#pragma unroll
                    for (int d = 0; d < MAX_STACK_DEPTH_DEV; ++d)
                        if (a == d) {
#pragma unroll
                            for (int gi = 0; gi < G; ++gi)
#pragma unroll
                                for (int t = 0; t < 4; ++t)
#pragma unroll
                                    for (int r = 0; r < 4; ++r) cur[gi][4 * t + r] = stk[d][gi][4 * t + r] * acc[gi][t][r];
                        }
                    break;
                }
                default: {  // OP_ROOT
#pragma unroll
                    for (int gi = 0; gi < G; ++gi) {
                        double z = 0.0;
#pragma unroll
                        for (int q = 0; q < 16; ++q) z += prior[q] * cur[gi][q];
                        z += __shfl_xor(z, 16);
                        z += __shfl_xor(z, 32);
                        const double lz = log(z);
                        if (model == 0) lnz0[gi] = lz;
                        else lnz1[gi] = lz;
                    }
                    break;
                }
            }
        }
    }

    if (g4 == 0) {
#pragma unroll
        for (int gi = 0; gi < G; ++gi)
            if (valid[gi]) {
                if (p.out_score) p.out_score[site[gi]] = 10.0 * (lnz0[gi] - lnz1[gi]) / log(10.0);
                if (p.out_lnz) {
                    p.out_lnz[site[gi]] = lnz0[gi];
                    p.out_lnz[p.n_sites + site[gi]] = lnz1[gi];
                }
            }
    }
}

template <int NW, int G>
static hipError_t launch_impl(const PruneParams& p, hipStream_t stream)
{
    const size_t sites_per_block = static_cast<size_t>(NW) * 16 * G;
    const size_t blocks = (p.n_sites + sites_per_block - 1) / sites_per_block;
    if (blocks == 0) return hipSuccess;
    if (blocks > 0x7fffffffULL) return hipErrorInvalidValue;
    hipLaunchKernelGGL((prune_fixed<NW, G>), dim3(static_cast<unsigned>(blocks)), dim3(NW * 64), 0, stream, p);
    return hipGetLastError();
}

hipError_t launch_prune_fixed(const PruneParams& p, int variant, hipStream_t stream)
{
    switch (variant) {
        case 0: return launch_impl<4, 1>(p, stream);
        case 1: return launch_impl<8, 1>(p, stream);
        case 2: return launch_impl<4, 2>(p, stream);
        case 3: return launch_impl<8, 2>(p, stream);
        default: return hipErrorInvalidValue;
    }
}

int prune_variant_count() { return 4; }

}  // namespace pcsf
