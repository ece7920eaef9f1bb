// kernels.h -- launch interface between the C-ABI layer (api.cpp) and the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace pcsf {

constexpr int FRAG_DOUBLES_DEV = 64 * 64;        // one branch matrix in fragment order
constexpr int LEAF_TABLE_DOUBLES_DEV = 65 * 64;  // one leaf-message table (65 codes x 64 states)
constexpr int MAX_STACK_DEPTH_DEV = 3;

// fused op codes of the pruning program (schedule.hpp FusedOp is the host-side twin of this list)
constexpr uint32_t F_CHERRY_DEV = 0, F_MM_LEAF_DEV = 1, F_MM_PUSH_CHERRY0_DEV = 2, F_MM_POP0_DEV = 5, F_ROOT_DEV = 8;

struct PruneParams {
    const uint8_t* codons;   // [n_leaves][leaf_stride], 0..63 or 64 = marginalise
    size_t n_sites;
    size_t leaf_stride;
    const uint32_t* program; // [n_ops]
    int n_ops;
    int n_edges;             // MATMUL ops per model (= internal non-root nodes)
    int n_leaves;
    const double* frag;      // [2][n_edges][4096]  branch matrices of internal children, fragment order
    const double* leaf_tab;  // [2][n_leaves][65][4][16] leaf-message tables
    const double* prior;     // [2][4][16] equilibrium prior in lane order
    unsigned long long* stamps;  // diagnostic variant only: [2][n_ops][4] s_memtime values, else null
    uint32_t stamp_block;
    double* out_score;       // [n_sites] or null
    double* out_lnz;         // [2][n_sites] or null
};

// variant: 0 = 4 waves (64 sites) per work-group, 1 = 8 waves (128 sites) per work-group,
// 2 = variant 0 with in-kernel stamps (diagnostic only)
hipError_t launch_prune_fixed(const PruneParams& p, int variant, hipStream_t stream, size_t extra_lds = 0);
int prune_variant_count();

}  // namespace pcsf
