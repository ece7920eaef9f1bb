// kernels.h -- launch interface between the C-ABI layer (api.cpp) and the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace pcsf {

constexpr int FRAG_DOUBLES_DEV = 64 * 64;        // one branch matrix in fragment order
constexpr int LEAF_TABLE_DOUBLES_DEV = 65 * 64;  // one leaf-message table (65 codes x 64 states)
constexpr int MAX_STACK_DEPTH_DEV = 3;

// op codes of the pruning program (schedule.hpp keeps the host-side twin of this list)
constexpr uint32_t OP_CHERRY_DEV = 0, OP_MATMUL_DEV = 1, OP_MUL_LEAF_DEV = 2, OP_PUSH_DEV = 3, OP_MUL_POP_DEV = 4,
                   OP_ROOT_DEV = 5;

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
    double* out_score;       // [n_sites] or null
    double* out_lnz;         // [2][n_sites] or null
};

// variant: 0 = 4 waves x 1 tile, 1 = 8 waves x 1 tile, 2 = 4 waves x 2 tiles, 3 = 8 waves x 2 tiles
hipError_t launch_prune_fixed(const PruneParams& p, int variant, hipStream_t stream);
int prune_variant_count();

}  // namespace pcsf
