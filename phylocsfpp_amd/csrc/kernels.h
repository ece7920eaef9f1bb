// kernels.h -- launch interface between the C-ABI layer (api.cpp) and the gfx950 kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace pcsf {

constexpr int FRAG_DOUBLES_DEV = 64 * 64;        // one branch matrix in fragment order
constexpr int LEAF_TABLE_DOUBLES_DEV = 65 * 64;  // one leaf-message table (65 codes x 64 states)
constexpr int MAX_STACK_DEPTH_DEV = 3;        // parked tiles in registers
constexpr int MAX_STACK_DEPTH_DEEP_DEV = 14;  // DEEP variants: parked tiles in scratch memory (schedule.hpp)

// fused op codes of the pruning program (schedule.hpp FusedOp is the host-side twin of this list)
constexpr uint32_t F_CHERRY_DEV = 0, F_MM_LEAF_DEV = 1, F_MM_PUSH_CHERRY0_DEV = 2, F_MM_POP0_DEV = 5, F_ROOT_DEV = 8;

struct FrameSegment;
struct PruneParams {
    const uint8_t* codons;   // [n_leaves][leaf_stride], 0..63 or 64 = marginalise (not read by the fused variant)
    // fused six-frame front end (variant 32): the sites are the codons of `segments` laid end to end
    const unsigned char* nucl;        // [n_leaves][nucl_stride] ASCII rows
    size_t nucl_stride;
    const FrameSegment* segments;     // in site order, none empty
    const uint32_t* wg_first_seg;     // [work-groups] segment that holds the work-group's first site
    unsigned long long* first_bad;    // atomicMin of row * nucl_stride + col over characters outside ACGTNacgtn-. (may be null)
    unsigned char* dump_codons;       // diagnostics (may be null): the codons the fused front end formed, [n_leaves][leaf_stride]
    size_t n_sites;
    size_t leaf_stride;
    const uint32_t* program; // [n_ops]
    const uint64_t* op_leaves;   // [n_ops][2] and
    const uint32_t* op_meta;     // [n_ops]: schedule.hpp ProgramFacts of `program` (variants + 128; null for trees of more than 128 leaves)
    int n_ops;
    int n_edges;             // MATMUL ops per model (= internal non-root nodes)
    int n_leaves;
    int n_passes;            // table sets walked per site: 2 = coding then non-coding (tracks / FIXED), 1 = one model
    const double* frag;      // [table][n_edges][4096]  branch matrices of internal children, fragment order
    const double* leaf_tab;  // [table][n_leaves][65 codes][8 q2][4 g][2 e] leaf-message tables
    const double* prior;     // [table][4][16] equilibrium prior in lane order
    const double* anc_weight;// [table][4][16] ln(prior)*prior in lane order (only read if out_anc)
    // block mode (variant 16): work-group w scores sites [wg_site0[w], wg_site_end[w]) (at most 64)
    // with table sets wg_table[w] .. wg_table[w]+n_passes-1
    const uint64_t* wg_site0;
    const uint64_t* wg_site_end;
    const uint32_t* wg_table;
    const uint32_t* wg_prior;    // optional: row of prior / anc_weight per work-group (else the table index)
    const uint64_t* wg_out;      // optional: output index of the work-group's first site (else the site index), so
                                 // that several evaluations of the same block can run in one launch
    size_t n_wgs;
    int stagger_sleep;           // first-generation stagger of co-resident work-groups, x 1024 cycles (0 = off)
    int prio_mode;               // experiment (PCSF_PRIO_MODE): 1 = the wave in the odd hardware slot of its SIMD runs at priority 0,
                                 // the other at 3 (asymmetric: one work-group of a CU owns the matrix pipe, the other fills its gaps)
    uint32_t stagger_first_blocks;
    // subtree tables (variants + 256, schedule.hpp reduce_cherries): `program` is the reduced tree's, an operand >= PAIR_OPERAND_BASE_DEV names a
    // cherry, >= TRIPLE_OPERAND_BASE_DEV a node over a cherry and a leaf, and every table set of leaf_tab is [n_leaves][65] rows, then
    // [n_pairs][65^2] rows, then [n_triples][65^3] rows (leaf_rows in all) of 64 doubles
    const uint16_t* pair_leaves; // [n_pairs][2]: the two leaves of every cherry; the first n_free_pairs are operands of the program
    const uint32_t* trip_info;   // [n_free_triples][3]: leaf A | leaf B << 16 (its cherry), leaf C, first row of the triple's table in a table set
    const uint32_t* quad_info;   // [n_quads][3]: leaf A | leaf B << 16, leaf C | leaf D << 16 (code order), first row of the quad's table
    int n_pairs, n_free_pairs, n_triples, n_free_triples, n_quads;   // n_pairs / n_triples: tables in the set; n_free_*: operands of the program (the first ones)
    uint32_t leaf_rows;
    uint32_t pair_lds_offset;    // set by the launcher: where the work-group's pair codes (2 bytes per free pair and site), then its triples' rows
                                 // (4 bytes per triple and site) start in its dynamic LDS
    const double* gap_tab;       // variants + 128: [n_passes][n_edges + 1][64] recorded messages of all-gap subtrees (lane order), see kernels.hip
    double* gap_dump;            // diagnostic variants (2, 66) only: where work-group 0 records them; else null
    unsigned long long* stamps;  // diagnostic variant only: [2][n_ops][4] s_memtime values, else null
    uint32_t stamp_block;
    double* out_score;       // [n_sites] or null
    double* out_lnz;         // [n_passes][n_sites] or null
    double* out_anc;         // [n_passes][n_sites] or null: ancestral-composition term per site
};

// variant: 0 = 4 waves (64 sites) per work-group, 1 = 8 waves (128 sites) per work-group,
// 2 = variant 0 with in-kernel stamps (diagnostic only), 16 = block mode (per-work-group windows and tables),
// 32 = variant 0 with the six-frame front end fused in (codons formed from nucleotide rows in the prologue);
// + 64 (variants 0, 16, 32): trees that need more than MAX_STACK_DEPTH_DEV parked tiles -- the stack lives in scratch memory
// + 128 (variants 0, 32, also with + 64): all-gap subtrees of a work-group's columns are skipped (needs op_leaves / op_meta)
// + 256 (variants 0, 2, 32, with + 64 / + 128): cherry tables -- the reduced tree's program, pair rows behind the leaves' rows
hipError_t launch_prune_fixed(const PruneParams& p, int variant, hipStream_t stream, size_t extra_lds = 0);
constexpr uint32_t PAIR_OPERAND_BASE_DEV = 8192, TRIPLE_OPERAND_BASE_DEV = 12288, QUAD_OPERAND_BASE_DEV = 14336, PAIR_CODES_DEV = 65 * 65;   // schedule.hpp
// One table of a table set made from two others (a leaf's 65 rows count as one): row x * y_size + y of the output = P_v (row x of X o
// row y of Y) in leaf-row layout, by the product the pruning kernel would run.  frag: index of P_v's fragment image in the launch's array.
struct CombineTable {
    uint32_t x_base, x_size, y_base, y_size, out_base, frag;
};
// frag: [2][n_frag][4096]; tables of one launch must not read each other's output (cherries first, then the triples over them)
hipError_t launch_combine_tables(const CombineTable* d_tables, int n_tables, uint32_t max_rows, const double* frag, int n_frag, double* leaf_tab, uint32_t leaf_rows, hipStream_t stream);
// rows of a table set for a tree's tables (leaves' rows first)
inline uint64_t table_set_rows(int n_leaves, int n_pairs, int n_triples, int n_quads)
{
    return static_cast<uint64_t>(n_leaves) * 65u + static_cast<uint64_t>(n_pairs) * PAIR_CODES_DEV + static_cast<uint64_t>(n_triples) * 65u * PAIR_CODES_DEV +
           static_cast<uint64_t>(n_quads) * PAIR_CODES_DEV * PAIR_CODES_DEV;
}
int prune_variant_count();

}  // namespace pcsf

namespace pcsf {

// ---- batched P(t) = S * diag(exp(lambda t)) * S^-1 with the reference's post-processing -------
// (src/instance.hpp:529-549, 602-640).  One job = one branch matrix of one table set.
// One EVALUATION = all n-1 branch matrices of one table set at one tree scale; job jx of a launch is branch
// jx % n_branches of evaluation jx / n_branches.  (Per-branch job records built on the host cost more host time and
// PCIe traffic than the kernel took.)
struct ExpmEval {
    uint32_t eig;      // eigen-system index: S at eig_right + eig*4096, S^-1 at eig_left + eig*4096, lambda at eig_lambda + eig*64
    uint32_t table;    // destination table set
    uint32_t block;    // block whose sites will read the tables (selects the leaf-table rows that are written)
    uint32_t pad;
    double x;          // tree scale: branch b gets t = (double)(float)(branch_len[b] * x), like the reference's float tree
};

struct ExpmParams {
    const ExpmEval* evals;
    uint32_t n_evals;
    uint32_t n_branches;       // n_nodes - 1
    int n_edges, n_leaves;
    const int32_t* branch_slot;  // [n_branches] >= 0: internal child -> fragment image `slot` of the table set; -1: leaf b
    const float* branch_len;     // [n_branches]
    const uint64_t* leaf_codes;  // [block][n_leaves][2] bit c = row c (codon 0..63, 64 = gap) of the leaf's table is read by
                                 // the block's sites; only those rows are written (null: all rows)
    const double* eig_right;   // [n_eig][64*64] row-major, right eigenvectors in columns
    const double* eig_left;    // [n_eig][64*64] row-major, left eigenvectors in rows
    const double* eig_lambda;  // [n_eig][64]
    double* frag;              // [table][n_edges][4096]
    double* leaf_tab;          // [table][n_leaves][65*64]
    uint32_t* fail;            // [table]: set to 1 when a matrix fails validation
    double* raw_out;           // optional [n_jobs][4096]: validated P, row-major (diagnostics/tests)
};

hipError_t launch_expm(const ExpmParams& p, hipStream_t stream);

// ---- factored block-mode pruning (factored.hip): no explicit branch matrices --------------------------------------
// One job = one tile: up to 16 consecutive sites of one evaluation.  e_b[k] = exp(lambda_k t_b) comes from
// launch_edge_scale (row `eval` of edge_scale, states in row-split order), R / L / priors from the eigen-system `eig`.
constexpr uint32_t FACTORED_JOB_VALIDATES = 1u;   // FactoredJob::pad: this tile job also runs the evaluation's row-sum check (one job per evaluation)
struct FactoredJob {
    uint64_t site0;     // first site of the tile in the codon matrix
    uint64_t out0;      // index of the tile's first site in out_lnz / out_anc
    uint32_t n_sites;   // 1..16
    uint32_t eig;       // eigen-system index
    uint32_t eval;      // row of edge_scale (= the evaluation's table index)
    uint32_t pad;
};

// tiles.hip: work-group w takes jobs [job0, job0 + n_jobs) -- all of the same eigen-system -- one wave per job at a time
struct TileGroup {
    uint32_t job0, n_jobs;
};
constexpr uint32_t TILE_JOBS_PER_GROUP = 4;   // one per wave: measured on the configs[2] MAF (4 / 8 / 16: OMEGA 1.67 / 1.82 / 1.95 s)

struct FactoredParams {
    const FactoredJob* jobs;
    uint32_t n_jobs;
    const TileGroup* groups;    // tiles.hip only: sorted by n_jobs
    uint32_t n_groups;
    uint32_t group_counts[3];   // groups of 1, 2 and 3 jobs (the rest have 4 or more): a work-group gets as many waves as it has jobs
    const uint8_t* codons;      // [n_leaves][leaf_stride]
    size_t leaf_stride;
    const uint32_t* program;    // the fused pruning program of the context (schedule.hpp)
    int n_ops;
    const int32_t* slot_branch; // [n_edges] MATMUL slot -> branch (node id of the child)
    int n_edges, n_leaves;
    uint32_t n_branches;
    const double* eig_right;    // [n_eig][64*64]
    const double* eig_left;     // [n_eig][64*64]
    // the same two matrices in the order prune_factored's waves take them as MFMA A operands (may be null: strided loads from the
    // row-major images): element ((w * 16 + q) * 64 + lane) = M[16 w + (lane & 15)][4 q + (lane >> 4)] -- a wave's 16 loads are
    // 512 contiguous bytes each instead of 64 pieces of 8
    const double* eig_right_frag;
    const double* eig_left_frag;
    const double* eig_w;        // [n_eig][4][16] lane order: w = root / iroot of the symmetrisation, L = R^T diag(w) (tiles.hip only)
    const double* prior;        // [n_eig][4][16] lane order
    const double* anc_weight;   // [n_eig][4][16] lane order (read only if out_anc)
    const double* edge_scale;   // [n_eval][n_branches][64] row-split order (prune_factored; prune_tiles only when eval_x is null)
    // prune_tiles computes its edge scales itself (tiles.hip: one exponential per lane and edge, handed round through LDS) and validates the
    // evaluation (the row sums of P(t), edge_scale_kernel's check) in the tile job that carries FACTORED_JOB_VALIDATES:
    const double* eval_x;       // [n_eval] tree scale of the evaluation (null: the scales come from edge_scale, the check from edge_scale_kernel)
    const double* eig_lambda;   // [n_eig][64]
    const double* eig_c;        // [n_eig][64] c = L 1 (row sums of the left eigenvectors)
    const float* branch_len;    // [n_branches]
    float longest_branch;
    uint32_t* fail;             // [n_eval]
    double* out_lnz;
    double* out_anc;            // or null
    uint32_t deep;              // the tree parks more tiles than the register stack holds (schedule.stack_depth > MAX_STACK_DEPTH)
    const uint64_t* op_leaves;  // [n_ops][2] and
    const uint32_t* op_meta;    // [n_ops]: schedule.hpp ProgramFacts of `program` (null: tree of more than 128 leaves -- no op is skipped)
    // prune_factored, round 5: the tile's op list COMPILED once per call (compile_tiles_kernel) instead of derived from the op facts by every
    // evaluation's work-group -- a tile's species set is fixed for its block's whole optimisation.  tile_programs[site_tile[job.site0]]:
    // TILE_PROGRAM_WORDS words per tile (layout at compile_tiles_kernel); null: the work-group runs the facts passes itself (rounds 2-4).
    const uint32_t* tile_programs;
    const uint32_t* site_tile;  // [sites of the codon matrix]: the tile that starts at a site
    unsigned long long* product_count;   // may be null: every tile job adds the 64x64 x 64x16 products it ran (64 MFMAs = 131,072 FLOP each): the executed work of a call, for the roofline of the block strategies
    uint32_t stamp_classes;     // diagnostics: also stamp every op that runs, by class (stamps[8..15]; a stamp costs ~200 cycles: PCSF_FACTORED_STAMPS=2)
    unsigned long long* stamps; // diagnostics (may be null; PCSF_FACTORED_STAMPS): work-groups add cycles -- [0] whole kernel, [1] waiting for
                                // edge-scale / leaf-column loads, [2] in products, [3] work-groups, [4] loads, [5] products
};

// One compiled tile: words [0] = ops that run, [4..7] = the leaves with a codon in the tile (two 64-bit sets), then three arrays of 128 by the
// op's RANK among those that run: the op word; the branch of its edge | bit 31: it only scales a message in eigen form; its program slot |
// stack depth << 14 | bit 18: the message before it is all ones | bit 19: the tile its pop takes back is all ones.
constexpr uint32_t TILE_PROGRAM_WORDS = 8 + 3 * 128;
struct TileDesc {
    uint64_t site0;
    uint32_t n_sites, pad;
};
// compiles tiles [0, n_tiles) of p's codon matrix and tree (p.op_meta must be there: trees of at most 128 leaves) into d_programs and
// fills d_site_tile[tile.site0] = tile
hipError_t launch_compile_tiles(const FactoredParams& p, const TileDesc* d_tiles, uint32_t n_tiles, uint32_t* d_programs, uint32_t* d_site_tile, hipStream_t stream);

// evals[i].table selects the output row of d_scale_out and the failure flag; evals[i].eig the eigen-system
// dense_rows: evaluation i of the list writes row i of d_scale_out (the caller passes the first row of the list) instead of row evals[i].table
hipError_t launch_edge_scale(const ExpmEval* d_evals, uint32_t n_evals, uint32_t n_branches, const float* d_branch_len, const double* d_eig_right,
                             const double* d_eig_left, const double* d_eig_lambda, double* d_scale_out, uint32_t* d_fail, hipStream_t stream, bool dense_rows = false);
hipError_t launch_prune_factored(const FactoredParams& p, hipStream_t stream);
// the same jobs, one wave per tile with R and L resident in LDS (tiles.hip); jobs sorted so that p.groups cut them by eigen-system
hipError_t launch_prune_tiles(const FactoredParams& p, hipStream_t stream);
size_t prune_tiles_lds_bytes(const FactoredParams& p);

// launches groups [0, p.n_groups) with work-groups of `waves` waves (1..TILE_JOBS_PER_GROUP): the caller keeps one list of groups per
// size (chain.hip writes them that way), largest first
hipError_t launch_prune_tiles_sized(const FactoredParams& p, int waves, hipStream_t stream);

// ---- the optimiser loop on the device (chain.hip + machine.hpp) ----------------------------------------------------------------------
struct StoreEntry;
struct EigenJob;
// A round's evaluations come in two sets: set 1 = those that wait for an eigen-system solved in this round (eigen -> edge scales -> pruning
// on the engine's side stream), set 0 = all others (edge scales -> pruning on its main stream, beside the eigen-solver's launch)
struct ChainCounters {       // zeroed before every round; read by the host after advance_blocks to size the round's launches
    uint32_t n_active;       // blocks that are not finished
    uint32_t n_eig_jobs;     // evaluations with a new eigen-system (eigen_omega_kernel work-groups)
    uint32_t n_transient;    // of which into a transient slot (transient0 + 0 .. n_transient - 1): the host grows the eigen-system buffers before the launch if need be
    uint32_t pad_;
    uint32_t n_evals[2];     // evaluations requested this round, by set (edge_scale_kernel work-groups)
    uint32_t n_tile_jobs[2];
    uint32_t n_groups[2][4]; // tile groups by set and number of tiles: 1, 2, 3, TILE_JOBS_PER_GROUP (all in [0] when !split_sizes)
    uint32_t n_list[2];      // pack_models: tile jobs of the coding / non-coding model (lists of their own: any four of a list share a work-group)
};
struct ChainParams {
    void* machines;              // [n_blocks] MleMachine or OmegaMachine
    StoreEntry* stores;          // [n_blocks][1 or 2][STORE_ENTRIES]
    const uint64_t* off;         // [n_blocks + 1] sites of the blocks in the codon matrix
    uint32_t slots_per_block;    // P: evaluation slots (tables, output ranges) per block
    const double* out_lnz;       // [P * n_sites] written by the pruning kernel of the previous round
    const double* out_anc;       // or null
    uint32_t* fail;              // [n_blocks * P]
    double* eval_x;              // [n_blocks * P] tree scale of every evaluation slot (prune_tiles computes its edge scales from it), or null
    const double* draws[3];      // restart points of the MLE, rho and kappa search ranges (optimizer.hpp DrawTable)
    uint32_t n_draws;
    uint32_t eager, max_points, max_restarts, held_systems;   // machine.hpp LookaheadPolicy
    uint32_t transient0;         // OMEGA: first eigen-system slot of the round's transient systems (probes that are not held: dense, counters->n_transient)
    uint32_t use_tiles, split_sizes;
    uint32_t own_scales;         // prune_factored makes its own edge scales: FactoredJob::eval = the evaluation slot (eval_x, failure flag) like prune_tiles', not a row of scales
    uint32_t pack_models;        // MLE + prune_tiles: the two ECM eigen-systems are shared by all blocks -- tile jobs go to one list per model
                                 // (the second at tile_jobs + list_capacity) and every four consecutive jobs of a list form a work-group
    uint32_t list_capacity;
    ChainCounters* counters;
    ExpmEval* evals;             // [2][eval_capacity]
    uint32_t eval_capacity;
    EigenJob* eig_jobs;          // [eval_capacity] (OMEGA)
    FactoredJob* tile_jobs;      // [2][job_capacity]
    uint32_t job_capacity;
    TileGroup* groups;           // [2][4][group_capacity]
    uint32_t group_capacity;
};
// algo 0: MLE machines, else OMEGA machines; one wave per block
hipError_t launch_advance_blocks(const ChainParams& p, int algo, uint32_t n_blocks, hipStream_t stream);
// out[i] = log prior of the OMEGA objective at xs[i]: kinds[i] == EVAL_RHO (2) the tree-scale prior, else the kappa prior
hipError_t launch_log_priors(const uint32_t* d_kinds, const double* d_xs, uint32_t n, double* d_out, hipStream_t stream);
// tests: MaxLik1D on the device, fed with recorded objective values (d_last[0] = last lpr, [1] = last x)
hipError_t launch_maxlik_replay(double init, double lo, double hi, const double* d_draws, uint32_t n_draws, const double* d_values, int n_values, double* d_xs_out,
                                int cap, int* d_n_requests, double* d_last, hipStream_t stream);

// ---- the smoothing pass (hmm.hip): chunks of runs of raw scores, one thread per chunk ---------------------------------------------
struct HmmModel {   // host/hmm.hpp `Hmm`: state 0 = coding, 1..3 = non-coding classes
    double init[4];
    double trans[4][4];
};
struct HmmChunk {   // positions [begin, end) of the run [run_begin, run_end); chunks of a run are consecutive, in position order
    uint64_t run_begin, run_end, begin, end;
};
// mode 0: forward (d_fwd[n][4]), 1: Viterbi (d_back[n]: four 2-bit predecessors per position), 2: backward + posterior (reads d_fwd,
// writes d_post[n]).  d_todo: the chunks to run (null: chunks 0..n_todo-1); rerun: start from the neighbouring chunk's exit state
// instead of burning in.  d_entry / d_exit [n_chunks][4]: the state a chunk started from / ended with.
hipError_t launch_hmm_chunks(int mode, const HmmModel& h, const HmmChunk* d_chunks, const uint32_t* d_todo, uint32_t n_todo, bool rerun, uint32_t burn_in,
                             const double* d_emit, double* d_fwd, uint8_t* d_back, double* d_post, double* d_entry, double* d_exit, hipStream_t stream);
// d_bins[i] = number of thresholds <= d_post[i] (ascending, fewer than 65,535); NaN: 0xffff and an entry in the exception list
hipError_t launch_hmm_bins(const double* d_post, uint64_t n, const double* d_thresholds, uint32_t n_thresholds, uint16_t* d_bins, uint64_t* d_exc_index,
                           double* d_exc_value, uint32_t exc_capacity, unsigned int* d_n_exc, hipStream_t stream);
// d_same[c] = chunk c's entry state equals, bit for bit, the exit state of the chunk it depends on (c - 1; backward: c + 1)
hipError_t launch_hmm_compare(const HmmChunk* d_chunks, uint32_t n_chunks, bool backward, const double* d_entry, const double* d_exit, uint8_t* d_same,
                              hipStream_t stream);

// Batched eigen-systems of the kappa/omega F3x4 rate matrix (eigen.hip): one job = one rate matrix.
struct EigenJob {
    uint32_t eig;      // destination eigen-system index
    uint32_t table;    // table set whose validation flag reports an invalid model (fail[table])
    int32_t warm;      // >= 0: start the Jacobi iteration in the eigenbasis stored in slot `warm` (may equal eig); -1: cold
    int32_t pad;
    double qs[12];     // kappa, omega, sigma, nine nucleotide-frequency ratios (src/run.hpp:75-79)
};

struct EigenParams {
    const EigenJob* jobs;
    uint32_t n_jobs;
    double* eig_right;    // [n_eig][64*64]
    double* eig_left;     // [n_eig][64*64] (may be null: prune_tiles uses L = R^T diag(w))
    double* eig_right_frag;   // [n_eig][64*64] in prune_factored's A-operand order (FactoredParams; may be null)
    double* eig_left_frag;
    double* eig_lambda;   // [n_eig][64]
    double* eig_basis;    // [basis_slots][64*64] orthogonal eigenvector matrix of the symmetrised rate matrix (warm starts; may be null)
    uint32_t basis_slots; // eigen-systems >= this index keep no basis (nothing warm-starts from them: the probes of a round that are not held); 0 = all do
    double* prior;        // [n_eig][64] equilibrium prior in lane order (may be null)
    double* anc_weight;   // [n_eig][64] ln(prior)*prior in lane order (may be null)
    double* prior_plain;  // optional [n_eig][64] prior in state order (diagnostics/tests)
    double* eig_w;        // optional [n_eig][64] lane order: root / iroot of the symmetrisation (left = right^T diag(w); tiles.hip)
    double* eig_c;        // optional [n_eig][64] state order: c = left 1, the row sums of the left eigenvectors (tiles.hip's row-sum check)
    uint32_t* fail;       // [table]: set to 1 when pi has a non-positive entry
    unsigned long long* stamps;  // diagnostics (may be null): job 0 adds its cycle counts -- [0] phase A (B <- J^T B J), [1] wave 0's
                                 // rotations, [2] wave 1's V update, [3] phase B incl. its barrier, [4] steps, [5] sweeps, [6] whole kernel
};

hipError_t launch_eigen_omega(const EigenParams& p, hipStream_t stream);
void eigen_schedule_for_tests(int* pairs, int* located);   // host: the register-resident variant's pair schedule
int eigen_schedule_quads_for_tests(int* pairs);               // host: the seven-wave variant's seats replayed on indices; returns the number of inconsistencies

}  // namespace pcsf

namespace pcsf {

// one (alignment, strand, frame) segment of the six-frame front end
struct FrameSegment {
    uint64_t col0;      // first column of the alignment in the nucleotide matrix
    uint64_t dst0;      // first site of the segment in the codon matrix
    uint32_t len;       // alignment length in columns
    uint32_t skip;      // columns skipped before the first codon (in the strand's own orientation)
    uint32_t n_codons;
    uint32_t minus;     // 0 = '+' strand, 1 = '-' strand (reverse complement, codons stored mirrored)
};

// d_first_bad (may be null): atomicMin of row * row_stride + col over the characters outside ACGTNacgtn-. that a codon touches
// (the caller initialises it to BLS_NO_BAD)
hipError_t launch_extract_frames(const FrameSegment* d_seg, uint32_t n_segments, uint32_t max_codons, const unsigned char* d_nucl,
                                 size_t row_stride, int n_leaves, unsigned char* d_codons, size_t leaf_stride, unsigned long long* d_first_bad,
                                 hipStream_t stream);

// ---- branch-length score (power track) ---------------------------------------------------------------------
constexpr int BLS_MAX_NODES = 255;  // species sets are 128-bit masks: at most 128 leaves

struct BlsTree {            // device pointers; nodes in the model's order (children before parents, root last)
    const int16_t* child1;
    const int16_t* child2;
    const double* len;      // [n_nodes] branch lengths in DOUBLE (the likelihood uses the float ones)
    const uint64_t* range;  // [n_nodes][2] leaves below the node
    int n_nodes, n_leaves;
    double total;           // score of the full species set (the divisor)
    // the internal nodes in the order a stack machine evaluates them (bls_program): node | left << 8 | right << 16 | flags << 24
    const uint32_t* prog;
    int n_prog;
};
constexpr int BLS_STACK_DEPTH = 10;   // values of internal nodes waiting for their parent (deeper subtree first: <= log2(n internal) + 1)
constexpr uint32_t BLS_LEFT_LEAF = 1u, BLS_RIGHT_LEAF = 2u, BLS_RIGHT_ON_TOP = 4u;   // flags: a child is a leaf (its value is its branch length, nothing
                                                                                    // is popped); both internal: the right child's value was pushed last
struct BlsSpan {            // columns [col0, col0 + len) of the nucleotide matrix belong to one alignment
    uint64_t col0;
    uint64_t len;
};
constexpr unsigned long long BLS_NO_BAD = ~0ull;
// d_out[col] for every column of every span; *d_first_bad = smallest row * row_stride + col holding a character
// outside ACGTNacgtn-. (BLS_NO_BAD if none; the caller initialises it)
hipError_t launch_bls(const BlsTree& tree, const BlsSpan* d_spans, uint32_t n_spans, uint64_t max_len, const unsigned char* d_nucl,
                      size_t row_stride, double* d_out, unsigned long long* d_first_bad, hipStream_t stream);

}  // namespace pcsf

namespace pcsf {

// ---- wig text on the device (wig.hip) ---------------------------------------------------------------------------------------------
// One record per (file, alignment): n line candidates -- the codons of a frame in forward genome order, or the codons of the power
// track -- whose scores start at scores[score0] (frames) and whose per-base branch-length scores start at bls[bls0] (three per candidate).
struct WigFrame {
    uint64_t elem0;       // index of the record's first candidate among all candidates of the call (files in order 0..6, alignments in order)
    uint64_t score0;
    uint64_t bls0;
    int64_t start_pos;    // genome position (1-based) of the first candidate's line
    uint32_t chrom0, chrom_len;   // the alignment's chromosome name in `chroms`
    uint32_t n;
    uint32_t pad;
};
struct WigParams {
    const WigFrame* frames;       // file 0 (power track), then files 1..6 (PhyloCSFRaw+1 .. -3)
    uint32_t file_frame0[8];      // records of file f: [file_frame0[f], file_frame0[f + 1])
    uint64_t file_elem0[8];       // candidates of file f: [file_elem0[f], file_elem0[f + 1])
    uint64_t n_elems;
    const double* scores;
    const double* bls;
    const char* chroms;
    float threshold3;             // 3 x --power-threshold: a codon whose three per-base scores sum to less is not written
    uint32_t* lens;               // [n_elems + 1]
    uint32_t* offsets;            // [n_elems + 1] exclusive prefix sums of lens: byte offset of every candidate's text
    char* text;
    uint64_t text_capacity;       // bytes of `text`: a line that would end beyond is not stored (the caller reads the total, grows and writes again)
    uint32_t* bounds;             // [9], written by the write pass: byte offsets of the eight file boundaries; [8] = needs_host
    uint32_t* needs_host;         // = bounds + 8: set when a finite |value| >= 9e14 turns up (format_trimmed's snprintf path: the caller formats on the host)
};
size_t wig_scan_temp_bytes(uint64_t n_elems);
hipError_t launch_wig_lengths(const WigParams& p, void* d_temp, size_t temp_bytes, hipStream_t stream);
hipError_t launch_wig_write(const WigParams& p, hipStream_t stream);
hipError_t launch_wig_format_debug(const float* v, uint32_t n, int decimals, char* text, uint32_t* lens, hipStream_t stream);

}  // namespace pcsf
