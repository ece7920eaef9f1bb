// blocks.cpp -- run() for a batch of alignment blocks (reference src/run.hpp:57-211) on the GPU.
//
// FIXED : one tracks-mode launch over all sites (ln z and the ancestral term per site and model),
//         per-block sums on the host in site order.
// MLE   : every block maximises over the tree scale, coding model first, then non-coding, sharing
//         one mt19937 per block like the reference.  All blocks advance in lock-step rounds.  Per round the
//         host asks every active block's optimiser for the points it will or may need next (optimizer.hpp:
//         upcoming()), queues one EVALUATION per point not seen before -- the n-1 branch matrices at that
//         scale (expm.hip) into a table set, then the block's sites pruned with them (block-mode kernel) --
//         runs the whole batch, stores the values, and feeds each optimiser from its store until it asks for
//         a point that is not there.  The sequence of values an optimiser sees is the one it would see
//         evaluating one point at a time (a re-evaluation of the same point returns the same bits), so
//         results do not change; the rounds drop from ~250 per stage for blocks that exhaust the random
//         restarts to a few tens.
// OMEGA : same machinery with the kappa/omega F3x4 rate matrix; rho evaluations only re-exponentiate,
//         kappa evaluations rebuild Q and its eigen-system on the device (eigen.hip).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <memory>
#include <mutex>
#include <random>
#include <stdexcept>
#include <thread>
#include <type_traits>
#include <vector>

#include "ctx.hpp"
#include "machine.hpp"
#include "optimizer.hpp"

namespace pcsf {

namespace {

#define HIP_TRY(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) throw std::runtime_error(std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

template <class F>
void parallel_for(size_t n, F&& f)
{
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    const size_t nt = std::min<size_t>(std::min<size_t>(hw, 32), n);
    if (nt <= 1) {
        for (size_t i = 0; i < n; ++i) f(i);
        return;
    }
    std::atomic<size_t> next{0};
    std::vector<std::thread> th;
    for (size_t t = 0; t < nt; ++t)
        th.emplace_back([&]() {
            for (size_t i = next++; i < n; i = next++) f(i);
        });
    for (auto& t : th) t.join();
}

// Device state shared by the lock-step strategies.  One EVALUATION = (block, eigen-system, tree scale): its
// branch matrices go to a table set of its own, its per-site ln z to an output segment of its own, so any
// number of evaluations -- also of the same block -- can share a run().
// The path rule of DESIGN.md section 4: does block b (sites [s0, s1) of the leaf-major codon matrix) go through the factored
// kernel?  max_tiles >= 0 replaces the cost rule by "at most max_tiles tiles" (PCSF_FACTORED_MAX_TILES).
static bool block_prefers_factored(const pcsf_ctx* c, const uint8_t* codons, size_t stride, uint64_t s0, uint64_t s1, long max_tiles)
{
    const int nl = c->n_leaves, n = c->n_nodes, ne = static_cast<int>(c->schedule.slot_node.size());
    const uint64_t sites = s1 - s0, tiles = (sites + 15) / 16;
    if (max_tiles >= 0) return tiles <= static_cast<uint64_t>(max_tiles);
    std::vector<char> live(static_cast<size_t>(n), 0);  // a species with a codon in the block below this node
    size_t present = 0, active_edges = 0;
    for (int l = 0; l < nl; ++l) {
        const uint8_t* row = codons + static_cast<size_t>(l) * stride;
        char any = 0;
        for (uint64_t s = s0; s < s1 && !any; ++s) any = row[s] < 64;
        live[l] = any;
        present += any;
    }
    for (int v = nl; v < n; ++v) live[v] = live[c->child1[v]] || live[c->child2[v]];
    for (int v = nl; v < n - 1; ++v) active_edges += live[v];
    const double factored_cost = static_cast<double>(tiles) * static_cast<double>(present + 2 * active_edges) * 64.0;
    const double explicit_cost = static_cast<double>(n - 1) * 256.0 + static_cast<double>((sites + 63) / 64) * ne * 64.0 * 4.0;
    return factored_cost <= explicit_cost;
}

// Which pruning kernel the blocks of a call get (see BlockEngine::use_tiles).  PCSF_BLOCK_KERNEL=tiles|factored overrides everything;
// PCSF_FACTORED_MAX_TILES (round 2's path rule) implies factored; then what the caller pinned (pcsf_set_block_kernel: the command line
// decides once per FILE, before it cuts the file into per-device lists); else the AUTO rule on this call's list.
static bool call_uses_tiles(const pcsf_ctx* c, const uint8_t* codons, size_t stride, const uint64_t* off, size_t n_blocks)
{
    if (std::getenv("PCSF_FACTORED_MAX_TILES")) return false;
    if (const char* v = std::getenv("PCSF_BLOCK_KERNEL")) return std::string(v) != "factored";
    if (c->block_kernel == PCSF_BLOCK_KERNEL_TILES) return true;
    if (c->block_kernel == PCSF_BLOCK_KERNEL_FACTORED) return false;
    return list_prefers_tiles(static_cast<size_t>(c->n_leaves), codons, stride, off, n_blocks);
}

struct BlockEngine {
    pcsf_ctx* c;
    size_t n_blocks, leaf_stride, n_sites_total;
    const uint64_t* off;
    hipStream_t stream;
    int ne, nl, n;
    DeviceBuffer d_codons;                 // [nl][leaf_stride] compact copy
    DeviceBuffer d_frag, d_leaf;   // table sets, one per evaluation of a run
    DeviceBuffer d_eig_r, d_eig_l, d_eig_lam, d_eig_basis, d_prior, d_ancw, d_eig_w, d_eig_c;  // eigen-systems, their root priors, L = R^T diag(w), c = L 1
    DeviceBuffer d_eig_rf, d_eig_lf;   // prune_factored: both matrices once more in its A-operand order (kernels.h FactoredParams::eig_right_frag)
    DeviceBuffer d_eval_x;    // host engine + prune_tiles: the tree scale of every evaluation of a run (the kernel makes its own edge scales)
    std::vector<double> eval_x;
    float longest_branch = 0.0f;
    DeviceBuffer d_jobs, d_wg0, d_wg1, d_wgt, d_wgp, d_wgo;
    DeviceBuffer d_branch_slot, d_branch_len, d_leaf_codes;  // per-branch constants; per (block, leaf) row masks
    // factored path (factored.hip): evaluations of blocks of at most fact_max_tiles 16-site tiles never form their
    // branch matrices; one job per tile, one edge-scale record per evaluation
    DeviceBuffer d_scale, d_slot_branch;
    // One upload and one download per round: the eigen jobs, the factored path's evaluations and tile jobs travel in ONE page-locked
    // blob (d_up), the outputs and failure flags come back in one (d_out) -- a round had ~9 small copies from / to pageable memory,
    // each a synchronous staging copy of its own (~0.1 of a ~1 ms round).
    struct PinnedBuffer {
        void* ptr = nullptr;
        size_t bytes = 0;
        PinnedBuffer() = default;
        PinnedBuffer(const PinnedBuffer&) = delete;
        PinnedBuffer& operator=(const PinnedBuffer&) = delete;
        ~PinnedBuffer() { if (ptr) (void)hipHostFree(ptr); }
        hipError_t reserve(size_t n)
        {
            if (n <= bytes) return hipSuccess;
            if (ptr) (void)hipHostFree(ptr);
            ptr = nullptr;
            bytes = 0;
            const hipError_t e = hipHostMalloc(&ptr, n + n / 2, hipHostMallocDefault);
            if (e == hipSuccess) bytes = n + n / 2;
            return e;
        }
    };
    DeviceBuffer d_up, d_out;
    PinnedBuffer h_up, h_out;
    EigenJob* p_eig_jobs = nullptr;      // views into d_up / d_out, set by run()
    ExpmEval* p_fevals = nullptr;
    FactoredJob* p_fjobs = nullptr;
    double* p_lnz = nullptr;
    double* p_anc = nullptr;
    uint32_t* p_fail = nullptr;
    std::vector<FactoredJob> fjobs[2];
    // tiles.hip (the default): every block of an engine that may apply the factors directly goes through prune_tiles -- one wave per
    // tile, R and L resident in LDS -- whatever its length; jobs are ordered by eigen-system and cut into work-groups at run().
    // PCSF_BLOCK_KERNEL=factored brings back round 2's pair (prune_factored for short / sparse blocks, expm + block-mode prune_fixed
    // for long dense ones).
    bool use_tiles;
    // ... for lists of DENSE blocks.  An evaluation with an eigen-system of its own (every OMEGA kappa probe) brings only its block's
    // tiles, and a work-group holds 36 KB of LDS whatever its size: the short, sparse blocks of real alignments (median 6 codons, 4 of
    // 100 species) leave the CUs at four waves, and prune_factored -- four waves sharing one tile, R / L rows in registers -- is the
    // better tool there (8,256 real blocks, OMEGA: 1.44 s factored, 1.78 s tiles; configs[2], 43 of 58 species: 1.97 s and 1.76 s).
    // One kernel per LIST (score_blocks_impl decides from the mean number of species present, before the list is cut into groups, so
    // grouping never changes a result): both kernels in one round run one after the other, slower than either alone, also on two
    // streams (measured).  PCSF_BLOCK_KERNEL=tiles|factored overrides.
    std::vector<char> block_tiles;
    size_t tile_jobs_per_group = TILE_JOBS_PER_GROUP;
    std::vector<FactoredJob> tjobs[2];
    FactoredJob* p_tjobs = nullptr;
    std::vector<TileGroup> tgroups[2];
    TileGroup* p_tgroups = nullptr;
    std::vector<uint32_t> order_scratch;
    DeviceBuffer d_products;     // products (64 MFMAs each) the pruning kernels of this engine ran: the executed work, read under PCSF_TIME_BLOCKS
    unsigned long long products_run()
    {
        unsigned long long v = 0;
        if (d_products.ptr) (void)hipMemcpy(&v, d_products.ptr, sizeof v, hipMemcpyDeviceToHost);
        return v;
    }
    DeviceBuffer d_fact_stamps;  // PCSF_FACTORED_STAMPS: cycle sums of prune_factored (printed when the engine goes away)
    bool fact_stamp_classes = false;   // PCSF_FACTORED_STAMPS=2: also per class of op (every op that runs is stamped: ~400 cycles each)
    std::vector<ExpmEval> fevals[2];
    // Which path an evaluation of block b takes is decided once per block from the matrix work either path needs (in
    // MFMA instructions): explicit = 256 per branch + 64 per internal edge, wave and 64-site window; factored = 64 per
    // tile and product, one product per species present in the block, two per internal edge above such a species
    // (all-gap subtrees cost nothing there).  Dense blocks break even at 4 tiles; the sparse blocks of real alignments
    // (median 4 of 100 species) stay factored at any length.  PCSF_FACTORED_MAX_TILES=n replaces the rule by "at most n tiles".
    std::vector<char> block_factored;
    long fixed_max_tiles = -1;
    bool host_eigen = false;     // PCSF_EIGEN_HOST=1: parity mode, see run()
    bool allow_factored = true;  // false: caller-supplied eigen-systems keep the entry-wise post-processing of P
    // Work of one run() is kept in two sets: evaluations whose eigen-system is rebuilt in this run (set 1: eigen ->
    // expm -> prune on the side stream) and all others (set 0: expm -> prune on the main stream), so that the
    // latency-bound eigen kernel overlaps the matrix work of the other evaluations.
    std::vector<ExpmEval> jobs[2];
    std::vector<EigenJob> eig_jobs;
    std::vector<uint64_t> wg0[2], wg1[2], wgo[2];
    std::vector<uint32_t> wgt[2], wgp[2];
    struct Eval {
        uint32_t block, table;
        size_t out0;
    };
    std::vector<Eval> evals;
    std::vector<double> lnz, anc;
    std::vector<uint32_t> fail;
    size_t cap_tables = 0, cap_eig = 0, out_used = 0;
    hipStream_t side = nullptr;
    hipEvent_t ev_start = nullptr, ev_side = nullptr;


    // own_streams: the engine works on streams of its own instead of the context's, so that several engines -- groups
    // of blocks advancing independently, one host thread each -- overlap on the device
    // explicit_path: the caller will queue evaluations with eigen-systems of its own (allow_factored = false): they form branch matrices
    BlockEngine(pcsf_ctx* ctx, const uint8_t* codons, const uint64_t* offsets, size_t nb, size_t stride, bool tiles, bool own_streams = false,
                bool explicit_path = false)
        : c(ctx), n_blocks(nb), leaf_stride(stride), off(offsets), stream(ctx->stream), use_tiles(tiles)
    {
        const bool allow_factored_at_construction = !explicit_path;
        const bool marks = std::getenv("PCSF_TIME_SETUP") != nullptr;
        auto t_mark = std::chrono::steady_clock::now();
        auto mark = [&](const char* what) {
            if (!marks) return;
            const auto now = std::chrono::steady_clock::now();
            fprintf(stderr, "  engine set-up: %-28s %7.0f us\n", what, std::chrono::duration<double>(now - t_mark).count() * 1e6);
            t_mark = now;
        };
        if (own_streams) {
            HIP_TRY(c->block_streams.acquire(&own_main));
            stream = own_main;
        }
        mark("main stream");
        struct Guard {  // a throw below must not leak the stream created above (the destructor does not run then)
            BlockEngine* e;
            ~Guard() { if (e) e->release_sync_objects(); }
        } guard{this};
        ne = static_cast<int>(c->schedule.slot_node.size());
        nl = c->n_leaves;
        n = c->n_nodes;
        n_sites_total = offsets[nb];
        for (float len : c->branch_len) longest_branch = std::max(longest_branch, len);
        if (n_sites_total > 0) {
            HIP_TRY(d_codons.reserve(static_cast<size_t>(nl) * n_sites_total));
            HIP_TRY(upload_rows(d_codons.ptr, codons, stride, n_sites_total, nl, stream));
        }
        mark("codon upload");
        if (const char* v = std::getenv("PCSF_FACTORED_MAX_TILES")) fixed_max_tiles = std::max(0, std::atoi(v));
        if (fixed_max_tiles >= 0) use_tiles = false;   // the tile-count rule is a statement about round 2's two paths
        if (const char* v = std::getenv("PCSF_TILE_JOBS")) tile_jobs_per_group = static_cast<size_t>(std::max(1, std::atoi(v)));
        if (const char* v = std::getenv("PCSF_EIGEN_HOST")) host_eigen = std::atoi(v) != 0;
        HIP_TRY(d_products.reserve(sizeof(unsigned long long)));
        HIP_TRY(hipMemsetAsync(d_products.ptr, 0, sizeof(unsigned long long), stream));
        if (const char* fs = std::getenv("PCSF_FACTORED_STAMPS")) {
            fact_stamp_classes = std::atoi(fs) >= 2;
            HIP_TRY(d_fact_stamps.reserve(16 * sizeof(unsigned long long)));
            HIP_TRY(hipMemsetAsync(d_fact_stamps.ptr, 0, 16 * sizeof(unsigned long long), stream));
        }
        block_factored.assign(nb, (use_tiles || fixed_max_tiles < 0) ? 1 : 0);   // the explicit path (expm + block-mode prune_fixed) only under PCSF_FACTORED_MAX_TILES
        block_tiles.assign(nb, use_tiles ? 1 : 0);
        // which rows of each leaf's message table the block's sites read (values above 64 count as 64, like the kernel): only the
        // explicit path writes tables, and only a caller-supplied eigen-system or PCSF_BLOCK_KERNEL=factored can take it (20 ms of a
        // 0.25 s MLE call on the configs[2] list when it was computed regardless, on one thread)
        if (!use_tiles || !allow_factored_at_construction) {
            leaf_codes.assign(nb * static_cast<size_t>(nl) * 2, 0);
            parallel_for(nb, [&](size_t b) {
                for (int l = 0; l < nl; ++l) {
                    uint64_t lo = 0, hi = 0;
                    const uint8_t* row = codons + static_cast<size_t>(l) * stride;
                    for (uint64_t s = offsets[b]; s < offsets[b + 1]; ++s) {
                        const unsigned cd = row[s] > 64 ? 64u : row[s];
                        if (cd < 64) lo |= uint64_t(1) << cd;
                        else hi |= 1;
                    }
                    leaf_codes[(b * nl + l) * 2] = lo;
                    leaf_codes[(b * nl + l) * 2 + 1] = hi;
                }
                if (!use_tiles && fixed_max_tiles >= 0) block_factored[b] = block_prefers_factored(c, codons, stride, offsets[b], offsets[b + 1], fixed_max_tiles);
            });
        }
        mark("row masks / path rule");
        std::vector<int32_t> slots(static_cast<size_t>(n - 1));
        for (int b = 0; b < n - 1; ++b) slots[b] = c->branch_slot[b];
        std::vector<int32_t> slot_branch(c->schedule.slot_node.begin(), c->schedule.slot_node.end());
        HIP_TRY(d_slot_branch.reserve(std::max<size_t>(slot_branch.size(), 1) * sizeof(int32_t)));
        if (!slot_branch.empty())
            HIP_TRY(hipMemcpyAsync(d_slot_branch.ptr, slot_branch.data(), slot_branch.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
        HIP_TRY(d_branch_slot.reserve(slots.size() * sizeof(int32_t)));
        HIP_TRY(d_branch_len.reserve(slots.size() * sizeof(float)));
        HIP_TRY(d_leaf_codes.reserve(std::max<size_t>(leaf_codes.size(), 2) * sizeof(uint64_t)));   // (expm_kernel writes every row when a block has none)
        HIP_TRY(hipMemcpyAsync(d_branch_slot.ptr, slots.data(), slots.size() * sizeof(int32_t), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(d_branch_len.ptr, c->branch_len.data(), slots.size() * sizeof(float), hipMemcpyHostToDevice, stream));
        if (!leaf_codes.empty())
            HIP_TRY(hipMemcpyAsync(d_leaf_codes.ptr, leaf_codes.data(), leaf_codes.size() * sizeof(uint64_t), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));  // `slots` is a stack temporary
        mark("constants + sync");
        // created last: a throw above leaves nothing behind that only the destructor (not run then) would release
        HIP_TRY(c->block_streams.acquire(&side));
        HIP_TRY(hipEventCreateWithFlags(&ev_start, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ev_side, hipEventDisableTiming));
        mark("side stream + events");
        guard.e = nullptr;  // constructed: the destructor owns the release from here on
    }
    hipStream_t own_main = nullptr;
    void release_sync_objects()
    {
        if (own_main) {
            (void)hipStreamSynchronize(own_main);
            c->block_streams.release(own_main);
            own_main = nullptr;
        }
        if (side) {
            (void)hipStreamSynchronize(side);
            c->block_streams.release(side);
            side = nullptr;
        }
        if (ev_start) (void)hipEventDestroy(ev_start);
        if (ev_side) (void)hipEventDestroy(ev_side);
        ev_start = ev_side = nullptr;

    }
    std::vector<uint64_t> leaf_codes;  // [block][leaf][2]
    ~BlockEngine()
    {
        if (d_fact_stamps.ptr) {
            unsigned long long h[16] = {0};
            if (hipMemcpy(h, d_fact_stamps.ptr, sizeof h, hipMemcpyDeviceToHost) == hipSuccess && h[3]) {
                fprintf(stderr, "prune_factored stamps: %llu work-groups, per work-group %.0f cycles (prologue %.0f, root op reached at %.0f): %.1f scale loads waiting %.0f, %.1f products taking %.0f\n", h[3],
                        double(h[0]) / h[3], double(h[6]) / h[3], double(h[7]) / h[3], double(h[4]) / h[3], double(h[1]) / h[3], double(h[5]) / h[3], double(h[2]) / h[3]);
                if (fact_stamp_classes) fprintf(stderr, "  per work-group, ops that ran (count, cycles incl. their loads and products): cherry %.1f %.0f | leaf %.1f %.0f | push %.1f %.0f | pop %.1f %.0f; op facts + row list %.0f\n",
                        double(h[12]) / h[3], double(h[8]) / h[3], double(h[13]) / h[3], double(h[9]) / h[3], double(h[14]) / h[3], double(h[10]) / h[3], double(h[15]) / h[3], double(h[11]) / h[3],
                        double(h[0] - h[6] - h[8] - h[9] - h[10] - h[11]) / h[3]);
            }
        }
        release_sync_objects();
    }
    BlockEngine(const BlockEngine&) = delete;
    BlockEngine& operator=(const BlockEngine&) = delete;

    // capacity: evaluations per run() / eigen-systems alive at a time
    // table sets only live within one run(): growing between runs loses nothing; growing with evaluations queued would.
    // The branch-matrix tables themselves (9.6 MB per evaluation of a 100-leaf tree) are allocated when the first
    // evaluation takes the explicit path -- on real alignments none does, and 7 GB of hipMalloc per call is not free.
    void reserve_tables(size_t tables)
    {
        if (tables <= cap_tables) return;
        if (!evals.empty()) throw std::runtime_error("BlockEngine: reserve_tables with evaluations queued");
        HIP_TRY(d_scale.reserve(tables * static_cast<size_t>(n - 1) * 64 * sizeof(double)));
        cap_tables = tables;
        if (tables_allocated) allocate_tables();
    }
    bool tables_allocated = false;
    void allocate_tables()
    {
        HIP_TRY(d_frag.reserve(std::max<size_t>(1, cap_tables * ne) * 4096 * sizeof(double)));
        HIP_TRY(d_leaf.reserve(cap_tables * nl * LEAF_TABLE_DOUBLES_DEV * sizeof(double)));
        tables_allocated = true;
    }
    // eigen-systems already resident (uploaded, or solved by an earlier run()) survive a growth.  basis_count: how many of them keep
    // the orthogonal basis a warm start reads (the held systems of OMEGA: the first N_ROLES * n_blocks; default: all); with_left = false:
    // no left eigenvectors (prune_tiles uses L = R^T diag(w): a third of the footprint of a slot)
    size_t cap_basis = 0;
    bool have_left = true, have_right = true;
    // with_right = false: no row-major right eigenvectors either (prune_factored with its own edge scales reads the fragment-ordered copies only)
    void reserve_eig(size_t count, size_t basis_count = ~size_t(0), bool with_left = true, bool with_right = true)
    {
        if (basis_count == ~size_t(0)) basis_count = count;
        if (basis_count > cap_basis) {
            HIP_TRY(d_eig_basis.reserve_keep(basis_count * 4096 * sizeof(double), cap_basis * 4096 * sizeof(double), stream));
            cap_basis = basis_count;
        }
        if (count <= cap_eig) return;
        have_left = with_left;
        have_right = with_right;
        if (with_right) HIP_TRY(d_eig_r.reserve_keep(count * 4096 * sizeof(double), cap_eig * 4096 * sizeof(double), stream));
        if (with_left) HIP_TRY(d_eig_l.reserve_keep(count * 4096 * sizeof(double), cap_eig * 4096 * sizeof(double), stream));
        if (!use_tiles) {
            HIP_TRY(d_eig_rf.reserve_keep(count * 4096 * sizeof(double), cap_eig * 4096 * sizeof(double), stream));
            HIP_TRY(d_eig_lf.reserve_keep(count * 4096 * sizeof(double), cap_eig * 4096 * sizeof(double), stream));
        }
        HIP_TRY(d_eig_lam.reserve_keep(count * 64 * sizeof(double), cap_eig * 64 * sizeof(double), stream));
        HIP_TRY(d_prior.reserve_keep(count * 64 * sizeof(double), cap_eig * 64 * sizeof(double), stream));
        HIP_TRY(d_ancw.reserve_keep(count * 64 * sizeof(double), cap_eig * 64 * sizeof(double), stream));
        HIP_TRY(d_eig_w.reserve_keep(count * 64 * sizeof(double), cap_eig * 64 * sizeof(double), stream));
        HIP_TRY(d_eig_c.reserve_keep(count * 64 * sizeof(double), cap_eig * 64 * sizeof(double), stream));
        cap_eig = count;
    }
    // a host-side eigen-system (the ECM models of the context) and its root prior into slot idx.  pi: the frequencies the rate matrix
    // was symmetrised with (hostmath.hpp eigen_reversible: left = right^T diag(root / iroot)) -- tiles.hip needs them; null for
    // caller-supplied eigen-systems, which never take that path.
    void upload_eig(size_t idx, const EigenSystem& es, const double* pi = nullptr)
    {
        HIP_TRY(hipMemcpyAsync(d_eig_r.as<double>() + idx * 4096, es.right.data(), 4096 * sizeof(double), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(d_eig_l.as<double>() + idx * 4096, es.left.data(), 4096 * sizeof(double), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(d_eig_lam.as<double>() + idx * 64, es.lambda.data(), 64 * sizeof(double), hipMemcpyHostToDevice, stream));
        if (!use_tiles) {
            std::vector<double> rf(4096), lf(4096);
            for (int i = 0; i < 4096; ++i) {
                const int lane = i & 63, r = 16 * (i >> 10) + (lane & 15), col = 4 * ((i >> 6) & 15) + (lane >> 4);
                rf[i] = es.right[static_cast<size_t>(r) * 64 + col];
                lf[i] = es.left[static_cast<size_t>(r) * 64 + col];
            }
            HIP_TRY(hipMemcpyAsync(d_eig_rf.as<double>() + idx * 4096, rf.data(), 4096 * sizeof(double), hipMemcpyHostToDevice, stream));
            HIP_TRY(hipMemcpyAsync(d_eig_lf.as<double>() + idx * 4096, lf.data(), 4096 * sizeof(double), hipMemcpyHostToDevice, stream));
            HIP_TRY(hipStreamSynchronize(stream));  // rf / lf are temporaries
        }
        Vec w, pl, wl;
        if (pi) {
            Vec sw, swl;
            for (int x = 0; x < 64; ++x) {
                const double root = std::sqrt(pi[x]), iroot = 1.0 / root;
                sw[x] = root / iroot;
            }
            to_lane_order(sw.data(), swl.data());
            HIP_TRY(hipMemcpyAsync(d_eig_w.as<double>() + idx * 64, swl.data(), 64 * sizeof(double), hipMemcpyHostToDevice, stream));
            HIP_TRY(hipStreamSynchronize(stream));  // swl is a stack temporary
        }
        {
            Vec cs;   // c_k = sum_i left[k][i]: the row sums of the left eigenvectors (tiles.hip's row-sum check)
            for (int k = 0; k < 64; ++k) {
                double acc = 0.0;
                for (int i = 0; i < 64; ++i) acc += es.left[static_cast<size_t>(k) * 64 + i];
                cs[k] = acc;
            }
            HIP_TRY(hipMemcpyAsync(d_eig_c.as<double>() + idx * 64, cs.data(), 64 * sizeof(double), hipMemcpyHostToDevice, stream));
            HIP_TRY(hipStreamSynchronize(stream));  // cs is a stack temporary
        }
        for (int x = 0; x < 64; ++x) w[x] = std::log(es.prior[x]) * es.prior[x];
        to_lane_order(es.prior.data(), pl.data());
        to_lane_order(w.data(), wl.data());
        HIP_TRY(hipMemcpyAsync(d_prior.as<double>() + idx * 64, pl.data(), 64 * sizeof(double), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(d_ancw.as<double>() + idx * 64, wl.data(), 64 * sizeof(double), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipStreamSynchronize(stream));  // pl / wl are stack temporaries
    }

    size_t free_evaluations() const { return cap_tables - evals.size(); }

    // queue one evaluation of block `blk` at tree scale x with eigen-system slot `eig`.  new_qs != null: slot
    // `eig` is (re)built first from these kappa/omega settings; after_eigen: the slot is rebuilt by ANOTHER
    // evaluation of this run, so this one has to run behind it.  Returns the evaluation's index in this run.
    // warm >= 0: the rebuilt eigen-system starts its Jacobi iteration from the eigenbasis held in slot `warm`.
    size_t queue(size_t blk, uint32_t eig, double x, const double* new_qs, bool after_eigen = false, int32_t warm = -1)
    {
        const uint32_t table = static_cast<uint32_t>(evals.size());
        if (table >= cap_tables) throw std::runtime_error("BlockEngine: more evaluations queued than table sets reserved");
        const int set = (new_qs || after_eigen) ? 1 : 0;
        if (new_qs) {
            EigenJob j;
            j.eig = eig;
            j.table = table;
            j.warm = warm;
            j.pad = 0;
            for (int i = 0; i < 12; ++i) j.qs[i] = new_qs[i];
            eig_jobs.push_back(j);
        }
        ExpmEval ev{};
        ev.eig = eig;
        ev.table = table;
        ev.block = static_cast<uint32_t>(blk);
        ev.x = x;
        if (allow_factored && block_factored[blk]) {
            // short block: factors applied directly, one work-group per 16-site tile (an empty block only has its
            // model validated)
            ev.pad = block_tiles[blk] ? 1u : 0u;   // 1: pruned by prune_tiles (jobs made at run()), 0: by prune_factored
            fevals[set].push_back(ev);
            for (uint64_t s = off[blk]; !block_tiles[blk] && s < off[blk + 1]; s += 16) {
                FactoredJob j{};
                j.site0 = s;
                j.out0 = out_used + (s - off[blk]);
                j.n_sites = static_cast<uint32_t>(std::min<uint64_t>(16, off[blk + 1] - s));
                j.eig = eig;
                j.eval = table;
                fjobs[set].push_back(j);
            }
        } else {
            if (!tables_allocated) allocate_tables();
            jobs[set].push_back(ev);
            for (uint64_t s = off[blk]; s < off[blk + 1]; s += 64) {
                wg0[set].push_back(s);
                wg1[set].push_back(std::min<uint64_t>(s + 64, off[blk + 1]));
                wgt[set].push_back(table);
                wgp[set].push_back(eig);
                wgo[set].push_back(out_used + (s - off[blk]));
            }
        }
        evals.push_back({static_cast<uint32_t>(blk), table, out_used});
        out_used += off[blk + 1] - off[blk];
        return evals.size() - 1;
    }

    // runs everything queued; afterwards value(k) / failed(k) describe evaluation k
    void run(bool want_anc)
    {
        const size_t tables = evals.size();
        if (tables == 0) return;
        if (use_tiles)
            for (int set = 0; set < 2; ++set) {
                // tile jobs in the order of their eigen-systems (an evaluation's tiles stay together), cut into work-groups of at most
                // TILE_JOBS_PER_GROUP jobs of ONE eigen-system
                const std::vector<ExpmEval>& fe = fevals[set];
                order_scratch.clear();
                for (uint32_t i = 0; i < fe.size(); ++i)
                    if (fe[i].pad) order_scratch.push_back(i);
                std::stable_sort(order_scratch.begin(), order_scratch.end(), [&](uint32_t a, uint32_t b) { return fe[a].eig < fe[b].eig; });
                tjobs[set].clear();
                tgroups[set].clear();
                size_t run0 = 0;
                auto close_run = [&](size_t end) {   // jobs [run0, end) share an eigen-system: even parts of at most TILE_JOBS_PER_GROUP
                    const size_t cnt = end - run0;
                    if (!cnt) return;
                    const size_t parts = (cnt + tile_jobs_per_group - 1) / tile_jobs_per_group;
                    for (size_t k = 0; k < parts; ++k) {
                        const size_t a = run0 + cnt * k / parts, b = run0 + cnt * (k + 1) / parts;
                        tgroups[set].push_back({static_cast<uint32_t>(a), static_cast<uint32_t>(b - a)});
                    }
                    run0 = end;
                };
                uint32_t run_eig = 0;
                for (uint32_t i : order_scratch) {
                    const ExpmEval& ev = fe[i];
                    if (ev.eig != run_eig) {
                        close_run(tjobs[set].size());
                        run_eig = ev.eig;
                    }
                    const uint64_t s0 = off[ev.block], s1 = off[ev.block + 1], out0 = evals[ev.table].out0;
                    for (uint64_t s = s0; s < s1; s += 16) {
                        FactoredJob j{};
                        j.site0 = s;
                        j.out0 = out0 + (s - s0);
                        j.n_sites = static_cast<uint32_t>(std::min<uint64_t>(16, s1 - s));
                        j.eig = ev.eig;
                        j.eval = ev.table;
                        j.pad = s == s0 ? FACTORED_JOB_VALIDATES : 0u;
                        tjobs[set].push_back(j);
                    }
                }
                close_run(tjobs[set].size());
                std::stable_sort(tgroups[set].begin(), tgroups[set].end(), [](const TileGroup& a, const TileGroup& b) { return a.n_jobs < b.n_jobs; });
            }
        const size_t nj0 = jobs[0].size(), nj1 = jobs[1].size(), nw0 = wg0[0].size(), nw1 = wg0[1].size();
        const size_t nf0 = fjobs[0].size(), nf1 = fjobs[1].size(), nfe0 = fevals[0].size(), nfe1 = fevals[1].size();
        const size_t ng0 = tgroups[0].size(), ng1 = tgroups[1].size(), nt0 = tjobs[0].size(), nt1 = tjobs[1].size();
        auto aligned = [](size_t b) { return (b + 255) & ~size_t(255); };
        {   // the download blob: [lnz | anc | fail]
            const size_t out_bytes = aligned(out_used * 8), anc_bytes = want_anc ? out_bytes : 0;
            HIP_TRY(d_out.reserve(out_bytes + anc_bytes + aligned(tables * sizeof(uint32_t))));
            p_lnz = d_out.as<double>();
            p_anc = reinterpret_cast<double*>(static_cast<char*>(d_out.ptr) + out_bytes);
            p_fail = reinterpret_cast<uint32_t*>(static_cast<char*>(d_out.ptr) + out_bytes + anc_bytes);
            HIP_TRY(hipMemsetAsync(p_fail, 0, tables * sizeof(uint32_t), stream));
        }
        {   // the upload blob: [eigen jobs | evaluations of the factored path (both sets) | tile jobs (both sets)]
            const bool dev_eigen = !eig_jobs.empty() && !host_eigen;
            const size_t b_eig = dev_eigen ? aligned(eig_jobs.size() * sizeof(EigenJob)) : 0, b_fev = aligned((nfe0 + nfe1) * sizeof(ExpmEval)),
                         b_fj = aligned((nf0 + nf1) * sizeof(FactoredJob)), b_tj = aligned((nt0 + nt1) * sizeof(FactoredJob)),
                         b_tg = aligned((ng0 + ng1) * sizeof(TileGroup)), total = b_eig + b_fev + b_fj + b_tj + b_tg;
            if (total) {
                HIP_TRY(d_up.reserve(total));
                HIP_TRY(h_up.reserve(total));
                char* h = static_cast<char*>(h_up.ptr);
                char* d = static_cast<char*>(d_up.ptr);
                p_eig_jobs = reinterpret_cast<EigenJob*>(d);
                p_fevals = reinterpret_cast<ExpmEval*>(d + b_eig);
                p_fjobs = reinterpret_cast<FactoredJob*>(d + b_eig + b_fev);
                p_tjobs = reinterpret_cast<FactoredJob*>(d + b_eig + b_fev + b_fj);
                if (nt0) std::memcpy(h + b_eig + b_fev + b_fj, tjobs[0].data(), nt0 * sizeof(FactoredJob));
                if (nt1) std::memcpy(h + b_eig + b_fev + b_fj + nt0 * sizeof(FactoredJob), tjobs[1].data(), nt1 * sizeof(FactoredJob));
                p_tgroups = reinterpret_cast<TileGroup*>(d + b_eig + b_fev + b_fj + b_tj);
                if (ng0) std::memcpy(h + b_eig + b_fev + b_fj + b_tj, tgroups[0].data(), ng0 * sizeof(TileGroup));
                if (ng1) std::memcpy(h + b_eig + b_fev + b_fj + b_tj + ng0 * sizeof(TileGroup), tgroups[1].data(), ng1 * sizeof(TileGroup));
                if (dev_eigen) std::memcpy(h, eig_jobs.data(), eig_jobs.size() * sizeof(EigenJob));
                if (nfe0) std::memcpy(h + b_eig, fevals[0].data(), nfe0 * sizeof(ExpmEval));
                if (nfe1) std::memcpy(h + b_eig + nfe0 * sizeof(ExpmEval), fevals[1].data(), nfe1 * sizeof(ExpmEval));
                if (nf0) std::memcpy(h + b_eig + b_fev, fjobs[0].data(), nf0 * sizeof(FactoredJob));
                if (nf1) std::memcpy(h + b_eig + b_fev + nf0 * sizeof(FactoredJob), fjobs[1].data(), nf1 * sizeof(FactoredJob));
                HIP_TRY(hipMemcpyAsync(d, h, total, hipMemcpyHostToDevice, stream));
            }
        }
        if (!eig_jobs.empty() && host_eigen) {
            // parity mode (PCSF_EIGEN_HOST=1): the kappa/omega eigen-systems come from the HOST solver -- row-cyclic Jacobi from
            // a cold start, the oracle's algorithm -- instead of eigen_omega_kernel.  Far too slow for production (a few
            // milliseconds per matrix); it exists to show how much of the OMEGA end-result scatter between this library
            // and the oracle is eigen-solver rounding (tests/test_gpu_parity_blocks.py).
            std::vector<EigenSystem> es(eig_jobs.size());
            std::vector<Vec> pis(eig_jobs.size());
            std::vector<char> bad(eig_jobs.size(), 0);
            parallel_for(eig_jobs.size(), [&](size_t i) {
                Mat Q;
                rate_matrix_omega(eig_jobs[i].qs, Q, pis[i]);
                bad[i] = !eigen_reversible(Q, pis[i].data(), es[i]);
            });
            const uint32_t one = 1u;
            for (size_t i = 0; i < eig_jobs.size(); ++i) {
                if (bad[i]) HIP_TRY(hipMemcpyAsync(p_fail + eig_jobs[i].table, &one, sizeof one, hipMemcpyHostToDevice, stream));
                else upload_eig(eig_jobs[i].eig, es[i], pis[i].data());
            }
            HIP_TRY(hipStreamSynchronize(stream));  // `one` is a stack temporary
        }
        HIP_TRY(d_jobs.reserve((nj0 + nj1) * sizeof(ExpmEval)));
        if (nj0) HIP_TRY(hipMemcpyAsync(d_jobs.as<ExpmEval>(), jobs[0].data(), nj0 * sizeof(ExpmEval), hipMemcpyHostToDevice, stream));
        if (nj1) HIP_TRY(hipMemcpyAsync(d_jobs.as<ExpmEval>() + nj0, jobs[1].data(), nj1 * sizeof(ExpmEval), hipMemcpyHostToDevice, stream));
        if (nw0 + nw1) {
            const size_t nw = nw0 + nw1;
            HIP_TRY(d_wg0.reserve(nw * 8));
            HIP_TRY(d_wg1.reserve(nw * 8));
            HIP_TRY(d_wgo.reserve(nw * 8));
            HIP_TRY(d_wgt.reserve(nw * 4));
            HIP_TRY(d_wgp.reserve(nw * 4));
            for (int set = 0; set < 2; ++set) {
                const size_t at = set ? nw0 : 0, cnt = wg0[set].size();
                if (!cnt) continue;
                HIP_TRY(hipMemcpyAsync(d_wg0.as<uint64_t>() + at, wg0[set].data(), cnt * 8, hipMemcpyHostToDevice, stream));
                HIP_TRY(hipMemcpyAsync(d_wg1.as<uint64_t>() + at, wg1[set].data(), cnt * 8, hipMemcpyHostToDevice, stream));
                HIP_TRY(hipMemcpyAsync(d_wgo.as<uint64_t>() + at, wgo[set].data(), cnt * 8, hipMemcpyHostToDevice, stream));
                HIP_TRY(hipMemcpyAsync(d_wgt.as<uint32_t>() + at, wgt[set].data(), cnt * 4, hipMemcpyHostToDevice, stream));
                HIP_TRY(hipMemcpyAsync(d_wgp.as<uint32_t>() + at, wgp[set].data(), cnt * 4, hipMemcpyHostToDevice, stream));
            }
        }
        if (use_tiles) {
            eval_x.assign(tables, 0.0);
            for (int set = 0; set < 2; ++set)
                for (const ExpmEval& ev : fevals[set]) eval_x[ev.table] = ev.x;
            HIP_TRY(d_eval_x.reserve(tables * sizeof(double)));
            HIP_TRY(hipMemcpyAsync(d_eval_x.ptr, eval_x.data(), tables * sizeof(double), hipMemcpyHostToDevice, stream));
        }
        const bool use_side = !eig_jobs.empty() || nj1 || nw1 || nfe1;
        if (use_side) {
            HIP_TRY(hipEventRecord(ev_start, stream));
            HIP_TRY(hipStreamWaitEvent(side, ev_start, 0));
        }
        auto expm_and_prune = [&](int set, hipStream_t st) {
            const size_t nj = set ? nj1 : nj0, j_at = set ? nj0 : 0, nw = set ? nw1 : nw0, w_at = set ? nw0 : 0;
            if (nj) {
                ExpmParams ep{};
                ep.evals = d_jobs.as<ExpmEval>() + j_at;
                ep.n_evals = static_cast<uint32_t>(nj);
                ep.n_branches = static_cast<uint32_t>(n - 1);
                ep.n_edges = ne;
                ep.n_leaves = nl;
                ep.branch_slot = d_branch_slot.as<int32_t>();
                ep.branch_len = d_branch_len.as<float>();
                ep.leaf_codes = leaf_codes.empty() ? nullptr : d_leaf_codes.as<uint64_t>();   // null: every row of every leaf table
                ep.eig_right = d_eig_r.as<double>();
                ep.eig_left = d_eig_l.as<double>();
                ep.eig_lambda = d_eig_lam.as<double>();
                ep.frag = d_frag.as<double>();
                ep.leaf_tab = d_leaf.as<double>();
                ep.fail = p_fail;
                HIP_TRY(launch_expm(ep, st));
            }
            if (nw) {
                PruneParams p = base_prune_params(c);
                p.codons = d_codons.as<uint8_t>();
                p.n_sites = out_used;  // rows of the output arrays (one pass)
                p.leaf_stride = n_sites_total;
                p.n_passes = 1;
                p.frag = d_frag.as<double>();
                p.leaf_tab = d_leaf.as<double>();
                p.prior = d_prior.as<double>();
                p.anc_weight = d_ancw.as<double>();
                p.wg_site0 = d_wg0.as<uint64_t>() + w_at;
                p.wg_site_end = d_wg1.as<uint64_t>() + w_at;
                p.wg_table = d_wgt.as<uint32_t>() + w_at;
                p.wg_prior = d_wgp.as<uint32_t>() + w_at;
                p.wg_out = d_wgo.as<uint64_t>() + w_at;
                p.n_wgs = nw;
                p.out_lnz = p_lnz;
                p.out_anc = want_anc ? p_anc : nullptr;
                HIP_TRY(launch_prune_fixed(p, 16 | (c->schedule.stack_depth > MAX_STACK_DEPTH ? 64 : 0), st, 0));
            }
            const size_t nfe = set ? nfe1 : nfe0, fe_at = set ? nfe0 : 0, nf = set ? nf1 : nf0, f_at = set ? nf0 : 0;
            if (nfe && !use_tiles)   // prune_tiles makes its own edge scales and runs the row-sum check itself
                HIP_TRY(launch_edge_scale(p_fevals + fe_at, static_cast<uint32_t>(nfe), static_cast<uint32_t>(n - 1), d_branch_len.as<float>(),
                                          d_eig_r.as<double>(), d_eig_l.as<double>(), d_eig_lam.as<double>(), d_scale.as<double>(), p_fail, st));
            const size_t ntj = set ? nt1 : nt0, t_at = set ? nt0 : 0;
            if (nf || ntj) {
                FactoredParams fp{};
                fp.codons = d_codons.as<uint8_t>();
                fp.leaf_stride = n_sites_total;
                fp.program = c->d_program.as<uint32_t>();
                fp.n_ops = static_cast<int>(c->schedule.program.size());
                fp.slot_branch = d_slot_branch.as<int32_t>();
                fp.op_leaves = c->d_op_meta.ptr ? c->d_op_leaves.as<uint64_t>() : nullptr;
                fp.op_meta = c->d_op_meta.ptr ? c->d_op_meta.as<uint32_t>() : nullptr;
                fp.n_edges = ne;
                fp.n_leaves = nl;
                fp.n_branches = static_cast<uint32_t>(n - 1);
                fp.eig_right = d_eig_r.as<double>();
                fp.eig_left = d_eig_l.as<double>();
                fp.eig_right_frag = use_tiles ? nullptr : d_eig_rf.as<double>();
                fp.eig_left_frag = use_tiles ? nullptr : d_eig_lf.as<double>();
                fp.eig_w = d_eig_w.as<double>();
                fp.prior = d_prior.as<double>();
                fp.anc_weight = d_ancw.as<double>();
                fp.edge_scale = d_scale.as<double>();
                fp.out_lnz = p_lnz;
                fp.out_anc = want_anc ? p_anc : nullptr;
                fp.deep = c->schedule.stack_depth > MAX_STACK_DEPTH ? 1u : 0u;
                fp.stamps = d_fact_stamps.ptr ? d_fact_stamps.as<unsigned long long>() : nullptr;
                fp.stamp_classes = fact_stamp_classes ? 1u : 0u;
                fp.product_count = d_products.as<unsigned long long>();
                if (use_tiles) {
                    fp.eval_x = d_eval_x.as<double>();
                    fp.eig_lambda = d_eig_lam.as<double>();
                    fp.eig_c = d_eig_c.as<double>();
                    fp.branch_len = d_branch_len.as<float>();
                    fp.longest_branch = longest_branch;
                    fp.fail = p_fail;
                }
                if (ntj) {
                    fp.jobs = p_tjobs + t_at;
                    fp.n_jobs = static_cast<uint32_t>(ntj);
                    fp.groups = p_tgroups + (set ? ng0 : 0);
                    fp.n_groups = static_cast<uint32_t>(set ? ng1 : ng0);
                    for (int k = 0; k < 3; ++k) fp.group_counts[k] = 0;
                    for (const TileGroup& tg : tgroups[set])
                        if (tg.n_jobs >= 1 && tg.n_jobs <= 3) ++fp.group_counts[tg.n_jobs - 1];
                    HIP_TRY(launch_prune_tiles(fp, st));
                }
                if (nf) {
                    fp.jobs = p_fjobs + f_at;
                    fp.n_jobs = static_cast<uint32_t>(nf);
                    fp.groups = nullptr;
                    fp.n_groups = 0;
                    HIP_TRY(launch_prune_factored(fp, st));
                }
            }
        };
        if (use_side) {
            if (!eig_jobs.empty() && !host_eigen) {
                EigenParams gp{};
                gp.jobs = p_eig_jobs;
                gp.n_jobs = static_cast<uint32_t>(eig_jobs.size());
                gp.eig_right = d_eig_r.as<double>();
                gp.eig_left = d_eig_l.as<double>();
                gp.eig_right_frag = use_tiles ? nullptr : d_eig_rf.as<double>();
                gp.eig_left_frag = use_tiles ? nullptr : d_eig_lf.as<double>();
                gp.eig_lambda = d_eig_lam.as<double>();
                gp.eig_basis = d_eig_basis.as<double>();
                gp.prior = d_prior.as<double>();
                gp.anc_weight = d_ancw.as<double>();
                gp.eig_w = d_eig_w.as<double>();
                gp.eig_c = d_eig_c.as<double>();
                gp.fail = p_fail;
                HIP_TRY(launch_eigen_omega(gp, side));
            }
            expm_and_prune(1, side);
            HIP_TRY(hipEventRecord(ev_side, side));
        }
        expm_and_prune(0, stream);
        if (use_side) HIP_TRY(hipStreamWaitEvent(stream, ev_side, 0));
        {   // one download: [lnz | anc | fail]
            const size_t out_bytes = aligned(out_used * 8), anc_bytes = want_anc ? out_bytes : 0, all = out_bytes + anc_bytes + tables * sizeof(uint32_t);
            HIP_TRY(h_out.reserve(all));
            HIP_TRY(hipMemcpyAsync(h_out.ptr, d_out.ptr, all, hipMemcpyDeviceToHost, stream));
            HIP_TRY(hipStreamSynchronize(stream));
            const char* h = static_cast<const char*>(h_out.ptr);
            if (out_used) {
                lnz.assign(reinterpret_cast<const double*>(h), reinterpret_cast<const double*>(h) + out_used);
                if (want_anc) anc.assign(reinterpret_cast<const double*>(h + out_bytes), reinterpret_cast<const double*>(h + out_bytes) + out_used);
            }
            fail.assign(reinterpret_cast<const uint32_t*>(h + out_bytes + anc_bytes), reinterpret_cast<const uint32_t*>(h + out_bytes + anc_bytes) + tables);
        }
        done_evals.swap(evals);
        evals.clear();
        for (int set = 0; set < 2; ++set) {
            jobs[set].clear();
            wg0[set].clear();
            wg1[set].clear();
            wgo[set].clear();
            wgt[set].clear();
            wgp[set].clear();
            fjobs[set].clear();
            tjobs[set].clear();
            tgroups[set].clear();
            fevals[set].clear();
        }
        eig_jobs.clear();
        out_used = 0;
    }
    std::vector<Eval> done_evals;  // the evaluations of the last run()

    bool failed(size_t k) const { return fail[done_evals[k].table] != 0; }
    // sum over the block's sites in site order, like lpr += log_z (src/fixed_lik.hpp:431-432)
    double value(size_t k, const std::vector<double>& v) const
    {
        const Eval& e = done_evals[k];
        double s = 0.0;
        for (uint64_t i = 0, cnt = off[e.block + 1] - off[e.block]; i < cnt; ++i) s += v[e.out0 + i];
        return s;
    }
};

inline float decibans(double a, double b) { return static_cast<float>(10.0 * (a - b) / std::log(10.0)); }

// ---- FIXED -------------------------------------------------------------------------------------
int run_fixed(pcsf_ctx* c, const uint8_t* codons, const uint64_t* off, size_t n_blocks, size_t leaf_stride, int compute_anc,
              float* out_phylo, float* out_anc)
{
    const size_t n_sites = off[n_blocks];
    std::vector<double> lnz(2 * std::max<size_t>(n_sites, 1)), anc(2 * std::max<size_t>(n_sites, 1));
    if (n_sites > 0) {
        const size_t nl = static_cast<size_t>(c->n_leaves);
        HIP_TRY(c->d_codons.reserve(nl * n_sites));
        HIP_TRY(c->d_lnz.reserve(2 * n_sites * sizeof(double)));
        HIP_TRY(c->d_anc.reserve(2 * n_sites * sizeof(double)));
        HIP_TRY(upload_rows(c->d_codons.ptr, codons, leaf_stride, n_sites, nl, c->stream));
        const int rc = launch_fixed(c, c->d_codons.as<uint8_t>(), n_sites, n_sites, nullptr, c->d_lnz.as<double>(),
                                    compute_anc ? c->d_anc.as<double>() : nullptr, c->stream);
        if (rc != PCSF_OK) return rc;
        HIP_TRY(hipMemcpyAsync(lnz.data(), c->d_lnz.ptr, 2 * n_sites * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        if (compute_anc) HIP_TRY(hipMemcpyAsync(anc.data(), c->d_anc.ptr, 2 * n_sites * sizeof(double), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    for (size_t b = 0; b < n_blocks; ++b) {
        double l[2] = {0, 0}, a[2] = {0, 0};
        for (int m = 0; m < 2; ++m)
            for (uint64_t s = off[b]; s < off[b + 1]; ++s) {
                l[m] += lnz[m * n_sites + s];
                if (compute_anc) a[m] += anc[m * n_sites + s];
            }
        out_phylo[b] = decibans(l[0], l[1]);
        if (out_anc) out_anc[b] = compute_anc ? decibans(a[0], a[1]) : decibans(0.0, 0.0);
    }
    return PCSF_OK;
}

// ---- MLE / OMEGA: blocks as chains of rounds ------------------------------------------------------------------------------------
// A block's optimisation alternates between its machine (machine.hpp: which points next?) and the evaluation of those points
// (eigen-systems -> edge scales -> pruning).  Two engines drive the same machines:
//   device (the default): the machines live in HBM and run in chain.hip's advance_blocks; per round the host reads 64 bytes of counters
//       and launches the round's kernels with the job lists the device wrote.  Nothing of a block's search crosses PCIe.
//   host (PCSF_BLOCK_ENGINE=host, rounds 1-3's lock-step engine): the machines run on host threads, the evaluations are queued through
//       BlockEngine::queue / run with one upload and one download blob per round.  Kept as the A/B reference; same bits.

// the restart points of one search range: libstdc++'s generator and distribution, the host's exp / log -- what MaxLik1D drew itself
// until round 3 (optimizer.hpp DrawTable)
std::vector<double> make_draw_table(uint32_t seed, double lo, double hi)
{
    std::vector<double> t(DRAW_TABLE_ENTRIES);
    std::mt19937 g(seed);
    const double width = std::log(hi) - std::log(lo);
    for (double& x : t) {
        std::uniform_real_distribution<> dis(0.0, width);
        const double r = dis(g);
        x = std::exp(std::log(lo) + r);
    }
    return t;
}

inline bool use_device_engine()
{
    const char* v = std::getenv("PCSF_BLOCK_ENGINE");
    return !(v && std::string(v) == "host");
}
// Random restarts are evaluated ahead of knowing that a search needs any only for short lists: with thousands of blocks a round is
// bound by its work, not its latency, and a third of OMEGA's evaluations were such restarts that no optimiser ever asked for
// (PCSF_EAGER_RESTARTS_BELOW overrides).  Decided from the length of the CALL's list; no result depends on it.
inline size_t eager_restarts_below()
{
    static const size_t v = [] {
        const char* e = std::getenv("PCSF_EAGER_RESTARTS_BELOW");
        return e ? static_cast<size_t>(std::max(0L, std::atol(e))) : size_t(1024);
    }();
    return v;
}

struct ChainStats {
    size_t rounds = 0, evals = 0, eig_jobs = 0, tile_jobs = 0;
    double t_setup = 0.0, t_rounds = 0.0, t_wait = 0.0;
};

// Groups of blocks run as concurrent engines (one host thread each).  hipMalloc while another group's kernels run takes tens of
// milliseconds per gigabyte (8,256 blocks, MLE: 0.24 s of set-up in the second group while the first was already in its rounds), so
// every group allocates first and the rounds start together.
struct StartGate {
    std::mutex m;
    std::condition_variable cv;
    size_t expected = 0, arrived = 0;
    void arrive_and_wait()
    {
        std::unique_lock<std::mutex> g(m);
        if (++arrived >= expected) cv.notify_all();
        else cv.wait(g, [&] { return arrived >= expected; });
    }
};
struct GateTicket {   // a group that fails before its rounds must not leave the others waiting
    StartGate* gate;
    bool used = false;
    explicit GateTicket(StartGate* g) : gate(g) {}
    void pass()
    {
        if (gate && !used) {
            used = true;
            gate->arrive_and_wait();
        }
    }
    ~GateTicket()
    {
        if (gate && !used) {
            std::lock_guard<std::mutex> g(gate->m);
            ++gate->arrived;
            gate->cv.notify_all();
        }
    }
};

// The device engine: `machines` (initialised by the caller) are uploaded, advanced on the device until none is active, and downloaded.
template <class Machine>
void run_chain(BlockEngine& eng, std::vector<Machine>& machines, int algo, bool want_anc, uint32_t seed, bool eager, uint32_t transient0, bool held_systems,
               const std::atomic<bool>* stop, StartGate* gate, ChainStats& st)
{
    GateTicket ticket(gate);
    const auto t_enter = std::chrono::steady_clock::now();
    constexpr bool MLE = std::is_same<Machine, MleMachine>::value;
    (void)algo;
    const LookaheadPolicy look = lookahead_policy(eager);
    const uint32_t P = static_cast<uint32_t>(MLE ? 2 * look.max_points : look.max_points);   // evaluation slots per block
    const size_t nb = eng.n_blocks, n_sites = eng.n_sites_total;
    hipStream_t st0 = eng.stream, st1 = eng.side;
    // PCSF_CONTROL_STREAM=1 (experiment, off by default): the small kernels of a round -- advance_blocks, the eigen-solver, the edge scales --
    // on a control stream of the highest priority, the pruning launches waiting for events.  The idea: on the engine's own streams those
    // kernels queue behind the OTHER group's pruning work-groups, which fill every SIMD (2 x 250 registers) and leave no room beside them
    // (advance_blocks 0.40 ms and edge_scale_kernel 0.36 ms per launch on BASELINE.json configs[2] against 0.01-0.03 alone).  Measured: no
    // gain -- the dispatcher does not hand a freed wave slot to the higher-priority queue first, the kernels take as long, and the events
    // cost 2-5 % (configs[2] OMEGA 1.41 -> 1.43 s, 8,256 real blocks 0.95 -> 1.06 s; profiles/r04/control_stream.txt).
    struct ControlStream {
        pcsf_ctx* c;
        hipStream_t s = nullptr;
        hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};   // edge scales of set 0 / 1 done; pruning of set 0 / 1 done
        ~ControlStream()
        {
            if (s) {
                (void)hipStreamSynchronize(s);
                c->control_streams.release(s);
            }
            for (hipEvent_t e : ev)
                if (e) (void)hipEventDestroy(e);
        }
    } ctl{eng.c};
    const bool use_ctl = std::getenv("PCSF_CONTROL_STREAM") && std::atoi(std::getenv("PCSF_CONTROL_STREAM")) != 0;
    if (use_ctl) {
        HIP_TRY(eng.c->control_streams.acquire(&ctl.s));
        for (hipEvent_t& e : ctl.ev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    }
    hipStream_t stc = use_ctl ? ctl.s : st0;
    size_t total_tiles = 0;
    for (size_t b = 0; b < nb; ++b) total_tiles += (eng.off[b + 1] - eng.off[b] + 15) / 16;
    DeviceBuffer d_machines, d_stores, d_off, d_draws, d_counters, d_evals, d_eig_jobs, d_jobs, d_groups, d_lnz, d_anc, d_fail, d_eval_x;
    BlockEngine::PinnedBuffer h_counters;
    const size_t eval_cap = nb * P, job_cap = std::max<size_t>(1, total_tiles) * P, group_cap = job_cap;
    HIP_TRY(d_machines.reserve(nb * sizeof(Machine)));
    HIP_TRY(d_stores.reserve(nb * (MLE ? 2 : 1) * STORE_ENTRIES * sizeof(StoreEntry)));
    HIP_TRY(d_off.reserve((nb + 1) * sizeof(uint64_t)));
    HIP_TRY(d_draws.reserve(3 * DRAW_TABLE_ENTRIES * sizeof(double)));
    HIP_TRY(d_counters.reserve(sizeof(ChainCounters)));
    HIP_TRY(h_counters.reserve(sizeof(ChainCounters)));
    HIP_TRY(d_evals.reserve(2 * eval_cap * sizeof(ExpmEval)));
    if (!MLE) HIP_TRY(d_eig_jobs.reserve(eval_cap * sizeof(EigenJob)));
    HIP_TRY(d_jobs.reserve(2 * job_cap * sizeof(FactoredJob)));
    if (eng.use_tiles) HIP_TRY(d_groups.reserve(2 * 4 * group_cap * sizeof(TileGroup)));
    HIP_TRY(d_lnz.reserve(std::max<size_t>(1, n_sites) * P * sizeof(double)));
    if (want_anc) HIP_TRY(d_anc.reserve(std::max<size_t>(1, n_sites) * P * sizeof(double)));
    HIP_TRY(d_fail.reserve(nb * P * sizeof(uint32_t)));
    // PCSF_FACTORED_OWN_SCALES=0: prune_factored reads rows of edge scales written by edge_scale_kernel (rounds 2-4's form; A/B, same bits)
    const bool own_scales = !eng.use_tiles && !([] { const char* v = std::getenv("PCSF_FACTORED_OWN_SCALES"); return v && v[0] == '0'; }());
    if (eng.use_tiles || own_scales) HIP_TRY(d_eval_x.reserve(nb * P * sizeof(double)));   // prune_tiles makes its own edge scales from the evaluation's tree scale
    const std::vector<double> t_mle = make_draw_table(seed, 1e-2, 10.0), t_rho = make_draw_table(seed, 0.001, 10.0), t_kappa = make_draw_table(seed, 1.0, 10.0);
    HIP_TRY(hipMemcpyAsync(d_draws.as<double>(), t_mle.data(), DRAW_TABLE_ENTRIES * sizeof(double), hipMemcpyHostToDevice, st0));
    HIP_TRY(hipMemcpyAsync(d_draws.as<double>() + DRAW_TABLE_ENTRIES, t_rho.data(), DRAW_TABLE_ENTRIES * sizeof(double), hipMemcpyHostToDevice, st0));
    HIP_TRY(hipMemcpyAsync(d_draws.as<double>() + 2 * DRAW_TABLE_ENTRIES, t_kappa.data(), DRAW_TABLE_ENTRIES * sizeof(double), hipMemcpyHostToDevice, st0));
    HIP_TRY(hipMemcpyAsync(d_machines.ptr, machines.data(), nb * sizeof(Machine), hipMemcpyHostToDevice, st0));
    HIP_TRY(hipMemcpyAsync(d_off.ptr, eng.off, (nb + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, st0));
    HIP_TRY(hipMemsetAsync(d_fail.ptr, 0, nb * P * sizeof(uint32_t), st0));
    HIP_TRY(hipStreamSynchronize(st0));   // the draw tables are locals

    ChainParams cp{};
    cp.machines = d_machines.ptr;
    cp.stores = d_stores.as<StoreEntry>();
    cp.off = d_off.as<uint64_t>();
    cp.slots_per_block = P;
    cp.out_lnz = d_lnz.as<double>();
    cp.out_anc = want_anc ? d_anc.as<double>() : nullptr;
    cp.fail = d_fail.as<uint32_t>();
    cp.eval_x = (eng.use_tiles || own_scales) ? d_eval_x.as<double>() : nullptr;
    for (int k = 0; k < 3; ++k) cp.draws[k] = d_draws.as<double>() + k * DRAW_TABLE_ENTRIES;
    cp.n_draws = DRAW_TABLE_ENTRIES;
    cp.eager = eager ? 1u : 0u;
    cp.max_points = static_cast<uint32_t>(look.max_points);
    cp.max_restarts = static_cast<uint32_t>(look.max_restarts);
    cp.held_systems = held_systems ? 1u : 0u;
    cp.transient0 = transient0;
    cp.use_tiles = eng.use_tiles ? 1u : 0u;
    cp.own_scales = own_scales ? 1u : 0u;
    cp.split_sizes = (eng.use_tiles && nb >= 1024) ? 1u : 0u;   // short lists: one launch of full work-groups (a launch that does not fill the chip is bound by one tile's latency)
    if (const char* v = std::getenv("PCSF_SPLIT_SIZES")) cp.split_sizes = (eng.use_tiles && std::atoi(v) != 0) ? 1u : 0u;   // A/B knob
    cp.counters = d_counters.as<ChainCounters>();
    cp.evals = d_evals.as<ExpmEval>();
    cp.eval_capacity = static_cast<uint32_t>(eval_cap);
    cp.eig_jobs = MLE ? nullptr : d_eig_jobs.as<EigenJob>();
    cp.tile_jobs = d_jobs.as<FactoredJob>();
    cp.job_capacity = static_cast<uint32_t>(job_cap);
    cp.groups = eng.use_tiles ? d_groups.as<TileGroup>() : nullptr;
    cp.group_capacity = static_cast<uint32_t>(group_cap);
    // MLE on prune_tiles: every block uses the same two eigen-systems, so the tile jobs of a round go to one list per model and any four
    // consecutive jobs of a list share a work-group (full work-groups whatever the blocks' lengths); the groups are the same every round
    const bool pack_models = MLE && eng.use_tiles;
    cp.pack_models = pack_models ? 1u : 0u;
    cp.list_capacity = static_cast<uint32_t>(job_cap);
    if (pack_models) {
        std::vector<TileGroup> fixed(job_cap / TILE_JOBS_PER_GROUP + 1);
        for (size_t g = 0; g < fixed.size(); ++g) fixed[g] = TileGroup{static_cast<uint32_t>(g * TILE_JOBS_PER_GROUP), TILE_JOBS_PER_GROUP};
        HIP_TRY(hipMemcpyAsync(d_groups.ptr, fixed.data(), fixed.size() * sizeof(TileGroup), hipMemcpyHostToDevice, st0));
        HIP_TRY(hipStreamSynchronize(st0));   // `fixed` is a local
    }

    FactoredParams fp{};
    fp.codons = eng.d_codons.template as<uint8_t>();
    fp.leaf_stride = n_sites;
    fp.program = eng.c->d_program.template as<uint32_t>();
    fp.n_ops = static_cast<int>(eng.c->schedule.program.size());
    fp.slot_branch = eng.d_slot_branch.template as<int32_t>();
    fp.op_leaves = eng.c->d_op_meta.ptr ? eng.c->d_op_leaves.template as<uint64_t>() : nullptr;
    fp.op_meta = eng.c->d_op_meta.ptr ? eng.c->d_op_meta.template as<uint32_t>() : nullptr;
    fp.n_edges = eng.ne;
    fp.n_leaves = eng.nl;
    fp.n_branches = static_cast<uint32_t>(eng.n - 1);
    fp.eig_right = eng.d_eig_r.template as<double>();
    fp.eig_left = eng.d_eig_l.template as<double>();
    fp.eig_right_frag = eng.use_tiles ? nullptr : eng.d_eig_rf.template as<double>();
    fp.eig_left_frag = eng.use_tiles ? nullptr : eng.d_eig_lf.template as<double>();
    fp.eig_w = eng.d_eig_w.template as<double>();
    fp.prior = eng.d_prior.template as<double>();
    fp.anc_weight = eng.d_ancw.template as<double>();
    fp.edge_scale = eng.d_scale.template as<double>();
    fp.out_lnz = d_lnz.as<double>();
    fp.out_anc = want_anc ? d_anc.as<double>() : nullptr;
    fp.deep = eng.c->schedule.stack_depth > MAX_STACK_DEPTH ? 1u : 0u;
    fp.stamps = eng.d_fact_stamps.ptr ? eng.d_fact_stamps.template as<unsigned long long>() : nullptr;
    fp.stamp_classes = eng.fact_stamp_classes ? 1u : 0u;
    fp.product_count = eng.d_products.template as<unsigned long long>();
    if (eng.use_tiles || own_scales) {
        fp.eval_x = d_eval_x.as<double>();
        fp.eig_lambda = eng.d_eig_lam.template as<double>();
        fp.eig_c = eng.d_eig_c.template as<double>();
        fp.branch_len = eng.d_branch_len.template as<float>();
        fp.longest_branch = eng.longest_branch;
        fp.fail = d_fail.as<uint32_t>();
    }
    // prune_factored: every tile's op list compiled once, before the first round (PCSF_FACTORED_COMPILED=0: the work-groups derive it
    // themselves at every evaluation, rounds 2-4's form -- A/B, same bits)
    DeviceBuffer d_tile_desc, d_tile_programs, d_site_tile;
    const bool compile_tiles = !eng.use_tiles && fp.op_meta && fp.n_ops <= 128 && eng.nl <= 128 && total_tiles > 0 &&
                               !([] { const char* v = std::getenv("PCSF_FACTORED_COMPILED"); return v && v[0] == '0'; }());
    if (compile_tiles) {
        std::vector<TileDesc> tiles;
        tiles.reserve(total_tiles);
        for (size_t b = 0; b < nb; ++b)
            for (uint64_t s0 = eng.off[b]; s0 < eng.off[b + 1]; s0 += 16)
                tiles.push_back(TileDesc{s0, static_cast<uint32_t>(std::min<uint64_t>(16, eng.off[b + 1] - s0)), 0u});
        HIP_TRY(d_tile_desc.reserve(tiles.size() * sizeof(TileDesc)));
        HIP_TRY(d_tile_programs.reserve(tiles.size() * TILE_PROGRAM_WORDS * sizeof(uint32_t)));
        HIP_TRY(d_site_tile.reserve(std::max<size_t>(1, n_sites) * sizeof(uint32_t)));
        HIP_TRY(hipMemcpyAsync(d_tile_desc.ptr, tiles.data(), tiles.size() * sizeof(TileDesc), hipMemcpyHostToDevice, st0));
        HIP_TRY(launch_compile_tiles(fp, d_tile_desc.as<TileDesc>(), static_cast<uint32_t>(tiles.size()), d_tile_programs.as<uint32_t>(), d_site_tile.as<uint32_t>(), st0));
        HIP_TRY(hipStreamSynchronize(st0));   // `tiles` is a local; every stream of the rounds reads the programs
        fp.tile_programs = d_tile_programs.as<uint32_t>();
        fp.site_tile = d_site_tile.as<uint32_t>();
    }
    EigenParams gp{};
    gp.jobs = cp.eig_jobs;
    gp.eig_right = eng.d_eig_r.template as<double>();
    gp.eig_left = eng.d_eig_l.template as<double>();
    gp.eig_right_frag = eng.use_tiles ? nullptr : eng.d_eig_rf.template as<double>();
    gp.eig_left_frag = eng.use_tiles ? nullptr : eng.d_eig_lf.template as<double>();
    gp.eig_lambda = eng.d_eig_lam.template as<double>();
    gp.eig_basis = eng.d_eig_basis.template as<double>();
    gp.prior = eng.d_prior.template as<double>();
    gp.anc_weight = eng.d_ancw.template as<double>();
    gp.eig_w = eng.d_eig_w.template as<double>();
    gp.eig_c = eng.d_eig_c.template as<double>();
    gp.fail = d_fail.as<uint32_t>();
    // OMEGA: the eigen-systems a block holds sit in slots [0, transient0) (with their bases, for warm starts); the probes of a round that
    // are not held (random restarts) take the next slots of a transient range that is dense per round -- one slot per block to begin with,
    // grown when a round needs more (the host knows before it launches the solver).  Twelve slots per block "for the worst case" were
    // 1.2 MB a block, and on boxes whose driver clears memory at allocation 19 GB of hipMalloc cost 0.7 s of a 1.5 s call.
    size_t transient_cap = MLE ? 0 : nb + 64;
    auto eigen_pointers = [&]() {
        fp.eig_right = eng.have_right ? eng.d_eig_r.template as<double>() : nullptr;
        fp.eig_left = eng.have_left ? eng.d_eig_l.template as<double>() : nullptr;
        fp.eig_right_frag = eng.use_tiles ? nullptr : eng.d_eig_rf.template as<double>();
        fp.eig_left_frag = eng.use_tiles ? nullptr : eng.d_eig_lf.template as<double>();
        fp.eig_w = eng.d_eig_w.template as<double>();
        fp.prior = eng.d_prior.template as<double>();
        fp.anc_weight = eng.d_ancw.template as<double>();
        fp.eig_lambda = (eng.use_tiles || own_scales) ? eng.d_eig_lam.template as<double>() : nullptr;
        fp.eig_c = (eng.use_tiles || own_scales) ? eng.d_eig_c.template as<double>() : nullptr;
        gp.eig_right = eng.have_right ? eng.d_eig_r.template as<double>() : nullptr;
        gp.eig_left = eng.have_left ? eng.d_eig_l.template as<double>() : nullptr;
        gp.eig_right_frag = eng.use_tiles ? nullptr : eng.d_eig_rf.template as<double>();
        gp.eig_left_frag = eng.use_tiles ? nullptr : eng.d_eig_lf.template as<double>();
        gp.eig_lambda = eng.d_eig_lam.template as<double>();
        gp.eig_basis = eng.d_eig_basis.template as<double>();
        gp.basis_slots = MLE ? 0u : transient0;
        gp.prior = eng.d_prior.template as<double>();
        gp.anc_weight = eng.d_ancw.template as<double>();
        gp.eig_w = eng.d_eig_w.template as<double>();
        gp.eig_c = eng.d_eig_c.template as<double>();
    };
    if (!MLE) eng.reserve_eig(transient0 + transient_cap, transient0, /*with_left=*/!eng.use_tiles && !own_scales, /*with_right=*/!own_scales);
    eigen_pointers();
    // prune_factored's rows of edge scales are dense per round too (a row per evaluation of the round, in list order)
    size_t row_cap = (eng.use_tiles || own_scales) ? 0 : 2 * nb + 64;
    if (row_cap) eng.reserve_tables(row_cap);
    const size_t row_doubles = static_cast<size_t>(eng.n - 1) * 64;
    st.t_setup = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_enter).count();
    ticket.pass();   // every group of the call has allocated

    // the round's evaluations of one set: edge scales (on `scale_stream`), then pruning (on `stream`)
    auto launch_scales = [&](int set, const ChainCounters& cnt, hipStream_t stream) {
        if (eng.use_tiles || own_scales) return;   // the pruning kernel makes its own edge scales and runs the row-sum check itself
        double* rows = eng.d_scale.template as<double>() + (set ? static_cast<size_t>(cnt.n_evals[0]) * row_doubles : 0);   // set 0's rows, then set 1's
        HIP_TRY(launch_edge_scale(cp.evals + static_cast<size_t>(set) * eval_cap, cnt.n_evals[set], static_cast<uint32_t>(eng.n - 1), eng.d_branch_len.template as<float>(),
                                  eng.d_eig_r.template as<double>(), eng.d_eig_l.template as<double>(), eng.d_eig_lam.template as<double>(), rows,
                                  d_fail.as<uint32_t>(), stream, /*dense_rows=*/true));
    };
    auto launch_pruning = [&](int set, const ChainCounters& cnt, hipStream_t stream) {
        if (pack_models) {
            for (int model = 0; model < 2 && set == 0; ++model) {
                if (!cnt.n_list[model]) continue;
                st.tile_jobs += cnt.n_list[model];
                fp.jobs = cp.tile_jobs + static_cast<size_t>(model) * cp.list_capacity;
                fp.n_jobs = cnt.n_list[model];
                fp.groups = cp.groups;
                fp.n_groups = (cnt.n_list[model] + TILE_JOBS_PER_GROUP - 1) / TILE_JOBS_PER_GROUP;
                HIP_TRY(launch_prune_tiles_sized(fp, static_cast<int>(TILE_JOBS_PER_GROUP), stream));
            }
            return;
        }
        if (!cnt.n_tile_jobs[set]) return;
        st.tile_jobs += cnt.n_tile_jobs[set];
        fp.jobs = cp.tile_jobs + static_cast<size_t>(set) * job_cap;
        fp.n_jobs = cnt.n_tile_jobs[set];
        if (eng.use_tiles) {
            for (int list = 3; list >= 0; --list) {   // largest work-groups first (tiles.hip launch_tiles_impl)
                if (!cnt.n_groups[set][list]) continue;
                fp.groups = cp.groups + (static_cast<size_t>(set) * 4 + list) * group_cap;
                fp.n_groups = cnt.n_groups[set][list];
                HIP_TRY(launch_prune_tiles_sized(fp, cp.split_sizes ? list + 1 : static_cast<int>(TILE_JOBS_PER_GROUP), stream));
            }
        } else {
            fp.groups = nullptr;
            fp.n_groups = 0;
            fp.edge_scale = eng.d_scale.template as<double>() + (set ? static_cast<size_t>(cnt.n_evals[0]) * row_doubles : 0);   // FactoredJob::eval = row within the set
            HIP_TRY(launch_prune_factored(fp, stream));
        }
    };
    const auto t_loop = std::chrono::steady_clock::now();
    const bool trace_rounds = std::getenv("PCSF_TRACE_ROUNDS") != nullptr;   // diagnostics: one line per round
    const ChainCounters* hc = static_cast<const ChainCounters*>(h_counters.ptr);
    bool pruned[2] = {false, false};   // the previous round launched pruning of the set (its event is recorded)
    for (;;) {
        if (stop && stop->load(std::memory_order_relaxed)) throw std::runtime_error("another group of this call failed");
        // advance_blocks reads what the pruning kernels of the previous round wrote
        if (use_ctl) {
            for (int set = 0; set < 2; ++set)
                if (pruned[set]) HIP_TRY(hipStreamWaitEvent(stc, ctl.ev[2 + set], 0));
        }
        HIP_TRY(hipMemsetAsync(d_counters.ptr, 0, sizeof(ChainCounters), stc));
        HIP_TRY(launch_advance_blocks(cp, MLE ? 0 : 2, static_cast<uint32_t>(nb), stc));
        HIP_TRY(hipMemcpyAsync(h_counters.ptr, d_counters.ptr, sizeof(ChainCounters), hipMemcpyDeviceToHost, stc));
        const auto t_w = std::chrono::steady_clock::now();
        HIP_TRY(hipStreamSynchronize(stc));
        st.t_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_w).count();
        const ChainCounters cnt = *hc;
        if (trace_rounds)
            fprintf(stderr, "round %zu at %.3f ms: %u blocks active, %u + %u evaluations, %u eigen-systems, %u + %u tile jobs (waited %.3f ms for the previous round)\n", st.rounds,
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - t_loop).count() * 1e3, cnt.n_active, cnt.n_evals[0], cnt.n_evals[1], cnt.n_eig_jobs,
                    cnt.n_tile_jobs[0] + cnt.n_list[0] + cnt.n_list[1], cnt.n_tile_jobs[1], std::chrono::duration<double>(std::chrono::steady_clock::now() - t_w).count() * 1e3);
        if (cnt.n_active == 0) break;
        ++st.rounds;
        st.evals += cnt.n_evals[0] + cnt.n_evals[1];
        st.eig_jobs += cnt.n_eig_jobs;
        // An ASSERTION of an invariant, not a guard: the capacities are worst cases by construction (every block asks for at most P
        // evaluations per round: nb * P evaluations and eigen jobs, total_tiles * P tile jobs and groups), and advance_blocks has already
        // written the round's records when these counters are read.
        bool overflow = cnt.n_evals[0] > eval_cap || cnt.n_evals[1] > eval_cap || cnt.n_tile_jobs[0] > job_cap || cnt.n_tile_jobs[1] > job_cap ||
                        cnt.n_list[0] > job_cap || cnt.n_list[1] > job_cap || cnt.n_eig_jobs > eval_cap;
        for (int set = 0; set < 2; ++set)
            for (int sz = 0; sz < 4; ++sz) overflow = overflow || cnt.n_groups[set][sz] > group_cap;
        if (overflow) throw std::runtime_error("advance_blocks: a job list overflowed (an invariant of the engine's capacities is broken)");
        if (cnt.n_eig_jobs && eng.host_eigen) throw std::runtime_error("PCSF_EIGEN_HOST needs PCSF_BLOCK_ENGINE=host");
        // everything of the previous round has finished (advance_blocks waited for it): buffers may grow here
        if (cnt.n_transient > transient_cap) {
            transient_cap = static_cast<size_t>(cnt.n_transient) + cnt.n_transient / 2 + 64;
            eng.reserve_eig(transient0 + transient_cap, transient0, /*with_left=*/!eng.use_tiles && !own_scales, /*with_right=*/!own_scales);
            eigen_pointers();
        }
        if (!eng.use_tiles && !own_scales && static_cast<size_t>(cnt.n_evals[0]) + cnt.n_evals[1] > row_cap) {
            row_cap = (static_cast<size_t>(cnt.n_evals[0]) + cnt.n_evals[1]) * 3 / 2 + 64;
            eng.reserve_tables(row_cap);
        }
        gp.n_jobs = cnt.n_eig_jobs;
        if (use_ctl) {
            // control stream: edge scales of set 0, the eigen-systems, edge scales of set 1; the pruning launches wait for their set's scales
            pruned[0] = pruned[1] = false;
            if (cnt.n_evals[0]) {
                launch_scales(0, cnt, stc);
                HIP_TRY(hipEventRecord(ctl.ev[0], stc));
                HIP_TRY(hipStreamWaitEvent(st0, ctl.ev[0], 0));
                launch_pruning(0, cnt, st0);
                HIP_TRY(hipEventRecord(ctl.ev[2], st0));
                pruned[0] = true;
            }
            if (cnt.n_evals[1]) {
                if (cnt.n_eig_jobs) HIP_TRY(launch_eigen_omega(gp, stc));
                launch_scales(1, cnt, stc);
                HIP_TRY(hipEventRecord(ctl.ev[1], stc));
                HIP_TRY(hipStreamWaitEvent(st1, ctl.ev[1], 0));
                launch_pruning(1, cnt, st1);
                HIP_TRY(hipEventRecord(ctl.ev[3], st1));
                pruned[1] = true;
            }
        } else {
            // set 1 on the side stream: the eigen-solver is bound by the latency of its slowest job and leaves most of the chip to set 0
            if (cnt.n_evals[1]) {
                if (cnt.n_eig_jobs) HIP_TRY(launch_eigen_omega(gp, st1));
                launch_scales(1, cnt, st1);
                launch_pruning(1, cnt, st1);
                HIP_TRY(hipEventRecord(eng.ev_side, st1));
            }
            launch_scales(0, cnt, st0);
            launch_pruning(0, cnt, st0);
            if (cnt.n_evals[1]) HIP_TRY(hipStreamWaitEvent(st0, eng.ev_side, 0));   // the next advance_blocks reads both sets' results
        }
    }
    st.t_rounds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_loop).count();
    HIP_TRY(hipMemcpyAsync(machines.data(), d_machines.ptr, nb * sizeof(Machine), hipMemcpyDeviceToHost, stc));
    HIP_TRY(hipStreamSynchronize(stc));
}

// The host engine: the same machines, advanced on host threads; a round's requests go through BlockEngine::queue / run.
// prior values (OMEGA) come from the device's math library like the device engine's (chain.hip), so that both add the same bits.
struct HostRound {
    std::vector<EvalRequest> req;      // [n_blocks][max requests]
    std::vector<int> n_req;
    std::vector<size_t> eval0;         // index of the block's first evaluation in the engine's run
    std::vector<EvalResult> res;       // [n_blocks][max requests]
};

template <class Machine, class Before, class Advance>
void run_host_rounds(BlockEngine& eng, std::vector<Machine>& machines, uint32_t max_req, bool want_anc, const std::atomic<bool>* stop, ChainStats& st, Before before_round,
                     Advance advance)
{
    const size_t nb = eng.n_blocks;
    eng.reserve_tables(std::max<size_t>(64, nb * max_req));
    HostRound hr;
    hr.req.resize(nb * max_req);
    hr.n_req.assign(nb, 0);
    hr.eval0.assign(nb, 0);
    hr.res.resize(nb * max_req);
    const auto t_loop = std::chrono::steady_clock::now();
    for (;;) {
        if (stop && stop->load(std::memory_order_relaxed)) throw std::runtime_error("another group of this call failed");
        before_round();
        parallel_for(nb, [&](size_t b) { hr.n_req[b] = advance(b, hr.res.data() + b * max_req, hr.req.data() + b * max_req); });
        size_t active = 0, queued = 0;
        for (size_t b = 0; b < nb; ++b) {
            active += machines[b].active;
            hr.eval0[b] = queued;
            for (int i = 0; i < hr.n_req[b]; ++i, ++queued) {
                const EvalRequest& r = hr.req[b * max_req + i];
                eng.queue(b, r.eig, r.tree_scale, r.new_eigen == 1 ? r.qs : nullptr, r.new_eigen == 2, r.warm);
                st.eig_jobs += r.new_eigen == 1;
            }
        }
        if (!active) break;
        ++st.rounds;
        st.evals += queued;
        const auto t_w = std::chrono::steady_clock::now();
        eng.run(want_anc);
        st.t_wait += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_w).count();
        parallel_for(nb, [&](size_t b) {
            for (int i = 0; i < hr.n_req[b]; ++i) {
                EvalResult& r = hr.res[b * max_req + i];
                const size_t k = hr.eval0[b] + i;
                r.lnz = eng.value(k, eng.lnz);
                r.anc = want_anc ? eng.value(k, eng.anc) : 0.0;
                r.failed = eng.failed(k) ? 1u : 0u;
            }
        });
    }
    st.t_rounds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_loop).count();
}

void print_chain_stats(const char* what, size_t n_blocks, bool device, bool tiles, const ChainStats& st, double whole, unsigned long long products)
{
    // executed matrix work of the pruning kernels: a product is 64 x v_mfma_f64_16x16x4_f64 = 131,072 FLOP (the eigen-solver's warm starts and
    // the element-wise work are not counted)
    fprintf(stderr, "%s: pruning products %llu = %.4g TFLOP executed on the matrix cores\n", what, products, static_cast<double>(products) * 131072.0 * 1e-12);
    fprintf(stderr, "%s: %zu blocks, %s engine, %s: %zu rounds, %zu evaluations (%zu with a new eigen-system): set-up %.3f s, rounds %.3f s (of which waiting for the device %.3f s), whole call %.3f s\n",
            what, n_blocks, device ? "device" : "host", tiles ? "prune_tiles" : "prune_factored", st.rounds, st.evals, st.eig_jobs, st.t_setup, st.t_rounds, st.t_wait, whole);
}

// ---- MLE ---------------------------------------------------------------------------------------
int run_mle(pcsf_ctx* c, const uint8_t* codons, const uint64_t* off, size_t n_blocks, size_t leaf_stride, int compute_anc,
            uint32_t seed, float* out_phylo, float* out_anc, bool tiles, bool own_streams, bool eager, const std::atomic<bool>* stop, StartGate* gate)
{
    const auto t_enter = std::chrono::steady_clock::now();
    GateTicket ticket(gate);   // (the host engine does not wait at the gate, but must not leave others waiting)
    BlockEngine eng(c, codons, off, n_blocks, leaf_stride, tiles, own_streams);
    eng.reserve_eig(2);
    eng.upload_eig(0, c->eig[0], c->pi[0].data());
    eng.upload_eig(1, c->eig[1], c->pi[1].data());
    std::vector<MleMachine> machines(n_blocks);
    for (size_t b = 0; b < n_blocks; ++b) machines[b].init(static_cast<uint32_t>(off[b + 1] - off[b]));
    ChainStats st;
    const bool device = use_device_engine() && !eng.host_eigen;
    if (device) {
        ticket.used = true;   // run_chain holds the ticket from here
        run_chain(eng, machines, PCSF_MLE, compute_anc != 0, seed, eager, 0u, true, stop, gate, st);
    } else {
        const std::vector<double> table = make_draw_table(seed, 1e-2, 10.0);
        std::vector<StoreEntry> stores(n_blocks * 2 * STORE_ENTRIES);
        const DrawTable dt{table.data(), DRAW_TABLE_ENTRIES};
        const LookaheadPolicy look = lookahead_policy(eager);
        run_host_rounds(eng, machines, 2 * look.max_points, compute_anc != 0, stop, st, [] {}, [&](size_t b, const EvalResult* res, EvalRequest* out) {
            Lookahead ahead[2];
            return machines[b].round(res, stores.data() + b * 2 * STORE_ENTRIES, STORE_ENTRIES, dt, look.eager, look.max_points, look.max_restarts, compute_anc != 0, out, ahead);
        });
    }
    if (std::getenv("PCSF_TIME_BLOCKS")) {
        size_t used = 0, unused = 0;
        for (const MleMachine& m : machines) {
            used += m.used;
            unused += m.unused;
        }
        fprintf(stderr, "run_mle: %zu evaluations consumed by the optimisers, %zu speculative ones never asked for\n", used, unused);
        print_chain_stats("run_mle", n_blocks, device, eng.use_tiles, st, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_enter).count(), eng.products_run());
    }
    for (size_t b = 0; b < n_blocks; ++b) {
        const MleMachine& m = machines[b];
        if (m.failed) {
            out_phylo[b] = NAN;
            if (out_anc) out_anc[b] = NAN;
            continue;
        }
        out_phylo[b] = decibans(m.opt[0].last_lpr(), m.opt[1].last_lpr());
        if (out_anc) out_anc[b] = decibans(m.opt[0].last_anc(), m.opt[1].last_anc());
    }
    return PCSF_OK;
}

// ---- OMEGA -------------------------------------------------------------------------------------
int run_omega(pcsf_ctx* c, const uint8_t* codons, const uint64_t* off, size_t n_blocks, size_t leaf_stride, uint32_t seed,
              float* out_phylo, float* out_anc, bool tiles, bool own_streams, bool eager, const std::atomic<bool>* stop, StartGate* gate)
{
    const auto t_enter = std::chrono::steady_clock::now();
    GateTicket ticket(gate);
    BlockEngine eng(c, codons, off, n_blocks, leaf_stride, tiles, own_streams);
    // eigen-system slots [role * n_blocks + b]: what block b holds (EigRole; roles are swapped between them, see OmegaMachine);
    // slots N_ROLES * n_blocks + b * max_points + k: the probes of a round that are not kept (random restarts)
    const bool held_systems = !(std::getenv("PCSF_OMEGA_HELD_SYSTEMS") && std::atoi(std::getenv("PCSF_OMEGA_HELD_SYSTEMS")) == 0);
    const size_t transient0 = N_ROLES * n_blocks;
    const LookaheadPolicy look = lookahead_policy(eager);
    const size_t nl = static_cast<size_t>(c->n_leaves);
    std::vector<OmegaMachine> machines(n_blocks);
    // F3x4 nucleotide counts over every non-gap codon of every row (src/run.hpp:105-134)
    parallel_for(n_blocks, [&](size_t b) {
        double cnt[3][4];
        for (auto& r : cnt)
            for (double& v : r) v = 1.0;
        for (size_t l = 0; l < nl; ++l)
            for (uint64_t s = off[b]; s < off[b + 1]; ++s) {
                const uint8_t cd = codons[l * leaf_stride + s];
                if (cd < 64) {
                    cnt[0][cd >> 4] += 1;
                    cnt[1][(cd >> 2) & 3] += 1;
                    cnt[2][cd & 3] += 1;
                }
            }
        double ratios[9];
        for (int p = 0; p < 3; ++p)
            for (int j = 0; j < 3; ++j) ratios[3 * p + j] = cnt[p][j] / cnt[p][3];
        uint32_t slots[N_ROLES];
        for (int r = 0; r < N_ROLES; ++r) slots[r] = static_cast<uint32_t>(r * n_blocks + b);
        machines[b].init(static_cast<uint32_t>(off[b + 1] - off[b]), slots, ratios);
    });
    ChainStats st;
    const bool device = use_device_engine() && !eng.host_eigen;
    if (device) {
        ticket.used = true;
        run_chain(eng, machines, PCSF_OMEGA, false, seed, eager, static_cast<uint32_t>(transient0), held_systems, stop, gate, st);
    } else {
        eng.reserve_eig(transient0 + n_blocks * look.max_points);   // (the A/B engine: every block its own transient slots)
        const std::vector<double> t_rho = make_draw_table(seed, 0.001, 10.0), t_kappa = make_draw_table(seed, 1.0, 10.0);
        const DrawTable dr{t_rho.data(), DRAW_TABLE_ENTRIES}, dk{t_kappa.data(), DRAW_TABLE_ENTRIES};
        std::vector<StoreEntry> stores(n_blocks * STORE_ENTRIES);
        // the log priors of the pending points, by the device (one small launch per round)
        std::vector<double> prior_x(n_blocks * OMEGA_MAX_REQUESTS), prior_v(n_blocks * OMEGA_MAX_REQUESTS);
        std::vector<uint32_t> prior_kind(n_blocks * OMEGA_MAX_REQUESTS);
        DeviceBuffer d_px, d_pk, d_pv;
        HIP_TRY(d_px.reserve(prior_x.size() * 8));
        HIP_TRY(d_pk.reserve(prior_x.size() * 4));
        HIP_TRY(d_pv.reserve(prior_x.size() * 8));
        const long trace_block = std::getenv("PCSF_TRACE_BLOCK") ? std::atol(std::getenv("PCSF_TRACE_BLOCK")) : -1;  // diagnostics
        struct HostPrior {
            const double* v;
            double operator()(uint32_t i, uint32_t, double) const { return v[i]; }
        };
        std::vector<size_t> at(n_blocks + 1, 0);
        auto fetch_priors = [&]() {   // for every pending point of every block, before the machines take the round's values
            size_t n = 0;
            for (size_t b = 0; b < n_blocks; ++b) {
                at[b] = n;
                for (uint32_t i = 0; i < machines[b].n_pending; ++i, ++n) {
                    prior_x[n] = machines[b].pend_x[i];
                    prior_kind[n] = machines[b].pend_kind[i];
                }
            }
            if (!n) return;
            HIP_TRY(hipMemcpyAsync(d_px.ptr, prior_x.data(), n * 8, hipMemcpyHostToDevice, eng.stream));
            HIP_TRY(hipMemcpyAsync(d_pk.ptr, prior_kind.data(), n * 4, hipMemcpyHostToDevice, eng.stream));
            HIP_TRY(launch_log_priors(d_pk.as<uint32_t>(), d_px.as<double>(), static_cast<uint32_t>(n), d_pv.as<double>(), eng.stream));
            HIP_TRY(hipMemcpyAsync(prior_v.data(), d_pv.ptr, n * 8, hipMemcpyDeviceToHost, eng.stream));
            HIP_TRY(hipStreamSynchronize(eng.stream));
        };
        run_host_rounds(eng, machines, look.max_points, false, stop, st, fetch_priors, [&](size_t b, const EvalResult* res, EvalRequest* out) {
            OmegaMachine& m = machines[b];
            if (trace_block >= 0 && static_cast<size_t>(trace_block) == b)
                for (uint32_t i = 0; i < m.n_pending; ++i) {
                    // (the evaluation's tree scale and model, as the machine formed them: rho probes move the tree scale, kappa probes the
                    // first setting -- tests replay the line through the oracle's objective)
                    const bool kappa = m.pend_kind[i] == EVAL_KAPPA;
                    fprintf(stderr, "TRACE omega stage %d kind %u x %.17g lnz %.17g prior %.17g scale %.17g qs", m.stage, m.pend_kind[i], m.pend_x[i], res[i].lnz,
                            prior_v[at[b] + i], kappa ? m.tree_scale : m.pend_x[i]);
                    for (int k = 0; k < 12; ++k) fprintf(stderr, " %.17g", (k == 0 && kappa) ? m.pend_x[i] : m.q_settings[k]);
                    fprintf(stderr, "\n");
                }
            Lookahead ahead;
            const int n = m.round(res, stores.data() + b * STORE_ENTRIES, dr, dk, look.eager, look.max_points, look.max_restarts, held_systems, HostPrior{prior_v.data() + at[b]}, out, &ahead);
            for (int i = 0; i < n; ++i)
                if (out[i].eig & TRANSIENT_EIG) out[i].eig = static_cast<uint32_t>(transient0 + b * look.max_points + (out[i].eig & ~TRANSIENT_EIG));
            return n;
        });
    }
    if (std::getenv("PCSF_TIME_BLOCKS")) {
        size_t used = 0, unused = 0, e_new = 0, e_reused = 0, e_adopted = 0;
        for (const OmegaMachine& m : machines) {
            used += m.used;
            unused += m.unused;
            e_new += m.n_eig_new;
            e_reused += m.n_eig_reused;
            e_adopted += m.n_eig_adopted;
        }
        fprintf(stderr, "run_omega: %zu evaluations consumed by the optimisers, %zu speculative ones never asked for; %zu kappa / stale-model evaluations with a new eigen-system, %zu kappa probes on a held one, %zu stage ends adopted one\n",
                used, unused, e_new, e_reused, e_adopted);
        print_chain_stats("run_omega", n_blocks, device, eng.use_tiles, st, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_enter).count(), eng.products_run());
    }
    for (size_t b = 0; b < n_blocks; ++b) {
        const OmegaMachine& m = machines[b];
        out_phylo[b] = m.failed ? NAN : decibans(m.lpr_h1, m.lpr_h0);
        if (out_anc) out_anc[b] = NAN;  // src/run.hpp:181
    }
    return PCSF_OK;
}

}  // namespace

// The AUTO rule: prune_tiles when, on average, at least 16 species have a codon in a block (blocks without a site do not count).
bool list_prefers_tiles(size_t nl, const uint8_t* codons, size_t stride, const uint64_t* off, size_t n_blocks)
{
    if (!codons || n_blocks == 0) return true;
    std::atomic<uint64_t> present{0}, counted{0};
    parallel_for(n_blocks, [&](size_t b) {
        if (off[b + 1] == off[b]) return;
        uint64_t n = 0;
        for (size_t l = 0; l < nl; ++l) {
            const uint8_t* row = codons + l * stride;
            bool any = false;
            for (uint64_t s = off[b]; s < off[b + 1] && !any; ++s) any = row[s] < 64;
            n += any;
        }
        present += n;
        counted += 1;
    });
    return counted == 0 || static_cast<double>(present) >= 16.0 * static_cast<double>(counted);
}

// one objective evaluation per block at caller-chosen parameters (no prior term).  model 3: caller-supplied
// eigen-systems (custom[n_custom], block b uses custom[n_custom == 1 ? 0 : b]) -- the parity tests feed the ORACLE's
// eigen-system through the device's P(t) and pruning kernels to separate eigen-solver noise from theirs.
int block_loglik_impl(pcsf_ctx* c, int model, const double* q_settings, const double* tree_scale, const uint8_t* codons,
                      const uint64_t* site_offsets, size_t n_blocks, size_t leaf_stride, double* out_lpr, int32_t* out_failed,
                      const EigenSystem* custom, size_t n_custom)
{
    std::vector<uint64_t> off(n_blocks + 1);
    for (size_t b = 0; b <= n_blocks; ++b) off[b] = site_offsets[b] - site_offsets[0];
    BlockEngine eng(c, codons ? codons + site_offsets[0] : nullptr, off.data(), n_blocks, leaf_stride,
                    model != 3 && call_uses_tiles(c, codons ? codons + site_offsets[0] : nullptr, leaf_stride, off.data(), n_blocks), false,
                    /*explicit_path=*/model == 3);
    eng.reserve_tables(n_blocks);
    if (model == 2) {
        eng.reserve_eig(n_blocks);  // eigen-systems and root priors come from the device solver (eigen.hip)
    } else if (model == 3) {
        eng.allow_factored = false;
        eng.reserve_eig(n_custom);
        for (size_t k = 0; k < n_custom; ++k) eng.upload_eig(k, custom[k]);
    } else {
        eng.reserve_eig(1);
        eng.upload_eig(0, c->eig[model], c->pi[model].data());
    }
    for (size_t b = 0; b < n_blocks; ++b) {
        const uint32_t eig = model == 2 ? static_cast<uint32_t>(b) : (model == 3 && n_custom > 1) ? static_cast<uint32_t>(b) : 0u;
        eng.queue(b, eig, tree_scale[b], model == 2 ? q_settings + 12 * b : nullptr);
    }
    eng.run(false);
    for (size_t b = 0; b < n_blocks; ++b) {
        const bool f = eng.failed(b);
        if (out_failed) out_failed[b] = f;
        out_lpr[b] = f ? NAN : (off[b + 1] > off[b] ? eng.value(b, eng.lnz) : 0.0);
    }
    return PCSF_OK;
}

// diagnostics/tests: eigen-systems of n kappa/omega rate matrices from the device solver.  warm_settings != null: the
// solver is first run (cold) on warm_settings[i], then on q_settings[i] starting from that eigenbasis -- the path OMEGA
// takes for every kappa probe; the results returned are those of the second run.
int debug_eigen_omega_impl(pcsf_ctx* c, const double* q_settings, const double* warm_settings, size_t n, double* right, double* left,
                           double* lambda, double* prior, int32_t* failed)
{
    const size_t slots = warm_settings ? 2 * n : n;
    DeviceBuffer d_jobs, d_r, d_l, d_lam, d_basis, d_prior, d_fail;
    HIP_TRY(d_jobs.reserve(n * sizeof(EigenJob)));
    HIP_TRY(d_r.reserve(slots * 4096 * 8));
    HIP_TRY(d_l.reserve(slots * 4096 * 8));
    HIP_TRY(d_lam.reserve(slots * 64 * 8));
    HIP_TRY(d_basis.reserve(slots * 4096 * 8));
    HIP_TRY(d_prior.reserve(slots * 64 * 8));
    HIP_TRY(d_fail.reserve(n * 4));
    HIP_TRY(hipMemsetAsync(d_fail.ptr, 0, n * 4, c->stream));
    EigenParams gp{};
    gp.jobs = d_jobs.as<EigenJob>();
    gp.n_jobs = static_cast<uint32_t>(n);
    gp.eig_right = d_r.as<double>();
    gp.eig_left = d_l.as<double>();
    gp.eig_lambda = d_lam.as<double>();
    gp.eig_basis = d_basis.as<double>();
    gp.prior_plain = d_prior.as<double>();  // state order; the lane-order copies are not wanted here
    gp.fail = d_fail.as<uint32_t>();
    DeviceBuffer d_stamps;
    const bool stamps = std::getenv("PCSF_EIGEN_STAMPS") != nullptr;  // diagnostics: where job 0's cycles go (tools/eigen_bench.py)
    if (stamps) {
        HIP_TRY(d_stamps.reserve(8 * sizeof(unsigned long long)));
        gp.stamps = d_stamps.as<unsigned long long>();
    }
    std::vector<EigenJob> jobs(n);
    for (int pass = warm_settings ? 0 : 1; pass < 2; ++pass) {
        const double* qs = pass == 0 ? warm_settings : q_settings;
        for (size_t i = 0; i < n; ++i) {
            jobs[i].eig = static_cast<uint32_t>(pass == 0 ? n + i : i);
            jobs[i].table = static_cast<uint32_t>(i);
            jobs[i].warm = (pass == 1 && warm_settings) ? static_cast<int32_t>(n + i) : -1;
            jobs[i].pad = 0;
            for (int k = 0; k < 12; ++k) jobs[i].qs[k] = qs[12 * i + k];
        }
        HIP_TRY(hipMemcpyAsync(d_jobs.ptr, jobs.data(), n * sizeof(EigenJob), hipMemcpyHostToDevice, c->stream));
        if (stamps) HIP_TRY(hipMemsetAsync(d_stamps.ptr, 0, 8 * sizeof(unsigned long long), c->stream));
        HIP_TRY(launch_eigen_omega(gp, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));  // `jobs` is rewritten for the second pass
        if (stamps) {
            unsigned long long h[8];
            HIP_TRY(hipMemcpy(h, d_stamps.ptr, sizeof h, hipMemcpyDeviceToHost));
            const double st = static_cast<double>(std::max<unsigned long long>(h[4], 1));
            fprintf(stderr, "eigen stamps (job 0, %s start; s_memtime cycles): %llu sweeps, %llu steps, kernel %llu (%llu before the sweeps); per step: phase A %.0f, phase B %.0f (wave 0 rotations %.0f, wave 1 V update %.0f)\n",
                    pass == 0 ? "cold" : (warm_settings ? "warm" : "cold"), h[5], h[4], h[6], h[7], h[0] / st, h[3] / st, h[1] / st, h[2] / st);
        }
    }
    std::vector<uint32_t> f(n);
    HIP_TRY(hipMemcpyAsync(right, d_r.ptr, n * 4096 * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(left, d_l.ptr, n * 4096 * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(lambda, d_lam.ptr, n * 64 * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(prior, d_prior.ptr, n * 64 * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(f.data(), d_fail.ptr, n * 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (size_t i = 0; i < n; ++i) failed[i] = static_cast<int32_t>(f[i]);
    return PCSF_OK;
}

int score_blocks_impl(pcsf_ctx* c, int algo, const uint8_t* codons, const uint64_t* site_offsets, size_t n_blocks,
                      size_t leaf_stride, int compute_anc, uint32_t seed, float* out_phylo, float* out_anc)
{
    if (algo != PCSF_FIXED && algo != PCSF_MLE && algo != PCSF_OMEGA) return c->fail(PCSF_ERR_INVALID_ARG, "unknown algorithm");
    struct CallTimer {   // PCSF_TIME_BLOCKS: the whole call, engines' set-up and release included
        const bool on = std::getenv("PCSF_TIME_BLOCKS") != nullptr;
        const std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
        ~CallTimer() { if (on) fprintf(stderr, "score_blocks: whole call %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()); }
    } call_timer;
    // Blocks are independent, so a long list is processed in chunks that bound the engines' footprint in HBM: per block and evaluation
    // slot one row of edge scales, per OMEGA block seventeen eigen-systems, per search a value store (and, only under
    // PCSF_FACTORED_MAX_TILES -- round 2's explicit path --, 9.6 MB of branch-matrix tables per evaluation of a 100-leaf tree).
    // PCSF_BLOCK_GROUPS=g advances g groups of blocks concurrently, one host thread and its own streams each: one group's kernels
    // fill the tails of the other's launches and its host work runs beside them (8,256 blocks: MLE 0.58 -> 0.31 s, OMEGA 2.85 -> 2.18 s
    // with the host engine; four groups added nothing).  Short lists lose: every group needs as many rounds as the whole list (the block
    // with the longest optimiser path sets it).  Blocks are independent: grouping never changes a result (tests/test_gpu_blocks.py).
    size_t chunk = n_blocks;
    size_t n_groups = 1;
    // one pruning kernel for the whole call, decided before the list is cut into chunks and groups (pinned by the caller: pcsf_set_block_kernel)
    const bool tiles = algo != PCSF_FIXED && call_uses_tiles(c, codons, leaf_stride, site_offsets, n_blocks);
    const bool eager = n_blocks < eager_restarts_below();
    if (algo != PCSF_FIXED && std::getenv("PCSF_TIME_BLOCKS")) fprintf(stderr, "pruning kernel of this list: %s\n", tiles ? "prune_tiles" : "prune_factored");
    if (algo != PCSF_FIXED) {
        const LookaheadPolicy look = lookahead_policy(eager);
        const size_t slots = static_cast<size_t>(algo == PCSF_MLE ? 2 * look.max_points : look.max_points);
        const size_t per_block = slots * static_cast<size_t>(c->n_nodes - 1) * 64 * sizeof(double) +
                                 (algo == PCSF_OMEGA ? (N_ROLES + static_cast<size_t>(look.max_points)) * size_t((tiles ? 3 : 5) * 4096 + 5 * 64) * sizeof(double) : 0) +   // R, L, basis (+ prune_factored's two fragment-ordered copies)
                                 (algo == PCSF_MLE ? 2 : 1) * STORE_ENTRIES * sizeof(StoreEntry);
        chunk = std::max<size_t>(64, (size_t(48) << 30) / per_block);
        if (std::getenv("PCSF_FACTORED_MAX_TILES")) {
            const size_t ne = c->schedule.slot_node.size();
            const size_t per_table = (ne * 4096 + static_cast<size_t>(c->n_leaves) * LEAF_TABLE_DOUBLES_DEV) * sizeof(double);
            chunk = std::min(chunk, std::max<size_t>(64, (size_t(16) << 30) / std::max<size_t>(per_table * slots, 1)));
        }
        // two groups once the list is long enough -- measured with the host engine: MLE from 512 blocks, OMEGA from 1,536 (profiles/r02/blocks_groups.txt)
        if (const char* v = std::getenv("PCSF_BLOCK_GROUPS")) n_groups = static_cast<size_t>(std::max(1, std::atoi(v)));
        else n_groups = n_blocks >= (algo == PCSF_MLE ? 512u : 1536u) ? 2 : 1;
        n_groups = std::min(n_groups, std::max<size_t>(n_blocks, 1));
        chunk = std::min(chunk / n_groups + 1, (n_blocks + n_groups - 1) / n_groups);
        chunk = std::max<size_t>(chunk, 1);
    }
    struct Range {
        size_t b0, nb;
    };
    std::vector<Range> ranges;
    for (size_t b0 = 0; b0 < n_blocks; b0 += chunk) ranges.push_back({b0, std::min(chunk, n_blocks - b0)});
    std::atomic<bool> stop_groups{false};   // set by the first group that fails: the others leave at their next round
    auto run_range = [&](const Range& r, bool own_streams, StartGate* gate) -> int {
        std::vector<uint64_t> off(r.nb + 1);  // CSR offsets re-based at the range's first site
        for (size_t b = 0; b <= r.nb; ++b) off[b] = site_offsets[r.b0 + b] - site_offsets[r.b0];
        const uint8_t* cod = codons ? codons + site_offsets[r.b0] : nullptr;
        float* ph = out_phylo + r.b0;
        float* an = out_anc ? out_anc + r.b0 : nullptr;
        switch (algo) {
            case PCSF_FIXED: return run_fixed(c, cod, off.data(), r.nb, leaf_stride, compute_anc, ph, an);
            case PCSF_MLE: return run_mle(c, cod, off.data(), r.nb, leaf_stride, compute_anc, seed, ph, an, tiles, own_streams, eager, &stop_groups, gate);
            default: return run_omega(c, cod, off.data(), r.nb, leaf_stride, seed, ph, an, tiles, own_streams, eager, &stop_groups, gate);
        }
    };
    if (n_groups <= 1 || ranges.size() <= 1) {
        for (const Range& r : ranges) {
            const int rc = run_range(r, false, nullptr);
            if (rc != PCSF_OK) return rc;
        }
        return PCSF_OK;
    }
    // ranges are dealt to n_groups threads; the first failure (an exception message) is reported after all have stopped
    std::atomic<size_t> next{0};
    std::mutex err_lock;
    std::string err;
    int err_rc = PCSF_OK;
    std::vector<std::thread> th;
    StartGate gate;   // the groups' first ranges allocate, then start their rounds together
    gate.expected = std::min(n_groups, ranges.size());
    for (size_t g = 0; g < std::min(n_groups, ranges.size()); ++g)
        th.emplace_back([&]() {
            const hipError_t dev_rc = hipSetDevice(c->device);  // the current device is per host thread
            bool first = true;
            GateTicket unused_ticket(&gate);   // a thread that never reaches a first range still counts
            for (size_t i = next++; i < ranges.size(); i = next++) {
                int rc = PCSF_OK;
                std::string msg;
                try {
                    if (dev_rc != hipSuccess) throw std::runtime_error(std::string("hipSetDevice: ") + hipGetErrorString(dev_rc));
                    StartGate* g_first = first ? &gate : nullptr;
                    if (first) unused_ticket.used = true;
                    first = false;
                    rc = run_range(ranges[i], true, g_first);
                } catch (const std::exception& ex) {
                    rc = PCSF_ERR_HIP;
                    msg = ex.what();
                } catch (...) {   // nothing may leave a worker thread: std::terminate inside a library call
                    rc = PCSF_ERR_HIP;
                    msg = "unknown exception in a block-group worker";
                }
                if (rc != PCSF_OK) {
                    std::lock_guard<std::mutex> g2(err_lock);
                    if (err_rc == PCSF_OK) {
                        err_rc = rc;
                        err = msg;
                    }
                    next = ranges.size();
                    stop_groups.store(true, std::memory_order_relaxed);
                    return;
                }
            }
        });
    for (auto& t : th) t.join();
    if (err_rc != PCSF_OK) {
        if (!err.empty()) throw std::runtime_error(err);
        return err_rc;
    }
    return PCSF_OK;
}

}  // namespace pcsf
