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
#include <cstring>
#include <memory>
#include <mutex>
#include <random>
#include <stdexcept>
#include <thread>
#include <vector>

#include "ctx.hpp"
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

// Which pruning kernel a LIST of blocks gets (see BlockEngine::use_tiles): prune_tiles when, on average, at least 16 species have a
// codon in a block.  PCSF_BLOCK_KERNEL=tiles|factored overrides; PCSF_FACTORED_MAX_TILES (round 2's path rule) implies factored.
static bool list_prefers_tiles(const pcsf_ctx* c, const uint8_t* codons, size_t stride, const uint64_t* off, size_t n_blocks)
{
    if (std::getenv("PCSF_FACTORED_MAX_TILES")) return false;
    if (const char* v = std::getenv("PCSF_BLOCK_KERNEL")) return std::string(v) != "factored";
    if (!codons || n_blocks == 0) return true;
    const size_t nl = static_cast<size_t>(c->n_leaves);
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

struct BlockEngine {
    pcsf_ctx* c;
    size_t n_blocks, leaf_stride, n_sites_total;
    const uint64_t* off;
    hipStream_t stream;
    int ne, nl, n;
    DeviceBuffer d_codons;                 // [nl][leaf_stride] compact copy
    DeviceBuffer d_frag, d_leaf;   // table sets, one per evaluation of a run
    DeviceBuffer d_eig_r, d_eig_l, d_eig_lam, d_eig_basis, d_prior, d_ancw, d_eig_w;  // eigen-systems, their root priors, L = R^T diag(w)
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
        if (n_sites_total > 0) {
            HIP_TRY(d_codons.reserve(static_cast<size_t>(nl) * n_sites_total));
            HIP_TRY(upload_rows(d_codons.ptr, codons, stride, n_sites_total, nl, stream));
        }
        mark("codon upload");
        if (const char* v = std::getenv("PCSF_FACTORED_MAX_TILES")) fixed_max_tiles = std::max(0, std::atoi(v));
        if (fixed_max_tiles >= 0) use_tiles = false;   // the tile-count rule is a statement about round 2's two paths
        if (const char* v = std::getenv("PCSF_TILE_JOBS")) tile_jobs_per_group = static_cast<size_t>(std::max(1, std::atoi(v)));
        if (const char* v = std::getenv("PCSF_EIGEN_HOST")) host_eigen = std::atoi(v) != 0;
        if (const char* fs = std::getenv("PCSF_FACTORED_STAMPS")) {
            fact_stamp_classes = std::atoi(fs) >= 2;
            HIP_TRY(d_fact_stamps.reserve(16 * sizeof(unsigned long long)));
            HIP_TRY(hipMemsetAsync(d_fact_stamps.ptr, 0, 16 * sizeof(unsigned long long), stream));
        }
        block_factored.assign(nb, use_tiles ? 1 : 0);
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
                if (!use_tiles) block_factored[b] = block_prefers_factored(c, codons, stride, offsets[b], offsets[b + 1], fixed_max_tiles);
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
    // eigen-systems already resident (uploaded, or solved by an earlier run()) survive a growth
    void reserve_eig(size_t count)
    {
        if (count <= cap_eig) return;
        HIP_TRY(d_eig_r.reserve_keep(count * 4096 * sizeof(double), cap_eig * 4096 * sizeof(double), stream));
        HIP_TRY(d_eig_l.reserve_keep(count * 4096 * sizeof(double), cap_eig * 4096 * sizeof(double), stream));
        HIP_TRY(d_eig_lam.reserve_keep(count * 64 * sizeof(double), cap_eig * 64 * sizeof(double), stream));
        HIP_TRY(d_eig_basis.reserve_keep(count * 4096 * sizeof(double), cap_eig * 4096 * sizeof(double), stream));
        HIP_TRY(d_prior.reserve_keep(count * 64 * sizeof(double), cap_eig * 64 * sizeof(double), stream));
        HIP_TRY(d_ancw.reserve_keep(count * 64 * sizeof(double), cap_eig * 64 * sizeof(double), stream));
        HIP_TRY(d_eig_w.reserve_keep(count * 64 * sizeof(double), cap_eig * 64 * sizeof(double), stream));
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
            if (nfe)
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
                fp.eig_w = d_eig_w.as<double>();
                fp.prior = d_prior.as<double>();
                fp.anc_weight = d_ancw.as<double>();
                fp.edge_scale = d_scale.as<double>();
                fp.out_lnz = p_lnz;
                fp.out_anc = want_anc ? p_anc : nullptr;
                fp.deep = c->schedule.stack_depth > MAX_STACK_DEPTH ? 1u : 0u;
                fp.stamps = d_fact_stamps.ptr ? d_fact_stamps.as<unsigned long long>() : nullptr;
                fp.stamp_classes = fact_stamp_classes ? 1u : 0u;
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
                gp.eig_lambda = d_eig_lam.as<double>();
                gp.eig_basis = d_eig_basis.as<double>();
                gp.prior = d_prior.as<double>();
                gp.anc_weight = d_ancw.as<double>();
                gp.eig_w = d_eig_w.as<double>();
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

// values an optimiser stage has seen: the objective is a pure function of the point, so a point asked for
// again (the reference re-evaluates lo, hi and the chosen start when it sets up Brent) is served from here
struct EvalStore {
    struct Entry {
        double x, lpr, anc;
        bool failed;
        mutable bool used = false;   // diagnostics: the optimiser consumed this value (speculative evaluations may never be)
    };
    std::vector<Entry> entries;
    size_t n_used = 0, n_unused = 0;   // over the stages cleared so far
    const Entry* find(double x) const
    {
        for (const Entry& e : entries)
            if (e.x == x) return &e;
        return nullptr;
    }
    void tally()
    {
        for (const Entry& e : entries) (e.used ? n_used : n_unused) += 1;
    }
    void clear()
    {
        tally();
        entries.clear();
    }
};

constexpr int MAX_POINTS_PER_ROUND = 12;   // per block: certain points first, then speculative restarts
constexpr int MAX_SPECULATIVE_RESTARTS = 8;
// Random restarts are evaluated ahead of knowing that a search needs any only while fewer blocks than this are active: with
// thousands of blocks a round is bound by its work, not its latency, and a third of OMEGA's evaluations were such restarts
// that no optimiser ever asked for (PCSF_EAGER_RESTARTS_BELOW overrides)
inline size_t eager_restarts_below()
{
    static const size_t v = [] {
        const char* e = std::getenv("PCSF_EAGER_RESTARTS_BELOW");
        return e ? static_cast<size_t>(std::max(0L, std::atol(e))) : size_t(1024);
    }();
    return v;
}

struct Lookahead {  // MaxLik1D::upcoming() of one block for the current round
    int n = 0;
    double x[MAX_POINTS_PER_ROUND];
    bool seen_before(int rank) const  // the same point earlier in the list (lo, hi, chosen recur when Brent is set up)
    {
        for (int i = 0; i < rank; ++i)
            if (x[i] == x[rank]) return true;
        return false;
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

// ---- MLE ---------------------------------------------------------------------------------------
int run_mle(pcsf_ctx* c, const uint8_t* codons, const uint64_t* off, size_t n_blocks, size_t leaf_stride, int compute_anc,
            uint32_t seed, float* out_phylo, float* out_anc, bool tiles, bool own_streams, const std::atomic<bool>* stop)
{
    const auto t_setup = std::chrono::steady_clock::now();
    BlockEngine eng(c, codons, off, n_blocks, leaf_stride, tiles, own_streams);
    eng.reserve_tables(std::max<size_t>(64, 2 * n_blocks));
    eng.reserve_eig(2);
    eng.upload_eig(0, c->eig[0], c->pi[0].data());
    eng.upload_eig(1, c->eig[1], c->pi[1].data());
    std::vector<std::mt19937> gen(n_blocks);
    std::vector<MaxLik1D> opt[2] = {std::vector<MaxLik1D>(n_blocks), std::vector<MaxLik1D>(n_blocks)};
    std::vector<EvalStore> store[2] = {std::vector<EvalStore>(n_blocks), std::vector<EvalStore>(n_blocks)};
    std::vector<double> res_lpr(2 * n_blocks, 0.0), res_anc(2 * n_blocks, 0.0);
    std::vector<char> failed(n_blocks, 0), started1(n_blocks, 0);
    for (size_t b = 0; b < n_blocks; ++b) gen[b].seed(seed);
    struct Pending {
        size_t block, eval;
        double x;
        int model;
    };
    // The reference optimises the coding model, then the non-coding one (src/run.hpp:193-194), drawing the random restarts of both
    // from one generator per block.  The two searches are independent otherwise, and a search draws only BEFORE its Brent iteration
    // (MaxLik1D::draws_finished): the non-coding search of a block starts as soon as the coding one has drawn its last number, sees
    // the same numbers as it would afterwards, and the two advance in the same rounds -- half the rounds, the same results.
    for (size_t b = 0; b < n_blocks; ++b) opt[0][b].start(1.0, 1e-2, 10.0, &gen[b]);
    const bool timing = std::getenv("PCSF_TIME_BLOCKS") != nullptr;  // diagnostics: where the rounds spend their time
    const auto t_enter = std::chrono::steady_clock::now();
    double t_run = 0.0;
    size_t rounds = 0, n_evals = 0;
    for (;;) {
        if (stop && stop->load(std::memory_order_relaxed)) return PCSF_ERR_HIP;   // another group of this call has failed: its error is the one reported
        // feed every optimiser from its store; blocks without a codon have lpr = 0 at every scale
        size_t active = 0;
        for (size_t b = 0; b < n_blocks; ++b) {
            for (int model = 0; model < 2 && !failed[b]; ++model) {
                if (model == 1) {
                    if (!started1[b]) {
                        if (!opt[0][b].draws_finished()) break;
                        opt[1][b].start(1.0, 1e-2, 10.0, &gen[b]);
                        started1[b] = 1;
                    }
                }
                MaxLik1D& o = opt[model][b];
                while (!o.done()) {
                    if (off[b + 1] == off[b]) {
                        o.supply(0.0, 0.0);
                        continue;
                    }
                    const EvalStore::Entry* e = store[model][b].find(o.request());
                    if (!e) break;
                    if (e->failed) {  // std::runtime_error in PhyloModel_make -> NaN row
                        failed[b] = 1;
                        break;
                    }
                    e->used = true;
                    o.supply(e->lpr, e->anc);
                }
            }
            if (!failed[b] && (!opt[0][b].done() || !started1[b] || !opt[1][b].done())) ++active;
        }
        if (!active) break;
        // queue the points the active optimisers will or may need: rank 0 (certain) for everybody first
        std::vector<Pending> pending;
        std::vector<Lookahead> ahead[2] = {std::vector<Lookahead>(n_blocks), std::vector<Lookahead>(n_blocks)};
        for (size_t b = 0; b < n_blocks; ++b)
            for (int model = 0; model < 2; ++model)
                if (!failed[b] && (model == 0 || started1[b]) && !opt[model][b].done())
                    ahead[model][b].n = opt[model][b].upcoming(ahead[model][b].x, MAX_POINTS_PER_ROUND, MAX_SPECULATIVE_RESTARTS, active < eager_restarts_below());
        for (int rank = 0; rank < MAX_POINTS_PER_ROUND && eng.free_evaluations() > 0; ++rank) {
            bool any = false;
            for (int model = 0; model < 2; ++model)
                for (size_t b = 0; b < n_blocks && eng.free_evaluations() > 0; ++b) {
                    if (rank >= ahead[model][b].n) continue;
                    any = true;
                    const double x = ahead[model][b].x[rank];
                    if (store[model][b].find(x) || ahead[model][b].seen_before(rank)) continue;
                    pending.push_back({b, eng.queue(b, static_cast<uint32_t>(model), x, nullptr), x, model});
                }
            if (!any) break;
        }
        const auto t_a = std::chrono::steady_clock::now();
        eng.run(compute_anc != 0);
        t_run += std::chrono::duration<double>(std::chrono::steady_clock::now() - t_a).count();
        ++rounds;
        n_evals += pending.size();
        for (const Pending& q : pending)
            store[q.model][q.block].entries.push_back({q.x, eng.value(q.eval, eng.lnz), compute_anc ? eng.value(q.eval, eng.anc) : 0.0, eng.failed(q.eval)});
    }
    if (timing) {
        size_t used = 0, unused = 0;
        for (int model = 0; model < 2; ++model)
            for (size_t b = 0; b < n_blocks; ++b) {
                store[model][b].tally();
                used += store[model][b].n_used;
                unused += store[model][b].n_unused;
            }
        fprintf(stderr, "run_mle: %zu evaluations consumed by the optimisers, %zu speculative ones never asked for\n", used, unused);
        const double all = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_enter).count();
        fprintf(stderr, "run_mle: %zu blocks (%zu on the factored path), %zu rounds, %zu evaluations: host %.3f s, device rounds %.3f s; engine set-up %.3f s\n", n_blocks,
                static_cast<size_t>(std::count(eng.block_factored.begin(), eng.block_factored.end(), 1)), rounds, n_evals, all - t_run, t_run,
                std::chrono::duration<double>(t_enter - t_setup).count());
    }
    for (int model = 0; model < 2; ++model)
        for (size_t b = 0; b < n_blocks; ++b) {
            res_lpr[model * n_blocks + b] = opt[model][b].last_lpr();
            res_anc[model * n_blocks + b] = opt[model][b].last_anc();
        }
    for (size_t b = 0; b < n_blocks; ++b) {
        if (failed[b]) {
            out_phylo[b] = NAN;
            if (out_anc) out_anc[b] = NAN;
            continue;
        }
        out_phylo[b] = decibans(res_lpr[b], res_lpr[n_blocks + b]);
        if (out_anc) out_anc[b] = decibans(res_anc[b], res_anc[n_blocks + b]);
    }
    return PCSF_OK;
}

// ---- OMEGA -------------------------------------------------------------------------------------
// Eigen-systems a block keeps between rounds, by role.  OWN: the model the block is at.  LO / HI: kappa = 1 and kappa = 10, the ends
// every kappa stage evaluates first -- the three kappa stages of a hypothesis ask for the SAME two rate matrices.  P0 / P1: the last two
// interior kappa probes (Brent's steps shrink geometrically: the nearest earlier probe is a far better start for the Jacobi iteration
// than the stage's first point).  A probe whose settings equal a held system is not solved again (a fifth of all solves on the
// configs[2] list), every other probe starts from the nearest held system of its family (same omega, sigma, frequencies).  Roles map
// to physical slots through `slot`, so that the system of a stage's last probe BECOMES the block's own by a swap when the stage ends.
enum EigRole { ROLE_OWN = 0, ROLE_LO = 1, ROLE_HI = 2, ROLE_P0 = 3, ROLE_P1 = 4, N_ROLES = 5 };
struct OmegaBlock {
    double q_settings[12];      // kappa, omega, sigma, nine ratios: the model the block is AT (src/run.hpp:75-79)
    double eig_settings[12];    // settings the block's own eigen-system slot currently holds
    bool eig_valid = false;
    uint32_t slot[N_ROLES];     // physical eigen-system slot of every role
    double held[N_ROLES][12];   // settings the role's slot holds (OWN: eig_settings / eig_valid above)
    bool held_valid[N_ROLES] = {false, false, false, false, false};
    int next_p = 0;             // which of P0 / P1 the next interior probe overwrites
    double tree_scale = 1.0;
    std::mt19937 gen;
    MaxLik1D opt;
    EvalStore store;
    int stage = 0;  // 0..5: H0 (rho,kappa)x3 ; 6..11: H1 (rho,kappa)x3 ; 12 = finished
    double lpr_h0 = 0.0, lpr_h1 = 0.0;
    bool failed = false, done = false;
    bool needs_validation = false;  // the model was re-instantiated (PhyloModel_make can throw) and must be checked
};

int run_omega(pcsf_ctx* c, const uint8_t* codons, const uint64_t* off, size_t n_blocks, size_t leaf_stride, uint32_t seed,
              float* out_phylo, float* out_anc, bool tiles, bool own_streams, const std::atomic<bool>* stop)
{
    const auto t_enter = std::chrono::steady_clock::now();
    BlockEngine eng(c, codons, off, n_blocks, leaf_stride, tiles, own_streams);
    const size_t cap = std::max<size_t>(64, 2 * n_blocks);
    eng.reserve_tables(cap);
    // slots [role * n_blocks + b]: the eigen-systems block b holds (EigRole; roles are swapped between them, see OmegaBlock);
    // slots N_ROLES * n_blocks + k: speculative kappa probes of a round (random restarts nobody may ask for)
    const bool held_systems = !(std::getenv("PCSF_OMEGA_HELD_SYSTEMS") && std::atoi(std::getenv("PCSF_OMEGA_HELD_SYSTEMS")) == 0);
    const size_t transient0 = N_ROLES * n_blocks;
    eng.reserve_eig(transient0 + cap);
    const double t_setup = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_enter).count();
    std::vector<std::unique_ptr<OmegaBlock>> blk(n_blocks);
    const long trace_block = std::getenv("PCSF_TRACE_BLOCK") ? std::atol(std::getenv("PCSF_TRACE_BLOCK")) : -1;  // diagnostics
    const size_t nl = static_cast<size_t>(c->n_leaves);

    // Validation-only pass (PhyloModel_make can throw): rebuild the listed blocks' eigen-systems from their
    // q_settings, build their matrices at their tree scale and read the flags.
    auto validate_all = [&](const std::vector<size_t>& which) {
        for (size_t at = 0; at < which.size();) {
            std::vector<std::pair<size_t, size_t>> queued;
            for (; at < which.size() && eng.free_evaluations() > 0; ++at) {
                OmegaBlock& ob = *blk[which[at]];
                queued.push_back({which[at], eng.queue(which[at], ob.slot[ROLE_OWN], ob.tree_scale, ob.q_settings, false,
                                                       ob.eig_valid ? static_cast<int32_t>(ob.slot[ROLE_OWN]) : -1)});
                std::memcpy(ob.eig_settings, ob.q_settings, sizeof ob.q_settings);
                ob.eig_valid = true;
            }
            eng.run(false);
            for (const auto& q : queued)
                if (eng.failed(q.second)) blk[q.first]->failed = true;
        }
    };

    std::vector<size_t> live;
    for (size_t b = 0; b < n_blocks; ++b) {
        blk[b].reset(new OmegaBlock);
        OmegaBlock& ob = *blk[b];
        ob.gen.seed(seed);
        for (int r = 0; r < N_ROLES; ++r) ob.slot[r] = static_cast<uint32_t>(r * n_blocks + b);
        // POLICY ("OMEGA NaN rows"): the reference yields NaN for every block without a
        // complete codon (GSL's non-symmetric eigen-solver degrades on the uniform F3x4 matrix)
        if (off[b + 1] == off[b]) {
            ob.failed = true;
            continue;
        }
        live.push_back(b);
        ob.q_settings[0] = 2.5;  // kappa, omega, sigma, nine ratios (src/run.hpp:75-79)
        ob.q_settings[1] = 1.0;
        ob.q_settings[2] = 1.0;
        for (int i = 3; i < 12; ++i) ob.q_settings[i] = 1.0;
    }
    const auto tp0 = std::chrono::steady_clock::now();
    validate_all(live);
    const auto tp1 = std::chrono::steady_clock::now();
    // F3x4 nucleotide counts over every non-gap codon of every row (src/run.hpp:105-134)
    parallel_for(live.size(), [&](size_t i) {
        OmegaBlock& ob = *blk[live[i]];
        if (ob.failed) return;
        double cnt[3][4];
        for (auto& r : cnt)
            for (double& v : r) v = 1.0;
        const size_t b = live[i];
        for (size_t l = 0; l < nl; ++l)
            for (uint64_t s = off[b]; s < off[b + 1]; ++s) {
                const uint8_t cd = codons[l * leaf_stride + s];
                if (cd < 64) {
                    cnt[0][cd >> 4] += 1;
                    cnt[1][(cd >> 2) & 3] += 1;
                    cnt[2][cd & 3] += 1;
                }
            }
        for (int p = 0; p < 3; ++p)
            for (int j = 0; j < 3; ++j) ob.q_settings[3 + 3 * p + j] = cnt[p][j] / cnt[p][3];
    });
    const auto tp2 = std::chrono::steady_clock::now();
    {
        std::vector<size_t> ok;
        for (size_t b : live)
            if (!blk[b]->failed) ok.push_back(b);
        validate_all(ok);
    }
    if (std::getenv("PCSF_TIME_BLOCKS"))
        fprintf(stderr, "run_omega prelude: first validation %.3f s, F3x4 counts %.3f s, second validation %.3f s\n", std::chrono::duration<double>(tp1 - tp0).count(),
                std::chrono::duration<double>(tp2 - tp1).count(), std::chrono::duration<double>(std::chrono::steady_clock::now() - tp2).count());

    auto start_stage = [&](OmegaBlock& ob) {
        ob.store.clear();
        if ((ob.stage % 2) == 0) ob.opt.start(ob.tree_scale, 0.001, 10.0, &ob.gen);
        else ob.opt.start(ob.q_settings[0], 1.0, 10.0, &ob.gen);
    };
    for (size_t b : live)
        if (!blk[b]->failed) start_stage(*blk[b]);

    const bool timing = std::getenv("PCSF_TIME_BLOCKS") != nullptr;  // diagnostics: where the rounds spend their time
    const double t_prelude = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_enter).count() - t_setup;
    double t_host = 0.0, t_run = 0.0;
    size_t rounds = 0, evals = 0, eig_evals = 0, eig_reused = 0, eig_adopted = 0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    struct Pending {
        size_t block, eval;
        double x;
    };
    for (;;) {
        if (stop && stop->load(std::memory_order_relaxed)) return PCSF_ERR_HIP;   // another group of this call has failed: its error is the one reported
        const auto t0 = now();
        // (1) feed every optimiser from its store; finished stages move the block on
        size_t active = 0;
        for (size_t b : live) {
            OmegaBlock& ob = *blk[b];
            while (!ob.failed && !ob.done && !ob.needs_validation) {
                if (ob.opt.done()) {
                    // max_lik_lpr_leaves returned: the model stays at the LAST evaluated point
                    if ((ob.stage % 2) == 0) {
                        ob.tree_scale = ob.opt.last_x();
                    } else {
                        ob.q_settings[0] = ob.opt.last_x();
                        // the model stays at the stage's LAST probe -- whose eigen-system, if it was kept, becomes the block's own by a
                        // swap of roles instead of being solved once more at the next rho evaluation
                        for (int role = ROLE_LO; role < N_ROLES && held_systems; ++role)
                            if (ob.held_valid[role] && std::memcmp(ob.held[role], ob.q_settings, sizeof ob.q_settings) == 0) {
                                std::swap(ob.slot[ROLE_OWN], ob.slot[role]);
                                double tmp[12];
                                std::memcpy(tmp, ob.eig_settings, sizeof tmp);
                                std::memcpy(ob.eig_settings, ob.held[role], sizeof tmp);
                                std::memcpy(ob.held[role], tmp, sizeof tmp);
                                std::swap(ob.eig_valid, ob.held_valid[role]);
                                ++eig_adopted;
                                break;
                            }
                    }
                    (ob.stage < 6 ? ob.lpr_h0 : ob.lpr_h1) = ob.opt.last_lpr();
                    ++ob.stage;
                    if (ob.stage == 12) {
                        ob.done = true;
                    } else if (ob.stage == 6) {  // omega_H1 = 0.2, sigma_H1 = 0.01, keep rho and kappa (src/run.hpp:161-166)
                        ob.q_settings[1] = 0.2;
                        ob.q_settings[2] = 0.01;
                        ob.needs_validation = true;   // validated in the round that follows
                    } else {
                        start_stage(ob);
                    }
                    continue;
                }
                const EvalStore::Entry* e = ob.store.find(ob.opt.request());
                if (!e) break;
                if (e->failed) {
                    ob.failed = true;
                    break;
                }
                if (trace_block >= 0 && static_cast<size_t>(trace_block) == b)
                    fprintf(stderr, "TRACE omega stage %d x %.17g lpr %.17g\n", ob.stage, e->x, e->lpr);
                e->used = true;
                ob.opt.supply(e->lpr, 0.0);
            }
            if (!ob.failed && !ob.done) ++active;
        }
        if (!active) break;
        // Blocks that switch to H1 need their new model validated (one evaluation whose flags are read) before their next
        // stage starts.  That evaluation rides along with this round's points of the other blocks -- as a round of its own it
        // cost a full eigen-solver launch and a synchronisation for a handful of blocks, some forty times per run (0.05 of
        // 0.36 s on the 516-block file).  Blocks whose validation does not fit this round wait for the next.
        std::vector<std::pair<size_t, size_t>> validating;  // (block, evaluation)
        for (size_t b = 0; b < n_blocks; ++b) {
            OmegaBlock& ob = *blk[b];
            if (ob.failed || ob.done || !ob.needs_validation || eng.free_evaluations() == 0) continue;
            validating.push_back({b, eng.queue(b, ob.slot[ROLE_OWN], ob.tree_scale, ob.q_settings, false, ob.eig_valid ? static_cast<int32_t>(ob.slot[ROLE_OWN]) : -1)});
            std::memcpy(ob.eig_settings, ob.q_settings, sizeof ob.q_settings);
            ob.eig_valid = true;
        }
        // (2) queue the points the optimisers will or may need next: rank 0 (certain) for every block first
        std::vector<Pending> pending;
        std::vector<Lookahead> ahead(n_blocks);
        std::vector<char> rebuilt(n_blocks, 0);  // block's own eigen-system slot is rebuilt in this run
        std::vector<unsigned char> written(n_blocks, 0);   // bit role: the role's slot is rewritten in this run
        std::vector<int> ahead_certain(n_blocks, 0);
        for (size_t b : live) {
            OmegaBlock& ob = *blk[b];
            if (!ob.failed && !ob.done && !ob.needs_validation)
                ahead[b].n = ob.opt.upcoming(ahead[b].x, MAX_POINTS_PER_ROUND, MAX_SPECULATIVE_RESTARTS, active < eager_restarts_below(), &ahead_certain[b]);
        }
        for (int rank = 0; rank < MAX_POINTS_PER_ROUND && eng.free_evaluations() > 0; ++rank) {
            bool any = false;
            for (size_t b : live) {
                if (eng.free_evaluations() == 0) break;
                if (rank >= ahead[b].n) continue;
                any = true;
                OmegaBlock& ob = *blk[b];
                const double x = ahead[b].x[rank];
                if (ob.store.find(x) || ahead[b].seen_before(rank)) continue;
                size_t k;
                if ((ob.stage % 2) == 0) {  // rho: the block's own eigen-system, brought up to date if need be
                    const bool stale = !ob.eig_valid || std::memcmp(ob.eig_settings, ob.q_settings, sizeof ob.q_settings) != 0;
                    if (stale && !rebuilt[b]) {
                        k = eng.queue(b, ob.slot[ROLE_OWN], x, ob.q_settings, false, ob.eig_valid ? static_cast<int32_t>(ob.slot[ROLE_OWN]) : -1);
                        std::memcpy(ob.eig_settings, ob.q_settings, sizeof ob.q_settings);
                        ob.eig_valid = true;
                        rebuilt[b] = 1;
                        ++eig_evals;
                    } else {
                        k = eng.queue(b, ob.slot[ROLE_OWN], x, nullptr, rebuilt[b] != 0);
                    }
                } else {  // kappa: a probe model
                    double qs[12];
                    std::memcpy(qs, ob.q_settings, sizeof qs);
                    qs[0] = x;
                    // Which probes may change what the block holds: the ends of the range, and the probes of the Brent phase (one per
                    // round, each asked for only after its predecessor was consumed).  Random restarts never do, and always start from
                    // the block's own system: whether a restart is evaluated ahead of need, and in which round, depends on how many
                    // blocks share the engine -- no result may (tests/test_gpu_blocks.py, block groups).
                    const bool certain = rank < ahead_certain[b];
                    const bool brent_phase = ob.opt.draws_finished();
                    auto holds = [&](int role) {   // the role's slot holds exactly this model, and is not being rewritten in this run
                        const bool valid = role == ROLE_OWN ? (ob.eig_valid && !rebuilt[b]) : (ob.held_valid[role] && !(written[b] & (1u << role)));
                        return valid && std::memcmp(role == ROLE_OWN ? ob.eig_settings : ob.held[role], qs, sizeof qs) == 0;
                    };
                    int hit = -1;
                    for (int role = 0; role < N_ROLES && held_systems && hit < 0; ++role)
                        if (holds(role)) hit = role;
                    if (hit >= 0) {   // solved before (the block's own model, or an end of the kappa range in an earlier stage): no new eigen-system
                        k = eng.queue(b, ob.slot[hit], ob.tree_scale, nullptr, false);
                        ++eig_reused;
                    } else {
                        // where the new system goes: the ends of the range and the certain interior probes are kept, restarts are not
                        int target = -1;
                        if (held_systems && certain) {
                            if (x == 1.0) target = ROLE_LO;
                            else if (x == 10.0) target = ROLE_HI;
                            else if (brent_phase) {
                                target = ROLE_P0 + ob.next_p;
                                ob.next_p ^= 1;
                            }
                        }
                        // Jacobi starts in the eigenbasis of the nearest held system of the same family (all settings but kappa equal).
                        // The ends only look at the block's own system and at what their slot held before: what else is complete when
                        // they are solved depends on how the round was packed, and no result may depend on that.
                        int32_t warm = -1;
                        double best = 0.0;
                        for (int role = 0; role < N_ROLES; ++role) {
                            if ((!held_systems || target < 0) && role != ROLE_OWN) break;
                            if ((target == ROLE_LO || target == ROLE_HI) && role != ROLE_OWN && role != target) continue;
                            const double* hs = role == ROLE_OWN ? ob.eig_settings : ob.held[role];
                            const bool valid = role == ROLE_OWN ? (ob.eig_valid && !rebuilt[b]) : (ob.held_valid[role] && !(written[b] & (1u << role)));
                            if (!valid || std::memcmp(hs + 1, qs + 1, 11 * sizeof(double)) != 0) continue;
                            const double dist = std::fabs(hs[0] - x);
                            if (warm < 0 || dist < best) {
                                warm = static_cast<int32_t>(ob.slot[role]);
                                best = dist;
                            }
                        }
                        if (warm < 0 && ob.eig_valid && !rebuilt[b]) warm = static_cast<int32_t>(ob.slot[ROLE_OWN]);   // another family: still better than a cold start
                        uint32_t dst;
                        if (target >= 0 && !(written[b] & (1u << target))) {
                            dst = ob.slot[target];
                            std::memcpy(ob.held[target], qs, sizeof qs);
                            ob.held_valid[target] = true;
                            written[b] |= 1u << target;
                        } else {
                            dst = static_cast<uint32_t>(transient0 + eng.evals.size());
                        }
                        k = eng.queue(b, dst, ob.tree_scale, qs, false, warm);
                        ++eig_evals;
                    }
                }
                pending.push_back({b, k, x});
            }
            if (!any) break;
        }
        ++rounds;
        evals += pending.size();
        const auto t1 = now();
        eng.run(false);
        const auto t2 = now();
        for (const Pending& q : pending) {
            const OmegaBlock& ob = *blk[q.block];
            double lpr = eng.value(q.eval, eng.lnz);
            lpr += ((ob.stage % 2) == 0) ? log_prior_rho(q.x) : log_prior_kappa(q.x);
            blk[q.block]->store.entries.push_back({q.x, lpr, 0.0, eng.failed(q.eval)});
        }
        for (const auto& q : validating) {
            OmegaBlock& ob = *blk[q.first];
            ob.needs_validation = false;
            if (eng.failed(q.second)) ob.failed = true;
            else start_stage(ob);
        }
        t_host += secs(t0, t1) + secs(t2, now());
        t_run += secs(t1, t2);
    }
    if (timing) {
        size_t used = 0, unused = 0;
        for (size_t b : live) {
            blk[b]->store.tally();
            used += blk[b]->store.n_used;
            unused += blk[b]->store.n_unused;
        }
        fprintf(stderr, "run_omega: %zu evaluations consumed by the optimisers, %zu speculative ones never asked for\n", used, unused);
    }
    if (timing)
        fprintf(stderr, "run_omega: %zu blocks (%zu on the factored path), %zu rounds, %zu evaluations (%zu with a new eigen-system, %zu kappa probes on a held one, %zu stage ends adopted one): host %.3f s, device rounds %.3f s; engine set-up %.3f s, validation passes + F3x4 counts %.3f s, whole call %.3f s\n",
                n_blocks, static_cast<size_t>(std::count(eng.block_factored.begin(), eng.block_factored.end(), 1)), rounds, evals, eig_evals, eig_reused, eig_adopted, t_host, t_run, t_setup, t_prelude, std::chrono::duration<double>(std::chrono::steady_clock::now() - t_enter).count());
    for (size_t b = 0; b < n_blocks; ++b) {
        const OmegaBlock& ob = *blk[b];
        out_phylo[b] = ob.failed ? NAN : decibans(ob.lpr_h1, ob.lpr_h0);
        if (out_anc) out_anc[b] = NAN;  // src/run.hpp:181
    }
    return PCSF_OK;
}

}  // namespace


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
                    model != 3 && list_prefers_tiles(c, codons ? codons + site_offsets[0] : nullptr, leaf_stride, off.data(), n_blocks), false,
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
            fprintf(stderr, "eigen stamps (job 0, %s start; s_memtime cycles): %llu sweeps, %llu steps, kernel %llu; per step: phase A %.0f, phase B %.0f (wave 0 rotations %.0f, wave 1 V update %.0f)\n",
                    pass == 0 ? "cold" : (warm_settings ? "warm" : "cold"), h[5], h[4], h[6], h[0] / st, h[3] / st, h[1] / st, h[2] / st);
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
    // MLE / OMEGA keep one set of branch-matrix tables per evaluation of a LONG block in HBM (9.6 MB for a 100-leaf tree),
    // up to two evaluations per block and round: blocks are independent, so a long list is processed in chunks that bound
    // that footprint to ~16 GB.
    // PCSF_BLOCK_GROUPS=g advances g groups of blocks concurrently, one host thread and one pair of streams each.  A round's
    // time is the latency of its longest dependency chain (eigen-solver sweeps -> edge scales -> pruning -> download) plus the
    // host's work on the round, and every group needs as many rounds as the whole list (the block with the longest optimiser
    // path sets it): short lists lose (516 blocks OMEGA 0.32 s as one group, 0.39 s as two), long lists win because one
    // group's host work and small kernels run beside the other's (8,256 blocks: MLE 0.58 -> 0.31 s, OMEGA 2.85 -> 2.18 s;
    // four groups add nothing).  Blocks are independent: grouping never changes a result (tests/test_gpu_blocks.py).
    size_t chunk = n_blocks;
    size_t n_groups = 1;
    // one pruning kernel for the whole list, decided before it is cut into chunks and groups
    const bool tiles = algo != PCSF_FIXED && list_prefers_tiles(c, codons, leaf_stride, site_offsets, n_blocks);
    if (algo != PCSF_FIXED && std::getenv("PCSF_TIME_BLOCKS")) fprintf(stderr, "pruning kernel of this list: %s\n", tiles ? "prune_tiles" : "prune_factored");
    if (algo != PCSF_FIXED) {
        const size_t ne = c->schedule.slot_node.size();
        const size_t per_table = (ne * 4096 + static_cast<size_t>(c->n_leaves) * LEAF_TABLE_DOUBLES_DEV) * sizeof(double);
        chunk = std::max<size_t>(64, (size_t(16) << 30) / std::max<size_t>(per_table, 1));
        // Blocks on the factored path keep no tables: when the whole list is factored (score-msa's short, sparse blocks) the chunks
        // are bounded by the eigen-system slots instead (3 per block, ~100 KB each) -- every chunk costs a full set of rounds.
        long max_tiles = -1;
        if (const char* v = std::getenv("PCSF_FACTORED_MAX_TILES")) max_tiles = std::max(0, std::atoi(v));
        std::atomic<bool> any_explicit{false};
        if (tiles) {
            // no block keeps tables
        } else if (codons)
            parallel_for(n_blocks, [&](size_t b) {
                if (!any_explicit.load(std::memory_order_relaxed) &&
                    !block_prefers_factored(c, codons, leaf_stride, site_offsets[b], site_offsets[b + 1], max_tiles))
                    any_explicit.store(true, std::memory_order_relaxed);
            });
        else any_explicit = true;
        if (!any_explicit) chunk = std::max<size_t>(chunk, 32768);
        // two groups once the list is long enough for one group's host work (optimisers, job lists: ~3 ms per round at 8,256
        // blocks) to matter beside the other's kernels -- measured: MLE from 512 blocks, OMEGA from 1,536 (profiles/r02/blocks_groups.txt)
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
    auto run_range = [&](const Range& r, bool own_streams) -> int {
        std::vector<uint64_t> off(r.nb + 1);  // CSR offsets re-based at the range's first site
        for (size_t b = 0; b <= r.nb; ++b) off[b] = site_offsets[r.b0 + b] - site_offsets[r.b0];
        const uint8_t* cod = codons ? codons + site_offsets[r.b0] : nullptr;
        float* ph = out_phylo + r.b0;
        float* an = out_anc ? out_anc + r.b0 : nullptr;
        switch (algo) {
            case PCSF_FIXED: return run_fixed(c, cod, off.data(), r.nb, leaf_stride, compute_anc, ph, an);
            case PCSF_MLE: return run_mle(c, cod, off.data(), r.nb, leaf_stride, compute_anc, seed, ph, an, tiles, own_streams, &stop_groups);
            default: return run_omega(c, cod, off.data(), r.nb, leaf_stride, seed, ph, an, tiles, own_streams, &stop_groups);
        }
    };
    if (n_groups <= 1 || ranges.size() <= 1) {
        for (const Range& r : ranges) {
            const int rc = run_range(r, false);
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
    for (size_t g = 0; g < std::min(n_groups, ranges.size()); ++g)
        th.emplace_back([&]() {
            const hipError_t dev_rc = hipSetDevice(c->device);  // the current device is per host thread
            for (size_t i = next++; i < ranges.size(); i = next++) {
                int rc = PCSF_OK;
                std::string msg;
                try {
                    if (dev_rc != hipSuccess) throw std::runtime_error(std::string("hipSetDevice: ") + hipGetErrorString(dev_rc));
                    rc = run_range(ranges[i], true);
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
