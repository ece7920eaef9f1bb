// chain.hip -- the MLE / OMEGA optimiser loop ON THE DEVICE (reference: max_lik_lpr_leaves / fit_find_init src/fixed_lik.hpp:469-544,
// the rho / kappa coordinate steps src/omega.hpp:204-234, the schedule of run() src/run.hpp:136-208).
//
// `advance_blocks<Machine>`: one wave per block.  The wave (1) adds up, in site order, the per-site ln z (and ancestral terms) that the
// pruning kernel of the previous round wrote for each of the block's pending evaluations -- one lane per evaluation --, (2) runs the
// block's state machine (machine.hpp: restart search, GSL Brent with its d/e swap, OMEGA's twelve stages with the held eigen-systems)
// in lane 0 on the state kept in HBM, and (3) writes, all lanes together, the job records the kernels of THIS round read: eigen jobs
// (eigen.hip), evaluations (edge_scale_kernel), tile jobs and tile groups (tiles.hip) or tile jobs (factored.hip), appended to the
// round's lists through a handful of atomic counters.  The host does not see a block's state between the first round and the last:
// per round it reads the counters (64 bytes) to size the launches, nothing else.  The order of the records in the lists is whatever
// the atomics make it; every record names its own output slot, so no result depends on it.
//
// Evaluation slots: block b's i-th request of a round owns table b * P + i (edge-scale row, failure flag) and the output range
// P * off[b] + i * sites(b) .. + sites(b).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"
// diagnostics (-DPCSF_ADVANCE_STAMPS=1, tools/build_variant.sh): work-group PCSF_ADVANCE_STAMP_BLOCK prints the s_memtime cycles of its phases at every launch
#ifndef PCSF_ADVANCE_STAMPS
#define PCSF_ADVANCE_STAMPS 0
#endif
#ifndef PCSF_ADVANCE_STAMP_BLOCK
#define PCSF_ADVANCE_STAMP_BLOCK 0
#endif
#include "machine.hpp"

namespace pcsf {

namespace {

// The log priors of the OMEGA objective (src/omega.hpp:130-149), with the device's math library.  BOTH engines take them from here
// (the host engine through prior_kernel), so both add the same bits; against glibc's they differ by at most an ulp of the prior.
__device__ double log_prior_rho_dev(double rho)
{
    const double mode = 1.0, scale = 0.5, pi = 3.14159265358979323846;
    const double numer = 1.0 / (pi * scale * (1.0 + pow(((rho - mode) / scale), 2.0)));
    const double cdf0 = atan((0.0 - mode) / scale) / pi + 0.5;
    return log(numer) - log(1.0 - cdf0);
}
__device__ double log_prior_kappa_dev(double kappa)
{
    const double k = kappa - 1.0 + 2.2204460492503131e-16, a = 7.0, b = 0.25;
    const double pdf = (k <= 0.0) ? 0.0 : exp((a - 1) * log(k / b) - k / b - lgamma(a)) / b;
    return log(pdf);
}
struct DevicePrior {
    __device__ double operator()(uint32_t, uint32_t kind, double x) const { return kind == EVAL_RHO ? log_prior_rho_dev(x) : log_prior_kappa_dev(x); }
};
// ... made by one lane per pending evaluation before lane 0 runs the machine (pow, atan, lgamma: ~2,000 cycles each, up to twelve per round)
struct LanePrior {
    const double* v;
    __device__ double operator()(uint32_t i, uint32_t, double) const { return v[i]; }
};

__global__ void prior_kernel(const uint32_t* __restrict__ kinds, const double* __restrict__ xs, uint32_t n, double* __restrict__ out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = DevicePrior()(i, kinds[i], xs[i]);
}

template <class M> struct MachineTraits;
template <> struct MachineTraits<MleMachine> {
    static constexpr int MAX_REQ = MLE_MAX_REQUESTS;
    static constexpr int STORES = 2;
};
template <> struct MachineTraits<OmegaMachine> {
    static constexpr int MAX_REQ = OMEGA_MAX_REQUESTS;
    static constexpr int STORES = 1;
};

// What a wave works on sits in LDS while it works: the machine (~1 KB), the stores' values (a search has seen 10-40 when it ends) and the
// per-site values to be added up.  The round's lists are appended to through a dozen global counters; a block first counts what it will
// append (lane 0, with the machine's requests in LDS), then sixteen lanes make the block's atomics AT ONCE -- one round trip instead of
// two per run of requests -- and the records are written from the bases they return.  (ADVANCE_BLOCKS > 1 lets several blocks share a
// work-group and its atomics: a returning atomic on ONE word is served at ~88 per microsecond chip-wide, MI355X_MICROARCH.md "dequeue".
// Measured with four: SLOWER -- 0.155 -> 0.28 ms per launch on configs[2], 8,256 real blocks MLE 0.14 -> 0.62 s: a work-group then lives as
// long as its slowest block and four blocks' LDS images cut the occupancy.  One block per work-group it stays.)
constexpr uint32_t SUM_CHUNK = 64;      // the per-site sums' buffer holds SUM_CHUNK sites of MAX_REQ evaluations
constexpr int ADVANCE_BLOCKS = 1;
// counters of a round a block appends to: n_evals[2], n_tile_jobs[2], n_groups[2][4], n_eig_jobs, n_list[2], n_active, n_transient
enum AdvanceCounter { AC_EVALS = 0, AC_TILE_JOBS = 2, AC_GROUPS = 4, AC_EIG = 12, AC_LIST = 13, AC_ACTIVE = 15, AC_TRANSIENT = 16, AC_COUNT = 17 };

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// diagnostics (-DPCSF_ADVANCE_STAMPS=1, tools/build_variant.sh): work-group PCSF_ADVANCE_STAMP_BLOCK prints the s_memtime cycles of its phases at every launch
#ifndef PCSF_ADVANCE_STAMPS
#define PCSF_ADVANCE_STAMPS 0
#endif
#ifndef PCSF_ADVANCE_STAMP_BLOCK
#define PCSF_ADVANCE_STAMP_BLOCK 0
#endif
__device__ __forceinline__ unsigned long long adv_clock()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}
#define ADV_STAMP(var) const unsigned long long var = PCSF_ADVANCE_STAMPS ? adv_clock() : 0ull

template <class M>
__global__ __launch_bounds__(ADVANCE_BLOCKS * 64) void advance_blocks(const ChainParams p, uint32_t n_blocks)
{
    ADV_STAMP(t_start);
    constexpr int MAX_REQ = MachineTraits<M>::MAX_REQ;
    constexpr int STORES = MachineTraits<M>::STORES;
    constexpr uint32_t M_WORDS = (sizeof(M) + 3) / 4;
    __shared__ EvalResult res_all[ADVANCE_BLOCKS][MAX_REQ];
    __shared__ EvalRequest req_all[ADVANCE_BLOCKS][MAX_REQ];
    __shared__ __attribute__((aligned(16))) uint32_t m_words_all[ADVANCE_BLOCKS][M_WORDS];
    __shared__ StoreEntry st_lds_all[ADVANCE_BLOCKS][STORES * STORE_ENTRIES];
    __shared__ __attribute__((aligned(16))) double sum_buf_all[ADVANCE_BLOCKS][MAX_REQ * SUM_CHUNK + 2 * MAX_REQ];
    __shared__ double prior_all[ADVANCE_BLOCKS][MAX_REQ];
    __shared__ Lookahead ahead_all[ADVANCE_BLOCKS][2];   // the machine's list of upcoming points (as a local of round(): scratch memory, ~500 cycles an access)
    __shared__ int n_req_all[ADVANCE_BLOCKS];
    __shared__ uint32_t cnt_all[ADVANCE_BLOCKS][AC_COUNT];   // what the wave's block appends to every counter; then its offset within the work-group's range
    __shared__ uint32_t base_all[AC_COUNT];
    __shared__ uint32_t dense_all[ADVANCE_BLOCKS][MAX_REQ];   // a request's position in its set's list of evaluations (prune_factored's edge-scale row)
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t b = blockIdx.x * ADVANCE_BLOCKS + wave;
    const bool live = b < n_blocks;
    EvalResult* res = res_all[wave];
    EvalRequest* req = req_all[wave];
    uint32_t* m_words = m_words_all[wave];
    StoreEntry* st_lds = st_lds_all[wave];
    double* sum_buf = sum_buf_all[wave];
    double* prior_lds = prior_all[wave];
    uint32_t* cnt = cnt_all[wave];
    uint32_t* dense = dense_all[wave];
    if (lane < AC_COUNT) cnt[lane] = 0u;
    uint64_t s0 = 0, sites = 0;
    const uint32_t P = p.slots_per_block;
    uint32_t n_tiles = 0;
    int n_req = 0;
    uint32_t* m_global = nullptr;
    StoreEntry* store_global = nullptr;
    if (live) {
        m_global = reinterpret_cast<uint32_t*>(reinterpret_cast<M*>(p.machines) + b);
        for (uint32_t w = lane; w < M_WORDS; w += 64) m_words[w] = m_global[w];
        wave_sync();
        M* m = reinterpret_cast<M*>(m_words);
        s0 = p.off[b];
        sites = p.off[b + 1] - s0;
        n_tiles = static_cast<uint32_t>((sites + 15) / 16);
        // the stores' values (machine.hpp store_compact bounds them: they always fit)
        store_global = p.stores + static_cast<size_t>(b) * STORES * STORE_ENTRIES;
        for (int k = 0; k < STORES; ++k) {
            uint32_t n_in;
            if constexpr (STORES == 2) n_in = m->n_store[k];
            else n_in = m->n_store;
            const uint32_t* src = reinterpret_cast<const uint32_t*>(store_global + static_cast<size_t>(k) * STORE_ENTRIES);
            uint32_t* dst = reinterpret_cast<uint32_t*>(st_lds + k * STORE_ENTRIES);
            for (uint32_t w = lane; w < n_in * (sizeof(StoreEntry) / 4); w += 64) dst[w] = src[w];
        }
        // (1) the values of the pending evaluations: per-site terms through LDS in chunks of SUM_CHUNK sites (coalesced), then lane e adds
        // evaluation e's values in site order
        const uint32_t n_pending = m->n_pending;
        ADV_STAMP(t_loaded);
        {
            // A chunk of sites of every pending evaluation goes global -> LDS rows (coalesced; rows an even number of doubles plus two apart:
            // the adding lanes' 16-byte reads do not collide), and lane e adds row e in site order -- the order is the result's definition
            // (the host engine and the oracle add a block's sites first to last), so the adds are one dependent chain per evaluation; what
            // can overlap does: sixteen values per LDS round trip instead of one (the first version: 64 dependent LDS reads per 64-site
            // chunk, 6,000 cycles a chunk, 54,000 for a 542-site block with ONE pending evaluation), and the chunk is as long as the
            // buffer allows for the evaluations there are -- SUM_CHUNK * MAX_REQ doubles: a whole 700-site block when one is pending,
            // 64 sites when twelve are -- so a typical round pays the trip to HBM once.
            double l = 0.0, a = 0.0;
            const uint32_t chunk = n_pending ? (SUM_CHUNK * MAX_REQ / n_pending) & ~63u : 0u, stride = chunk + 2;
            for (int pass = 0; pass < ((p.out_anc && n_pending) ? 2 : (n_pending ? 1 : 0)); ++pass) {
                const double* __restrict__ src = (pass ? p.out_anc : p.out_lnz) + static_cast<size_t>(P) * s0;
                double acc = 0.0;
                // 64-site pieces, evaluation by evaluation: at most MAX_REQ of them per chunk and lane, all in flight at once, and the NEXT chunk's
                // while this one is added (left to a loop per evaluation the compiler waits for every load before it issues the next)
                const uint32_t kpe = chunk >> 6, pieces = n_pending * kpe;
                double pre[MAX_REQ];
                auto fetch = [&](uint64_t c0, uint32_t len) {
                    uint32_t e_at = 0, k_at = 0;
#pragma unroll
                    for (int q = 0; q < MAX_REQ; ++q) {
                        const uint32_t i = k_at * 64 + lane;
                        pre[q] = (static_cast<uint32_t>(q) < pieces && i < len) ? src[static_cast<size_t>(e_at) * sites + c0 + i] : 0.0;
                        if (++k_at == kpe) {
                            k_at = 0;
                            ++e_at;
                        }
                    }
                };
                auto chunk_len = [&](uint64_t c0) { return static_cast<uint32_t>(sites - c0 < chunk ? sites - c0 : chunk); };
                if (sites) fetch(0, chunk_len(0));
                for (uint64_t c0 = 0; c0 < sites; c0 += chunk) {
                    const uint32_t len = chunk_len(c0);
                    wave_sync();   // the rows of the chunk before have been added
                    {
                        uint32_t e_at = 0, k_at = 0;
#pragma unroll
                        for (int q = 0; q < MAX_REQ; ++q) {
                            const uint32_t i = k_at * 64 + lane;
                            if (static_cast<uint32_t>(q) < pieces && i < len) sum_buf[e_at * stride + i] = pre[q];
                            if (++k_at == kpe) {
                                k_at = 0;
                                ++e_at;
                            }
                        }
                    }
                    if (c0 + chunk < sites) fetch(c0 + chunk, chunk_len(c0 + chunk));
                    wave_sync();
                    if (lane < n_pending) {
                        typedef double double2_sum __attribute__((ext_vector_type(2)));
                        const double* row = sum_buf + lane * stride;
                        uint32_t i = 0;
                        for (; i + 16 <= len; i += 16) {
                            double2_sum v[8];
#pragma unroll
                            for (int k = 0; k < 8; ++k) v[k] = *reinterpret_cast<const double2_sum*>(row + i + 2 * k);
#pragma unroll
                            for (int k = 0; k < 8; ++k) {
                                acc += v[k].x;
                                acc += v[k].y;
                            }
                        }
                        for (; i < len; ++i) acc += row[i];
                    }
                }
                (pass ? a : l) = acc;
            }
            if (lane < n_pending) {
                res[lane].lnz = l;
                res[lane].anc = a;
                res[lane].failed = p.fail[static_cast<size_t>(b) * P + lane];
                if constexpr (STORES == 1) prior_lds[lane] = DevicePrior()(lane, m->pend_kind[lane], m->pend_x[lane]);   // (OMEGA: the objective's log prior)
            }
        }
        wave_sync();
        ADV_STAMP(t_summed);
        // (2) the block's machine
        if (lane == 0) {
            int n;
            // (the stores always sit in LDS: a call on a pointer chosen between LDS and HBM, or two calls the compiler merges into one, trip
            // the gfx950 back end of ROCm 7.2, "Illegal instruction detected: V_CMP_NE_U32_e32 0, $src_shared_base")
            if constexpr (STORES == 2) {
                n = m->round(res, st_lds, STORE_ENTRIES, DrawTable{p.draws[0], p.n_draws}, p.eager != 0, static_cast<int>(p.max_points), static_cast<int>(p.max_restarts), p.out_anc != nullptr, req,
                             ahead_all[wave]);
            } else {
                n = m->round(res, st_lds, DrawTable{p.draws[1], p.n_draws}, DrawTable{p.draws[2], p.n_draws}, p.eager != 0, static_cast<int>(p.max_points),
                             static_cast<int>(p.max_restarts), p.held_systems != 0, LanePrior{prior_lds}, req, ahead_all[wave]);
            }
            n_req_all[wave] = n;
            // what this block appends to the round's lists
            uint32_t* c = cnt;   // (zeroed at the kernel's start; a local array indexed at run time would live in scratch memory)
            c[AC_ACTIVE] = m->active ? 1u : 0u;
            for (int i0 = 0; i0 < n;) {
                const uint32_t eig = req[i0].eig, set = req[i0].new_eigen ? 1u : 0u;
                int i1 = i0 + 1;
                while (i1 < n && req[i1].eig == eig && (req[i1].new_eigen ? 1u : 0u) == set && !p.pack_models) ++i1;
                const uint32_t run_tiles = n_tiles * static_cast<uint32_t>(i1 - i0);
                for (int i = i0; i < i1; ++i) {
                    c[AC_EVALS + set] += 1u;
                    if (req[i].new_eigen == 1u) c[AC_EIG] += 1u;
                    if (req[i].eig & TRANSIENT_EIG) c[AC_TRANSIENT] += 1u;
                }
                if (p.pack_models) {
                    c[AC_LIST + eig] += run_tiles;
                } else {
                    c[AC_TILE_JOBS + set] += run_tiles;
                    if (p.use_tiles) {
                        const uint32_t parts = (run_tiles + TILE_JOBS_PER_GROUP - 1) / TILE_JOBS_PER_GROUP;
                        for (uint32_t k = 0; k < parts; ++k) {
                            const uint32_t size = run_tiles * (k + 1) / parts - run_tiles * k / parts;
                            c[AC_GROUPS + 4 * set + (p.split_sizes ? (size < TILE_JOBS_PER_GROUP ? size - 1 : TILE_JOBS_PER_GROUP - 1) : 0u)] += 1u;
                        }
                    }
                }
                i0 = i1;
            }
        }
        wave_sync();
        ADV_STAMP(t_machine);
        if (PCSF_ADVANCE_STAMPS && b == PCSF_ADVANCE_STAMP_BLOCK && lane == 0)
        {
            printf("advance_blocks block %u (%u sites, %u pending): load %llu, sums %llu, machine %llu cycles; %d requests\n", b, static_cast<unsigned>(sites), n_pending,
                   t_loaded - t_start, t_summed - t_loaded, t_machine - t_summed, n_req_all[wave]);
#if PCSF_ADVANCE_STAMPS
            if constexpr (STORES == 1)
                printf("   machine: values in %llu, searches fed %llu, (validation / set-up) %llu, upcoming points %llu, requests built %llu\n", g_machine_stamp[1] - g_machine_stamp[0],
                       g_machine_stamp[2] - g_machine_stamp[1], 0ull, g_machine_stamp[3] - g_machine_stamp[2], g_machine_stamp[4] - g_machine_stamp[3]);
#endif
        }
        n_req = n_req_all[wave];
        // the machine and what the stores hold now, back to HBM
        for (uint32_t w = lane; w < M_WORDS; w += 64) m_global[w] = m_words[w];
        for (int k = 0; k < STORES; ++k) {
            uint32_t n_now;
            if constexpr (STORES == 2) n_now = m->n_store[k];
            else n_now = m->n_store;
            uint32_t* dst = reinterpret_cast<uint32_t*>(store_global + static_cast<size_t>(k) * STORE_ENTRIES);
            const uint32_t* src = reinterpret_cast<const uint32_t*>(st_lds + k * STORE_ENTRIES);
            for (uint32_t w = lane; w < n_now * (sizeof(StoreEntry) / 4); w += 64) dst[w] = src[w];
        }
    }
    // (3) one atomic per counter and work-group; a wave's range starts at the base plus what the waves before it append
    __syncthreads();
    if (threadIdx.x < AC_COUNT) {
        const int k = threadIdx.x;
        uint32_t total = 0;
        for (int w = 0; w < ADVANCE_BLOCKS; ++w) {
            const uint32_t mine = cnt_all[w][k];
            cnt_all[w][k] = total;
            total += mine;
        }
        uint32_t* counter = k < AC_TILE_JOBS ? &p.counters->n_evals[k - AC_EVALS]
                            : k < AC_GROUPS ? &p.counters->n_tile_jobs[k - AC_TILE_JOBS]
                            : k < AC_EIG    ? &p.counters->n_groups[(k - AC_GROUPS) >> 2][(k - AC_GROUPS) & 3]
                            : k < AC_LIST   ? &p.counters->n_eig_jobs
                            : k < AC_ACTIVE ? &p.counters->n_list[k - AC_LIST]
                            : k == AC_ACTIVE ? &p.counters->n_active
                                             : &p.counters->n_transient;
        base_all[k] = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    if (!live) return;
    // (4) this round's job records.  Set 1: evaluations that wait for an eigen-system solved in this round; set 0: the others.
    uint32_t at[AC_COUNT];   // next free index of every list for this block (all lanes keep the same values)
#pragma unroll
    for (int k = 0; k < AC_COUNT; ++k) at[k] = base_all[k] + cnt[k];
    // the probes of this round that are not held get the next slots of the round's transient range (machine.hpp TRANSIENT_EIG)
    if (static_cast<int>(lane) < n_req && (req[lane].eig & TRANSIENT_EIG)) req[lane].eig = p.transient0 + at[AC_TRANSIENT] + (req[lane].eig & ~TRANSIENT_EIG);
    wave_sync();
    {
        // evaluations and eigen jobs: lane i writes request i's records; its indices are the counts of the requests before it
        uint32_t my_eval = 0, my_eig = 0;
        if (static_cast<int>(lane) < n_req) {
            const uint32_t set = req[lane].new_eigen ? 1u : 0u;
            for (uint32_t i = 0; i < lane; ++i) {
                if ((req[i].new_eigen ? 1u : 0u) == set) ++my_eval;
                if (req[i].new_eigen == 1u) ++my_eig;
            }
            const EvalRequest& r = req[lane];
            const uint32_t table = b * P + lane;
            p.fail[table] = 0u;
            ExpmEval ev;
            ev.eig = r.eig;
            ev.table = table;
            ev.block = b;
            ev.pad = p.use_tiles;
            ev.x = r.tree_scale;
            if (p.eval_x) p.eval_x[table] = r.tree_scale;
            p.evals[static_cast<size_t>(set) * p.eval_capacity + at[AC_EVALS + set] + my_eval] = ev;
            dense[lane] = at[AC_EVALS + set] + my_eval;
            if (r.new_eigen == 1u) {
                EigenJob j;
                j.eig = r.eig;
                j.table = table;
                j.warm = r.warm;
                j.pad = 0;
                for (int k = 0; k < 12; ++k) j.qs[k] = r.qs[k];
                p.eig_jobs[at[AC_EIG] + my_eig] = j;
            }
        }
    }
    wave_sync();
    // tile jobs: requests that share an eigen-system (the rho points of a round all use the block's own; MLE: the model's) are laid out
    // together, so that their tiles can share work-groups
    for (int i0 = 0; i0 < n_req;) {
        const uint32_t eig = req[i0].eig, set = req[i0].new_eigen ? 1u : 0u;
        int i1 = i0 + 1;
        while (i1 < n_req && req[i1].eig == eig && (req[i1].new_eigen ? 1u : 0u) == set && !p.pack_models) ++i1;
        const uint32_t run_tiles = n_tiles * static_cast<uint32_t>(i1 - i0);
        uint32_t job0;
        if (p.pack_models) {
            job0 = eig * p.list_capacity + at[AC_LIST + eig];
            at[AC_LIST + eig] += run_tiles;
        } else {
            job0 = at[AC_TILE_JOBS + set];
            at[AC_TILE_JOBS + set] += run_tiles;
        }
        FactoredJob* __restrict__ jobs = p.tile_jobs + (p.pack_models ? 0 : static_cast<size_t>(set) * p.job_capacity);
        for (uint32_t t = lane; t < run_tiles; t += 64) {
            const uint32_t i = static_cast<uint32_t>(i0) + t / n_tiles, tile = t % n_tiles;
            FactoredJob j;
            j.site0 = s0 + 16ull * tile;
            j.out0 = static_cast<uint64_t>(P) * s0 + static_cast<uint64_t>(i) * sites + 16ull * tile;
            j.n_sites = static_cast<uint32_t>(sites - 16ull * tile < 16 ? sites - 16ull * tile : 16);
            j.eig = eig;
            j.eval = (p.use_tiles || p.own_scales) ? b * P + i : dense[i];   // prune_tiles: the evaluation slot (eval_x, failure flag); prune_factored: the evaluation's row of edge scales, dense per set
            j.pad = tile == 0 ? FACTORED_JOB_VALIDATES : 0u;   // (read by prune_tiles when it makes its own edge scales)
            jobs[job0 + t] = j;
        }
        if (p.use_tiles && !p.pack_models) {
            // the run's tiles in even parts of at most TILE_JOBS_PER_GROUP (one wave per tile, one eigen-system per work-group);
            // a part goes to the list of its size, so that a work-group is launched with as many waves as it has tiles
            const uint32_t parts = (run_tiles + TILE_JOBS_PER_GROUP - 1) / TILE_JOBS_PER_GROUP;
            for (uint32_t k = 0; k < parts; ++k) {   // (every lane walks the parts: `at` stays the same in all of them; lane 0 writes)
                const uint32_t a = run_tiles * k / parts, e = run_tiles * (k + 1) / parts, size = e - a;
                const uint32_t list = p.split_sizes ? (size < TILE_JOBS_PER_GROUP ? size - 1 : TILE_JOBS_PER_GROUP - 1) : 0u;
                const uint32_t g = at[AC_GROUPS + 4 * set + list]++;
                if (lane == 0) p.groups[(static_cast<size_t>(set) * 4 + list) * p.group_capacity + g] = TileGroup{job0 + a, size};
            }
        }
        i0 = i1;
    }
}

}  // namespace

hipError_t launch_advance_blocks(const ChainParams& p, int algo, uint32_t n_blocks, hipStream_t stream)
{
    if (n_blocks == 0) return hipSuccess;
    const uint32_t grid = (n_blocks + ADVANCE_BLOCKS - 1) / ADVANCE_BLOCKS;
    if (algo == 0) hipLaunchKernelGGL(advance_blocks<MleMachine>, dim3(grid), dim3(ADVANCE_BLOCKS * 64), 0, stream, p, n_blocks);
    else hipLaunchKernelGGL(advance_blocks<OmegaMachine>, dim3(grid), dim3(ADVANCE_BLOCKS * 64), 0, stream, p, n_blocks);
    return hipGetLastError();
}

hipError_t launch_log_priors(const uint32_t* d_kinds, const double* d_xs, uint32_t n, double* d_out, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(prior_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, d_kinds, d_xs, n, d_out);
    return hipGetLastError();
}

// ---- tests: the 1-D maximiser alone on the device, fed with a recorded sequence of objective values ----------------------------
// The search asks for x_0, is given values[0], asks for x_1, ... -- equal histories give equal requests, so a search that takes the
// same decisions as the one that produced `values` asks for exactly its points (tests/test_optimizer.py: against the oracle's).
namespace {
__global__ void maxlik_replay_kernel(double init, double lo, double hi, const double* __restrict__ draws, uint32_t n_draws, const double* __restrict__ values,
                                     int n_values, double* __restrict__ xs_out, int cap, int* __restrict__ n_requests, double* __restrict__ last)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    MaxLik1D opt;
    opt.start(init, lo, hi);
    const DrawTable dt{draws, n_draws};
    uint32_t next = 0;
    int n = 0;
    while (!opt.done() && n < n_values) {
        if (n < cap) xs_out[n] = opt.request();
        opt.supply(values[n], 0.0, dt, next);
        ++n;
    }
    if (!opt.done() && n < cap) xs_out[n] = opt.request();   // the point it is waiting for (tools/mle_divergence.py drives a search step by step)
    *n_requests = opt.done() ? n : -1;   // -1: the search wanted more values than the recording holds (it took another path)
    last[0] = opt.last_lpr();
    last[1] = opt.last_x();
}
}  // namespace

hipError_t launch_maxlik_replay(double init, double lo, double hi, const double* d_draws, uint32_t n_draws, const double* d_values, int n_values, double* d_xs_out,
                                int cap, int* d_n_requests, double* d_last, hipStream_t stream)
{
    hipLaunchKernelGGL(maxlik_replay_kernel, dim3(1), dim3(64), 0, stream, init, lo, hi, d_draws, n_draws, d_values, n_values, d_xs_out, cap, d_n_requests, d_last);
    return hipGetLastError();
}

}  // namespace pcsf
