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

// What the wave works on sits in LDS while it works: the machine (~1 KB), the stores' values (a search has seen 10-40 when it ends) and the
// per-site values to be added up.  Lane 0's walk through the state machine is a chain of dependent reads -- from HBM it was ~0.4 ms per
// round whatever the number of blocks (profiles/r04), every round of every group waiting for it.
constexpr uint32_t SUM_CHUNK = 64;

template <class M>
__global__ __launch_bounds__(64) void advance_blocks(const ChainParams p)
{
    constexpr int MAX_REQ = MachineTraits<M>::MAX_REQ;
    constexpr int STORES = MachineTraits<M>::STORES;
    constexpr uint32_t M_WORDS = (sizeof(M) + 3) / 4;
    __shared__ EvalResult res[MAX_REQ];
    __shared__ EvalRequest req[MAX_REQ];
    __shared__ __attribute__((aligned(16))) uint32_t m_words[M_WORDS];
    __shared__ StoreEntry st_lds[STORES * STORE_ENTRIES];
    __shared__ double sum_buf[MAX_REQ * SUM_CHUNK];
    __shared__ int n_req_s;
    __shared__ uint32_t base_s[4];
    const uint32_t b = blockIdx.x, lane = threadIdx.x;
    uint32_t* m_global = reinterpret_cast<uint32_t*>(reinterpret_cast<M*>(p.machines) + b);
    for (uint32_t w = lane; w < M_WORDS; w += 64) m_words[w] = m_global[w];
    __syncthreads();
    M* m = reinterpret_cast<M*>(m_words);
    const uint64_t s0 = p.off[b], s1 = p.off[b + 1], sites = s1 - s0;
    const uint32_t P = p.slots_per_block;
    // the stores' values: into LDS when they (and this round's new ones) fit
    StoreEntry* store_global = p.stores + static_cast<size_t>(b) * STORES * STORE_ENTRIES;
    for (int k = 0; k < STORES; ++k) {
        uint32_t n_in;
        if constexpr (STORES == 2) n_in = m->n_store[k];
        else n_in = m->n_store;
        const uint32_t* src = reinterpret_cast<const uint32_t*>(store_global + static_cast<size_t>(k) * STORE_ENTRIES);
        uint32_t* dst = reinterpret_cast<uint32_t*>(st_lds + k * STORE_ENTRIES);
        for (uint32_t w = lane; w < n_in * (sizeof(StoreEntry) / 4); w += 64) dst[w] = src[w];
    }
    // (1) the values of the pending evaluations: per-site terms through LDS in chunks (coalesced), lane e adds evaluation e's in site order
    const uint32_t n_pending = m->n_pending;
    {
        double l = 0.0, a = 0.0;
        for (int pass = 0; pass < (p.out_anc ? 2 : 1); ++pass) {
            const double* __restrict__ src = (pass ? p.out_anc : p.out_lnz) + static_cast<size_t>(P) * s0;
            double acc = 0.0;
            for (uint64_t c0 = 0; c0 < sites; c0 += SUM_CHUNK) {
                const uint32_t len = static_cast<uint32_t>(sites - c0 < SUM_CHUNK ? sites - c0 : SUM_CHUNK);
                __syncthreads();
                if (lane < len)
                    for (uint32_t e = 0; e < n_pending; ++e) sum_buf[e * SUM_CHUNK + lane] = src[static_cast<size_t>(e) * sites + c0 + lane];
                __syncthreads();
                if (lane < n_pending)
                    for (uint32_t i = 0; i < len; ++i) acc += sum_buf[lane * SUM_CHUNK + i];
            }
            (pass ? a : l) = acc;
        }
        if (lane < n_pending) {
            res[lane].lnz = l;
            res[lane].anc = a;
            res[lane].failed = p.fail[static_cast<size_t>(b) * P + lane];
        }
    }
    __syncthreads();
    // (2) the block's machine
    if (lane == 0) {
        int n;
        // (the stores always sit in LDS -- machine.hpp store_compact bounds them: a call on a pointer chosen between LDS and HBM, or two calls
        // the compiler merges into one, trip the gfx950 back end of ROCm 7.2, "Illegal instruction detected: V_CMP_NE_U32_e32 0, $src_shared_base")
        if constexpr (STORES == 2) {
            n = m->round(res, st_lds, STORE_ENTRIES, DrawTable{p.draws[0], p.n_draws}, p.eager != 0, static_cast<int>(p.max_points), static_cast<int>(p.max_restarts), p.out_anc != nullptr, req);
        } else {
            n = m->round(res, st_lds, DrawTable{p.draws[1], p.n_draws}, DrawTable{p.draws[2], p.n_draws}, p.eager != 0, static_cast<int>(p.max_points),
                         static_cast<int>(p.max_restarts), p.held_systems != 0, p.transient0 + b * P, DevicePrior(), req);
        }
        n_req_s = n;
        if (m->active) atomicAdd(&p.counters->n_active, 1u);
    }
    __syncthreads();
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
    // (3) this round's job records.  Set 1: evaluations that wait for an eigen-system solved in this round; set 0: the others.
    const int n_req = n_req_s;
    const uint32_t n_tiles = static_cast<uint32_t>((sites + 15) / 16);
    for (int i = lane; i < n_req; i += 64) {
        const EvalRequest& r = req[i];
        const uint32_t table = b * P + static_cast<uint32_t>(i), set = r.new_eigen ? 1u : 0u;
        p.fail[table] = 0u;
        ExpmEval ev;
        ev.eig = r.eig;
        ev.table = table;
        ev.block = b;
        ev.pad = p.use_tiles;
        ev.x = r.tree_scale;
        if (p.eval_x) p.eval_x[table] = r.tree_scale;
        p.evals[static_cast<size_t>(set) * p.eval_capacity + atomicAdd(&p.counters->n_evals[set], 1u)] = ev;
        if (r.new_eigen == 1u) {
            EigenJob j;
            j.eig = r.eig;
            j.table = table;
            j.warm = r.warm;
            j.pad = 0;
            for (int k = 0; k < 12; ++k) j.qs[k] = r.qs[k];
            p.eig_jobs[atomicAdd(&p.counters->n_eig_jobs, 1u)] = j;
        }
    }
    // tile jobs: requests that share an eigen-system (the rho points of a round all use the block's own; MLE: the model's) are laid out
    // together, so that their tiles can share work-groups
    for (int i0 = 0; i0 < n_req;) {
        const uint32_t eig = req[i0].eig, set = req[i0].new_eigen ? 1u : 0u;
        int i1 = i0 + 1;
        while (i1 < n_req && req[i1].eig == eig && (req[i1].new_eigen ? 1u : 0u) == set && !p.pack_models) ++i1;
        const uint32_t run_tiles = n_tiles * static_cast<uint32_t>(i1 - i0);
        if (lane == 0) {
            if (p.pack_models) base_s[0] = eig * p.list_capacity + atomicAdd(&p.counters->n_list[eig], run_tiles);
            else base_s[0] = atomicAdd(&p.counters->n_tile_jobs[set], run_tiles);
        }
        __syncthreads();
        const uint32_t job0 = base_s[0];
        FactoredJob* __restrict__ jobs = p.tile_jobs + (p.pack_models ? 0 : static_cast<size_t>(set) * p.job_capacity);
        for (uint32_t t = lane; t < run_tiles; t += 64) {
            const uint32_t i = static_cast<uint32_t>(i0) + t / n_tiles, tile = t % n_tiles;
            FactoredJob j;
            j.site0 = s0 + 16ull * tile;
            j.out0 = static_cast<uint64_t>(P) * s0 + static_cast<uint64_t>(i) * sites + 16ull * tile;
            j.n_sites = static_cast<uint32_t>(sites - 16ull * tile < 16 ? sites - 16ull * tile : 16);
            j.eig = eig;
            j.eval = b * P + i;
            j.pad = tile == 0 ? FACTORED_JOB_VALIDATES : 0u;   // (read by prune_tiles when it makes its own edge scales)
            jobs[job0 + t] = j;
        }
        if (p.use_tiles && !p.pack_models) {
            // the run's tiles in even parts of at most TILE_JOBS_PER_GROUP (one wave per tile, one eigen-system per work-group);
            // a part goes to the list of its size, so that a work-group is launched with as many waves as it has tiles
            const uint32_t parts = (run_tiles + TILE_JOBS_PER_GROUP - 1) / TILE_JOBS_PER_GROUP;
            for (uint32_t k = lane; k < parts; k += 64) {
                const uint32_t a = run_tiles * k / parts, e = run_tiles * (k + 1) / parts, size = e - a;
                const uint32_t list = p.split_sizes ? (size < TILE_JOBS_PER_GROUP ? size - 1 : TILE_JOBS_PER_GROUP - 1) : 0u;
                const uint32_t g = atomicAdd(&p.counters->n_groups[set][list], 1u);
                p.groups[(static_cast<size_t>(set) * 4 + list) * p.group_capacity + g] = TileGroup{job0 + a, size};
            }
        }
        __syncthreads();   // base_s is rewritten by the next run
        i0 = i1;
    }
}

}  // namespace

hipError_t launch_advance_blocks(const ChainParams& p, int algo, uint32_t n_blocks, hipStream_t stream)
{
    if (n_blocks == 0) return hipSuccess;
    if (algo == 0) hipLaunchKernelGGL(advance_blocks<MleMachine>, dim3(n_blocks), dim3(64), 0, stream, p);
    else hipLaunchKernelGGL(advance_blocks<OmegaMachine>, dim3(n_blocks), dim3(64), 0, stream, p);
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
