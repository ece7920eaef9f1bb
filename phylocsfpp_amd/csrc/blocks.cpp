// blocks.cpp -- run() for a batch of alignment blocks (reference src/run.hpp:57-211) on the GPU.
//
// FIXED : one tracks-mode launch over all sites (ln z and the ancestral term per site and model),
//         per-block sums on the host in site order.
// MLE   : every block maximises over the tree scale, coding model first, then non-coding, sharing
//         one mt19937 per block like the reference.  All blocks advance in lock-step: per round
//         the host collects one scale per active block, the GPU builds that block's 2*(n-1)... (n-1)
//         branch matrices (expm.hip) into the block's table set and prunes the block's sites with
//         them (block-mode kernel), and the per-site ln z come back for the optimisers.
// OMEGA : same machinery with the kappa/omega F3x4 rate matrix; rho rounds only re-exponentiate,
//         kappa rounds rebuild Q and its eigen-system (host, one thread per block).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <memory>
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

// Device state shared by the lock-step strategies
struct BlockEngine {
    pcsf_ctx* c;
    size_t n_blocks, leaf_stride, n_sites_total;
    const uint64_t* off;
    hipStream_t stream;
    int ne, nl, n;
    DeviceBuffer d_codons;                 // [nl][leaf_stride] compact copy
    DeviceBuffer d_frag, d_leaf, d_prior, d_ancw, d_fail;  // table sets, one per block slot
    DeviceBuffer d_eig_r, d_eig_l, d_eig_lam;              // eigen systems
    DeviceBuffer d_jobs, d_eig_jobs, d_wg0, d_wg1, d_wgt, d_lnz, d_anc;
    // Work of one run() is kept in two sets: tables whose eigen-system is rebuilt in this run (set 1: eigen ->
    // expm -> prune on the side stream) and all others (set 0: expm -> prune on the main stream), so that the
    // latency-bound eigen kernel of one half of the blocks overlaps the matrix work of the other half.
    std::vector<ExpmJob> jobs[2];
    std::vector<EigenJob> eig_jobs;
    std::vector<uint64_t> wg0[2], wg1[2];
    std::vector<uint32_t> wgt[2];
    std::vector<uint8_t> table_set;   // [table] -> 0 / 1 for the current run()
    hipStream_t side = nullptr;
    hipEvent_t ev_start = nullptr, ev_side = nullptr;
    std::vector<double> lnz, anc;
    std::vector<uint32_t> fail;
    size_t cap_tables = 0, cap_eig = 0;

    BlockEngine(pcsf_ctx* ctx, const uint8_t* codons, const uint64_t* offsets, size_t nb, size_t stride)
        : c(ctx), n_blocks(nb), leaf_stride(stride), off(offsets), stream(ctx->stream)
    {
        ne = static_cast<int>(c->schedule.slot_node.size());
        nl = c->n_leaves;
        n = c->n_nodes;
        n_sites_total = offsets[nb];
        HIP_TRY(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&ev_start, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&ev_side, hipEventDisableTiming));
        if (n_sites_total > 0) {
            HIP_TRY(d_codons.reserve(static_cast<size_t>(nl) * n_sites_total));
            HIP_TRY(hipMemcpy2DAsync(d_codons.ptr, n_sites_total, codons, stride, n_sites_total, nl, hipMemcpyHostToDevice, stream));
        }
    }

    ~BlockEngine()
    {
        if (side) {
            (void)hipStreamSynchronize(side);
            (void)hipStreamDestroy(side);
        }
        if (ev_start) (void)hipEventDestroy(ev_start);
        if (ev_side) (void)hipEventDestroy(ev_side);
    }
    BlockEngine(const BlockEngine&) = delete;
    BlockEngine& operator=(const BlockEngine&) = delete;

    void reserve_tables(size_t tables)
    {
        if (table_set.size() < tables) table_set.resize(tables, 0);
        if (tables <= cap_tables) return;
        HIP_TRY(d_frag.reserve(std::max<size_t>(1, tables * ne) * 4096 * sizeof(double)));
        HIP_TRY(d_leaf.reserve(tables * nl * LEAF_TABLE_DOUBLES_DEV * sizeof(double)));
        HIP_TRY(d_prior.reserve(tables * 64 * sizeof(double)));
        HIP_TRY(d_ancw.reserve(tables * 64 * sizeof(double)));
        HIP_TRY(d_fail.reserve(tables * sizeof(uint32_t)));
        cap_tables = tables;
    }
    void reserve_eig(size_t count)
    {
        if (count <= cap_eig) return;
        HIP_TRY(d_eig_r.reserve(count * 4096 * sizeof(double)));
        HIP_TRY(d_eig_l.reserve(count * 4096 * sizeof(double)));
        HIP_TRY(d_eig_lam.reserve(count * 64 * sizeof(double)));
        cap_eig = count;
    }
    void upload_eig(size_t idx, const EigenSystem& es)
    {
        HIP_TRY(hipMemcpyAsync(d_eig_r.as<double>() + idx * 4096, es.right.data(), 4096 * sizeof(double), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(d_eig_l.as<double>() + idx * 4096, es.left.data(), 4096 * sizeof(double), hipMemcpyHostToDevice, stream));
        HIP_TRY(hipMemcpyAsync(d_eig_lam.as<double>() + idx * 64, es.lambda.data(), 64 * sizeof(double), hipMemcpyHostToDevice, stream));
    }
    // root prior (and ln(prior)*prior) of a table set: staged on the host, uploaded once per run()
    std::vector<double> h_prior, h_ancw;
    bool priors_dirty = false;
    void set_prior(size_t table, const Vec& prior)
    {
        if (h_prior.size() < cap_tables * 64) {
            h_prior.resize(cap_tables * 64, 0.0);
            h_ancw.resize(cap_tables * 64, 0.0);
        }
        Vec w;
        for (int x = 0; x < 64; ++x) w[x] = std::log(prior[x]) * prior[x];
        to_lane_order(prior.data(), h_prior.data() + table * 64);
        to_lane_order(w.data(), h_ancw.data() + table * 64);
        priors_dirty = true;
    }

    // queue the eigen-system `eig` (and root prior of table set `table`) of the kappa/omega rate matrix with
    // settings qs[12]; computed on the device before the branch matrices of the same run()
    void add_eigen_job(uint32_t eig, uint32_t table, const double* qs)
    {
        EigenJob j;
        j.eig = eig;
        j.table = table;
        for (int i = 0; i < 12; ++i) j.qs[i] = qs[i];
        eig_jobs.push_back(j);
        table_set[table] = 1;
    }
    // queue the n-1 branch matrices of table set `table` from eigen system `eig` at tree scale x
    void add_expm_jobs(uint32_t table, uint32_t eig, double x)
    {
        for (int b = 0; b < n - 1; ++b) {
            ExpmJob j;
            j.eig = eig;
            j.table = table;
            j.slot = c->branch_slot[b];
            j.leaf = b < nl ? b : -1;
            j.t = scaled_branch(c->branch_len[b], x);
            jobs[table_set[table]].push_back(j);
        }
    }
    // queue the work-groups that prune block `blk` with table set `table`
    void add_prune(size_t blk, uint32_t table)
    {
        const int set = table_set[table];
        for (uint64_t s = off[blk]; s < off[blk + 1]; s += 64) {
            wg0[set].push_back(s);
            wg1[set].push_back(std::min<uint64_t>(s + 64, off[blk + 1]));
            wgt[set].push_back(table);
        }
    }

    // runs the queued expm jobs and prune work-groups (one pass per work-group); afterwards
    // lnz[site] (and anc[site] if wanted) hold the per-site values and fail[table] the validation flags
    void run(size_t tables, bool want_anc)
    {
        HIP_TRY(hipMemsetAsync(d_fail.ptr, 0, tables * sizeof(uint32_t), stream));
        if (priors_dirty) {
            HIP_TRY(hipMemcpyAsync(d_prior.ptr, h_prior.data(), h_prior.size() * sizeof(double), hipMemcpyHostToDevice, stream));
            HIP_TRY(hipMemcpyAsync(d_ancw.ptr, h_ancw.data(), h_ancw.size() * sizeof(double), hipMemcpyHostToDevice, stream));
            priors_dirty = false;
        }
        // uploads (main stream), then the two sets on their streams
        const size_t nj0 = jobs[0].size(), nj1 = jobs[1].size(), nw0 = wg0[0].size(), nw1 = wg0[1].size();
        if (!eig_jobs.empty()) {
            HIP_TRY(d_eig_jobs.reserve(eig_jobs.size() * sizeof(EigenJob)));
            HIP_TRY(hipMemcpyAsync(d_eig_jobs.ptr, eig_jobs.data(), eig_jobs.size() * sizeof(EigenJob), hipMemcpyHostToDevice, stream));
        }
        if (nj0 + nj1) {
            HIP_TRY(d_jobs.reserve((nj0 + nj1) * sizeof(ExpmJob)));
            if (nj0) HIP_TRY(hipMemcpyAsync(d_jobs.as<ExpmJob>(), jobs[0].data(), nj0 * sizeof(ExpmJob), hipMemcpyHostToDevice, stream));
            if (nj1) HIP_TRY(hipMemcpyAsync(d_jobs.as<ExpmJob>() + nj0, jobs[1].data(), nj1 * sizeof(ExpmJob), hipMemcpyHostToDevice, stream));
        }
        if (nw0 + nw1) {
            const size_t nw = nw0 + nw1;
            HIP_TRY(d_wg0.reserve(nw * 8));
            HIP_TRY(d_wg1.reserve(nw * 8));
            HIP_TRY(d_wgt.reserve(nw * 4));
            HIP_TRY(d_lnz.reserve(n_sites_total * 8));
            HIP_TRY(d_anc.reserve(n_sites_total * 8));
            for (int set = 0; set < 2; ++set) {
                const size_t at = set ? nw0 : 0, cnt = wg0[set].size();
                if (!cnt) continue;
                HIP_TRY(hipMemcpyAsync(d_wg0.as<uint64_t>() + at, wg0[set].data(), cnt * 8, hipMemcpyHostToDevice, stream));
                HIP_TRY(hipMemcpyAsync(d_wg1.as<uint64_t>() + at, wg1[set].data(), cnt * 8, hipMemcpyHostToDevice, stream));
                HIP_TRY(hipMemcpyAsync(d_wgt.as<uint32_t>() + at, wgt[set].data(), cnt * 4, hipMemcpyHostToDevice, stream));
            }
        }
        const bool use_side = !eig_jobs.empty() || nj1 || nw1;
        if (use_side) {
            HIP_TRY(hipEventRecord(ev_start, stream));
            HIP_TRY(hipStreamWaitEvent(side, ev_start, 0));
        }
        auto expm_and_prune = [&](int set, hipStream_t st) {
            const size_t nj = set ? nj1 : nj0, j_at = set ? nj0 : 0, nw = set ? nw1 : nw0, w_at = set ? nw0 : 0;
            if (nj) {
                ExpmParams ep{};
                ep.jobs = d_jobs.as<ExpmJob>() + j_at;
                ep.n_jobs = static_cast<uint32_t>(nj);
                ep.n_edges = ne;
                ep.n_leaves = nl;
                ep.eig_right = d_eig_r.as<double>();
                ep.eig_left = d_eig_l.as<double>();
                ep.eig_lambda = d_eig_lam.as<double>();
                ep.frag = d_frag.as<double>();
                ep.leaf_tab = d_leaf.as<double>();
                ep.fail = d_fail.as<uint32_t>();
                HIP_TRY(launch_expm(ep, st));
            }
            if (nw) {
                PruneParams p = base_prune_params(c);
                p.codons = d_codons.as<uint8_t>();
                p.n_sites = n_sites_total;
                p.leaf_stride = n_sites_total;
                p.n_passes = 1;
                p.frag = d_frag.as<double>();
                p.leaf_tab = d_leaf.as<double>();
                p.prior = d_prior.as<double>();
                p.anc_weight = d_ancw.as<double>();
                p.wg_site0 = d_wg0.as<uint64_t>() + w_at;
                p.wg_site_end = d_wg1.as<uint64_t>() + w_at;
                p.wg_table = d_wgt.as<uint32_t>() + w_at;
                p.n_wgs = nw;
                p.out_lnz = d_lnz.as<double>();
                p.out_anc = want_anc ? d_anc.as<double>() : nullptr;
                HIP_TRY(launch_prune_fixed(p, 16, st, 0));
            }
        };
        if (use_side) {
            if (!eig_jobs.empty()) {
                EigenParams gp{};
                gp.jobs = d_eig_jobs.as<EigenJob>();
                gp.n_jobs = static_cast<uint32_t>(eig_jobs.size());
                gp.eig_right = d_eig_r.as<double>();
                gp.eig_left = d_eig_l.as<double>();
                gp.eig_lambda = d_eig_lam.as<double>();
                gp.prior = d_prior.as<double>();
                gp.anc_weight = d_ancw.as<double>();
                gp.fail = d_fail.as<uint32_t>();
                HIP_TRY(launch_eigen_omega(gp, side));
            }
            expm_and_prune(1, side);
            HIP_TRY(hipEventRecord(ev_side, side));
        }
        expm_and_prune(0, stream);
        if (use_side) HIP_TRY(hipStreamWaitEvent(stream, ev_side, 0));
        if (nw0 + nw1) {
            lnz.resize(n_sites_total);
            HIP_TRY(hipMemcpyAsync(lnz.data(), d_lnz.ptr, n_sites_total * 8, hipMemcpyDeviceToHost, stream));
            if (want_anc) {
                anc.resize(n_sites_total);
                HIP_TRY(hipMemcpyAsync(anc.data(), d_anc.ptr, n_sites_total * 8, hipMemcpyDeviceToHost, stream));
            }
        }
        fail.resize(tables);
        HIP_TRY(hipMemcpyAsync(fail.data(), d_fail.ptr, tables * sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        HIP_TRY(hipStreamSynchronize(stream));
        for (int set = 0; set < 2; ++set) {
            jobs[set].clear();
            wg0[set].clear();
            wg1[set].clear();
            wgt[set].clear();
        }
        for (const EigenJob& j : eig_jobs) table_set[j.table] = 0;
        eig_jobs.clear();
    }

    double block_sum(const std::vector<double>& v, size_t blk) const
    {
        double s = 0.0;  // in site order, like lpr += log_z (src/fixed_lik.hpp:431-432)
        for (uint64_t i = off[blk]; i < off[blk + 1]; ++i) s += v[i];
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
        HIP_TRY(hipMemcpy2DAsync(c->d_codons.ptr, n_sites, codons, leaf_stride, n_sites, nl, hipMemcpyHostToDevice, c->stream));
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
            uint32_t seed, float* out_phylo, float* out_anc)
{
    BlockEngine eng(c, codons, off, n_blocks, leaf_stride);
    eng.reserve_tables(n_blocks);
    eng.reserve_eig(2);
    eng.upload_eig(0, c->eig[0]);
    eng.upload_eig(1, c->eig[1]);
    std::vector<std::mt19937> gen(n_blocks);
    std::vector<MaxLik1D> opt(n_blocks);
    std::vector<double> res_lpr(2 * n_blocks, 0.0), res_anc(2 * n_blocks, 0.0);
    std::vector<char> failed(n_blocks, 0);
    for (size_t b = 0; b < n_blocks; ++b) gen[b].seed(seed);

    for (int model = 0; model < 2; ++model) {  // coding first, then non-coding (src/run.hpp:193-194)
        for (size_t b = 0; b < n_blocks; ++b) eng.set_prior(b, c->eig[model].prior);
        for (size_t b = 0; b < n_blocks; ++b)
            if (!failed[b]) opt[b].start(1.0, 1e-2, 10.0, &gen[b]);
        for (;;) {
            size_t active = 0;
            for (size_t b = 0; b < n_blocks; ++b) {
                if (failed[b] || opt[b].done()) continue;
                ++active;
                if (off[b + 1] > off[b]) {  // blocks without a codon have lpr = 0 at every scale
                    eng.add_expm_jobs(static_cast<uint32_t>(b), static_cast<uint32_t>(model), opt[b].request());
                    eng.add_prune(b, static_cast<uint32_t>(b));
                }
            }
            if (!active) break;
            eng.run(n_blocks, compute_anc != 0);
            for (size_t b = 0; b < n_blocks; ++b) {
                if (failed[b] || opt[b].done()) continue;
                if (off[b + 1] > off[b] && eng.fail[b]) {
                    failed[b] = 1;  // std::runtime_error in PhyloModel_make -> NaN row
                    continue;
                }
                const bool has = off[b + 1] > off[b];
                opt[b].supply(has ? eng.block_sum(eng.lnz, b) : 0.0, (has && compute_anc) ? eng.block_sum(eng.anc, b) : 0.0);
            }
        }
        for (size_t b = 0; b < n_blocks; ++b) {
            res_lpr[model * n_blocks + b] = opt[b].last_lpr();
            res_anc[model * n_blocks + b] = opt[b].last_anc();
        }
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
struct OmegaBlock {
    double q_settings[12];
    double tree_scale = 1.0;
    std::mt19937 gen;
    MaxLik1D opt;
    int stage = 0;  // 0..5: H0 (rho,kappa)x3 ; 6..11: H1 (rho,kappa)x3 ; 12 = finished
    double lpr_h0 = 0.0, lpr_h1 = 0.0;
    bool failed = false, done = false;
};

int run_omega(pcsf_ctx* c, const uint8_t* codons, const uint64_t* off, size_t n_blocks, size_t leaf_stride, uint32_t seed,
              float* out_phylo, float* out_anc)
{
    BlockEngine eng(c, codons, off, n_blocks, leaf_stride);
    eng.reserve_tables(n_blocks);
    eng.reserve_eig(n_blocks);
    std::vector<std::unique_ptr<OmegaBlock>> blk(n_blocks);
    const long trace_block = std::getenv("PCSF_TRACE_BLOCK") ? std::atol(std::getenv("PCSF_TRACE_BLOCK")) : -1;  // diagnostics
    const size_t nl = static_cast<size_t>(c->n_leaves);

    // Rate matrix -> eigen-system -> root prior are (re)built on the device (eigen.hip) whenever a block's
    // q_settings change; block b owns eigen-system b and table set b.
    // validation-only pass (PhyloModel_make can throw): build every listed block's eigen-system and matrices
    // and read the flags
    auto validate_all = [&](const std::vector<size_t>& which) {
        for (size_t b : which) {
            eng.add_eigen_job(static_cast<uint32_t>(b), static_cast<uint32_t>(b), blk[b]->q_settings);
            eng.add_expm_jobs(static_cast<uint32_t>(b), static_cast<uint32_t>(b), blk[b]->tree_scale);
        }
        eng.run(n_blocks, false);
        for (size_t b : which)
            if (eng.fail[b]) blk[b]->failed = true;
    };

    std::vector<size_t> live;
    for (size_t b = 0; b < n_blocks; ++b) {
        blk[b].reset(new OmegaBlock);
        OmegaBlock& ob = *blk[b];
        ob.gen.seed(seed);
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
    validate_all(live);
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
    {
        std::vector<size_t> ok;
        for (size_t b : live)
            if (!blk[b]->failed) ok.push_back(b);
        validate_all(ok);
    }

    auto start_stage = [&](OmegaBlock& ob) {
        const bool rho = (ob.stage % 2) == 0;
        if (rho) ob.opt.start(ob.tree_scale, 0.001, 10.0, &ob.gen);
        else ob.opt.start(ob.q_settings[0], 1.0, 10.0, &ob.gen);
    };
    for (size_t b : live)
        if (!blk[b]->failed) start_stage(*blk[b]);

    const bool timing = std::getenv("PCSF_TIME_BLOCKS") != nullptr;  // diagnostics: where the rounds spend their time
    double t_eig = 0.0, t_queue = 0.0, t_run = 0.0, t_opt = 0.0;
    size_t rounds = 0, evals = 0, eig_evals = 0;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double>(b - a).count(); };
    for (;;) {
        std::vector<size_t> act;
        for (size_t b : live)
            if (!blk[b]->failed && !blk[b]->done) act.push_back(b);
        if (act.empty()) break;
        const auto t0 = now();
        ++rounds;
        evals += act.size();
        for (size_t b : act) eig_evals += (blk[b]->stage % 2) == 1;
        // apply each block's request: rho -> tree scale; kappa -> new rate matrix + eigen-system (device)
        const auto t1 = now();
        for (size_t b : act) {
            OmegaBlock& ob = *blk[b];
            if (ob.failed) continue;
            const double x = ob.opt.request();
            if ((ob.stage % 2) == 0) ob.tree_scale = x;
            else {
                ob.q_settings[0] = x;
                eng.add_eigen_job(static_cast<uint32_t>(b), static_cast<uint32_t>(b), ob.q_settings);
            }
            eng.add_expm_jobs(static_cast<uint32_t>(b), static_cast<uint32_t>(b), ob.tree_scale);
            eng.add_prune(b, static_cast<uint32_t>(b));
        }
        const auto t2 = now();
        eng.run(n_blocks, false);
        const auto t3 = now();
        std::vector<size_t> entering_h1;
        for (size_t b : act) {
            OmegaBlock& ob = *blk[b];
            if (ob.failed) continue;
            if (eng.fail[b]) {
                ob.failed = true;
                continue;
            }
            const double x = ob.opt.request();
            double lpr = eng.block_sum(eng.lnz, b);
            lpr += ((ob.stage % 2) == 0) ? log_prior_rho(x) : log_prior_kappa(x);
            if (trace_block >= 0 && static_cast<size_t>(trace_block) == b)
                fprintf(stderr, "TRACE omega stage %d x %.17g lpr %.17g\n", ob.stage, x, lpr);
            ob.opt.supply(lpr, 0.0);
            if (!ob.opt.done()) continue;
            // max_lik_lpr_leaves returned: the instance stays at the last evaluated point
            (ob.stage < 6 ? ob.lpr_h0 : ob.lpr_h1) = ob.opt.last_lpr();
            ++ob.stage;
            if (ob.stage == 6) entering_h1.push_back(b);
            else if (ob.stage == 12) ob.done = true;
            else start_stage(ob);
        }
        if (!entering_h1.empty()) {  // omega_H1 = 0.2, sigma_H1 = 0.01, keep rho and kappa (src/run.hpp:161-166)
            for (size_t b : entering_h1) {
                blk[b]->q_settings[1] = 0.2;
                blk[b]->q_settings[2] = 0.01;
            }
            std::vector<size_t> ok;
            for (size_t b : entering_h1)
                if (!blk[b]->failed) ok.push_back(b);
            validate_all(ok);
            for (size_t b : ok)
                if (!blk[b]->failed) start_stage(*blk[b]);
        }
        const auto t4 = now();
        t_eig += secs(t0, t1);
        t_queue += secs(t1, t2);
        t_run += secs(t2, t3);
        t_opt += secs(t3, t4);
    }
    if (timing)
        fprintf(stderr, "run_omega: %zu blocks, %zu rounds, %zu evaluations (%zu with a new eigen-system): bookkeeping %.3f s, queue %.3f s, device rounds %.3f s, optimiser %.3f s\n",
                n_blocks, rounds, evals, eig_evals, t_eig, t_queue, t_run, t_opt);
    for (size_t b = 0; b < n_blocks; ++b) {
        const OmegaBlock& ob = *blk[b];
        out_phylo[b] = ob.failed ? NAN : decibans(ob.lpr_h1, ob.lpr_h0);
        if (out_anc) out_anc[b] = NAN;  // src/run.hpp:181
    }
    return PCSF_OK;
}

}  // namespace

// one objective evaluation per block at caller-chosen parameters (no prior term)
int block_loglik_impl(pcsf_ctx* c, int model, const double* q_settings, const double* tree_scale, const uint8_t* codons,
                      const uint64_t* site_offsets, size_t n_blocks, size_t leaf_stride, double* out_lpr, int32_t* out_failed)
{
    std::vector<uint64_t> off(n_blocks + 1);
    for (size_t b = 0; b <= n_blocks; ++b) off[b] = site_offsets[b] - site_offsets[0];
    BlockEngine eng(c, codons ? codons + site_offsets[0] : nullptr, off.data(), n_blocks, leaf_stride);
    eng.reserve_tables(n_blocks);
    std::vector<char> bad(n_blocks, 0);
    if (model == 2) {
        eng.reserve_eig(n_blocks);  // eigen-systems and root priors come from the device solver (eigen.hip)
    } else {
        eng.reserve_eig(1);
        eng.upload_eig(0, c->eig[model]);
    }
    for (size_t b = 0; b < n_blocks; ++b) {
        if (model == 2) eng.add_eigen_job(static_cast<uint32_t>(b), static_cast<uint32_t>(b), q_settings + 12 * b);
        else eng.set_prior(b, c->eig[model].prior);
        eng.add_expm_jobs(static_cast<uint32_t>(b), model == 2 ? static_cast<uint32_t>(b) : 0u, tree_scale[b]);
        eng.add_prune(b, static_cast<uint32_t>(b));
    }
    eng.run(n_blocks, false);
    for (size_t b = 0; b < n_blocks; ++b) {
        const bool f = bad[b] || eng.fail[b];
        if (out_failed) out_failed[b] = f;
        out_lpr[b] = f ? NAN : (off[b + 1] > off[b] ? eng.block_sum(eng.lnz, b) : 0.0);
    }
    return PCSF_OK;
}

// diagnostics/tests: eigen-systems of n kappa/omega rate matrices from the device solver
int debug_eigen_omega_impl(pcsf_ctx* c, const double* q_settings, size_t n, double* right, double* left, double* lambda, double* prior,
                           int32_t* failed)
{
    DeviceBuffer d_jobs, d_r, d_l, d_lam, d_prior, d_fail;
    std::vector<EigenJob> jobs(n);
    for (size_t i = 0; i < n; ++i) {
        jobs[i].eig = jobs[i].table = static_cast<uint32_t>(i);
        for (int k = 0; k < 12; ++k) jobs[i].qs[k] = q_settings[12 * i + k];
    }
    HIP_TRY(d_jobs.reserve(n * sizeof(EigenJob)));
    HIP_TRY(d_r.reserve(n * 4096 * 8));
    HIP_TRY(d_l.reserve(n * 4096 * 8));
    HIP_TRY(d_lam.reserve(n * 64 * 8));
    HIP_TRY(d_prior.reserve(n * 64 * 8));
    HIP_TRY(d_fail.reserve(n * 4));
    HIP_TRY(hipMemsetAsync(d_fail.ptr, 0, n * 4, c->stream));
    HIP_TRY(hipMemcpyAsync(d_jobs.ptr, jobs.data(), n * sizeof(EigenJob), hipMemcpyHostToDevice, c->stream));
    EigenParams gp{};
    gp.jobs = d_jobs.as<EigenJob>();
    gp.n_jobs = static_cast<uint32_t>(n);
    gp.eig_right = d_r.as<double>();
    gp.eig_left = d_l.as<double>();
    gp.eig_lambda = d_lam.as<double>();
    gp.prior_plain = d_prior.as<double>();  // state order; the lane-order copies are not wanted here
    gp.fail = d_fail.as<uint32_t>();
    HIP_TRY(launch_eigen_omega(gp, c->stream));
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
    // MLE / OMEGA keep one set of branch-matrix tables per block in HBM (9.6 MB for a 100-leaf tree): blocks are
    // independent, so a long list is processed in chunks that bound that footprint to ~32 GB
    size_t chunk = n_blocks;
    if (algo != PCSF_FIXED) {
        const size_t ne = c->schedule.slot_node.size();
        const size_t per_table = (ne * 4096 + static_cast<size_t>(c->n_leaves) * LEAF_TABLE_DOUBLES_DEV) * sizeof(double);
        chunk = std::max<size_t>(64, (size_t(32) << 30) / std::max<size_t>(per_table, 1));
    }
    for (size_t b0 = 0; b0 < n_blocks; b0 += chunk) {
        const size_t nb = std::min(chunk, n_blocks - b0);
        std::vector<uint64_t> off(nb + 1);  // CSR offsets re-based at the chunk's first site
        for (size_t b = 0; b <= nb; ++b) off[b] = site_offsets[b0 + b] - site_offsets[b0];
        const uint8_t* cod = codons ? codons + site_offsets[b0] : nullptr;
        float* ph = out_phylo + b0;
        float* an = out_anc ? out_anc + b0 : nullptr;
        int rc;
        switch (algo) {
            case PCSF_FIXED: rc = run_fixed(c, cod, off.data(), nb, leaf_stride, compute_anc, ph, an); break;
            case PCSF_MLE: rc = run_mle(c, cod, off.data(), nb, leaf_stride, compute_anc, seed, ph, an); break;
            case PCSF_OMEGA: rc = run_omega(c, cod, off.data(), nb, leaf_stride, seed, ph, an); break;
            default: return c->fail(PCSF_ERR_INVALID_ARG, "unknown algorithm");
        }
        if (rc != PCSF_OK) return rc;
    }
    return PCSF_OK;
}

}  // namespace pcsf
