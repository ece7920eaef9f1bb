// machine.hpp -- what ONE block does between two batches of objective evaluations in the MLE / OMEGA strategies
// (reference: run(), src/run.hpp:57-211 -- max_lik_lpr_leaves twice for MLE, src/run.hpp:184-208; the twelve chained searches over
// rho and kappa for OMEGA, src/run.hpp:136-179 with src/omega.hpp:204-234).
//
// A block's optimisation is a chain: [the block's machine decides which points it needs] -> [eigen-systems, edge scales, pruning for
// those points] -> [the machine takes the values, moves its searches on, decides again] ...  The machines are plain data plus
// `round()`; one source compiled twice:
//   * chain.hip `advance_blocks` runs round() for every block of a list ON THE DEVICE, one wave per block, and writes the job lists
//     the eigen / edge-scale / pruning kernels of the same round read -- the whole optimiser loop (restarts, Brent, the OMEGA
//     schedule with its held eigen-systems) runs without the host; the host only enqueues rounds and polls a counter.
//   * blocks.cpp's lock-step engine runs the same round() on the host (rounds 1-3's engine, kept as the A/B reference:
//     PCSF_BLOCK_ENGINE=host); both engines must produce the same bits (tests/test_gpu_blocks.py).
// A block's requests depend on the block alone -- never on how many other blocks share the call, on evaluation capacity or on launch
// packing -- so a block's result is independent of the list it is scored in (src/phylocsf++score_msa.hpp:115).
#pragma once

#include <cstdint>

#include "optimizer.hpp"

namespace pcsf {

// diagnostics of the device engine (-DPCSF_ADVANCE_STAMPS=1, chain.hip): where OmegaMachine::round spends one lane's cycles
#if defined(__HIPCC__) && defined(PCSF_ADVANCE_STAMPS) && PCSF_ADVANCE_STAMPS
__device__ unsigned long long g_machine_stamp[8];
#endif
#if defined(__HIP_DEVICE_COMPILE__) && defined(PCSF_ADVANCE_STAMPS) && PCSF_ADVANCE_STAMPS
#define PCSF_MACHINE_STAMP(i_)                                                            \
    do {                                                                                  \
        if (blockIdx.x == PCSF_ADVANCE_STAMP_BLOCK) {                                     \
            unsigned long long t_;                                                        \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");    \
            g_machine_stamp[i_] = t_;                                                     \
        }                                                                                 \
    } while (0)
#else
#define PCSF_MACHINE_STAMP(i_) do { } while (0)
#endif
constexpr int MAX_POINTS_PER_SEARCH = 12;     // per search and round: certain points first, then speculative restarts
constexpr int MAX_SPECULATIVE_RESTARTS = 8;
constexpr int MIN_POINTS_PER_SEARCH = 5;      // a Brent set-up asks for chosen, lo, hi, chosen, v: all certain, all in one round
// How far ahead a search looks.  Short lists are bound by the latency of a round: random restarts are also evaluated before it is known that
// the search needs any (lo, hi and the start not yet evaluated).  Long lists are bound by work: a third of OMEGA's evaluations on a
// 5,088-block list were such restarts that no optimiser ever asked for.  A search that IS restarting always has its next eight points
// evaluated with the one it waits for (a block that exhausts its 250 restarts would otherwise set the number of rounds of the whole list:
// 58 -> 170 rounds for MLE, 233 -> 569 for OMEGA on BASELINE.json configs[2] with two ahead).
struct LookaheadPolicy {
    int max_points, max_restarts;
    bool eager;
};
PCSF_HD inline LookaheadPolicy lookahead_policy(bool eager) { return LookaheadPolicy{MAX_POINTS_PER_SEARCH, MAX_SPECULATIVE_RESTARTS, eager}; }
constexpr int MLE_MAX_REQUESTS = 2 * MAX_POINTS_PER_SEARCH;   // the coding and the non-coding search advance in the same rounds
constexpr int OMEGA_MAX_REQUESTS = MAX_POINTS_PER_SEARCH;     // a block either validates its model or searches
constexpr uint32_t STORE_ENTRIES = 96;        // values a search keeps (store_compact): it can evaluate ~520 points -- lo, hi, start, <= 250 restarts, v, <= 250 Brent probes
constexpr uint32_t STORE_COMPACT_AT = 48;

// EvalRequest::eig of a probe that is not held (a random restart): TRANSIENT_EIG | k = the k-th such system of the block's round; the engine
// gives it a slot for the round
constexpr uint32_t TRANSIENT_EIG = 0x80000000u;

enum EvalKind : uint32_t { EVAL_MLE_CODING = 0, EVAL_MLE_NONCODING = 1, EVAL_RHO = 2, EVAL_KAPPA = 3, EVAL_VALIDATION = 4 };

// values a search has seen: the objective is a pure function of the point, so a point asked for again (the reference
// re-evaluates lo, hi and the chosen start when it sets up Brent) is served from here
struct StoreEntry {
    double x, lpr, anc;
    uint32_t failed, used;
};

struct EvalRequest {
    double x;            // the point the search asked for
    double tree_scale;   // the tree scale the evaluation's branches are scaled with
    double qs[12];       // new_eigen: kappa, omega, sigma, nine ratios of the rate matrix to build (src/run.hpp:75-79)
    uint32_t eig;        // eigen-system slot the evaluation uses
    int32_t warm;        // new_eigen: slot whose eigenbasis the Jacobi iteration starts from, or -1
    uint32_t kind;       // EvalKind
    uint32_t new_eigen;  // slot `eig` is (re)built first; 2: it is rebuilt by ANOTHER request of this round (run behind it)
};

struct EvalResult {
    double lnz, anc;     // sums over the block's sites in site order, like lpr += log_z (src/fixed_lik.hpp:431-432)
    uint32_t failed;     // PhyloModel_make would have thrown (src/instance.hpp:618,635)
};

// index of the value at x, or -1 (an index, not a pointer: on the device the store may sit in LDS, and null tests of generic pointers
// into LDS trip the gfx950 back end of ROCm 7.2 -- "Illegal instruction detected: V_CMP_NE_U32_e32 0, $src_shared_base")
// (No early exit: on the device the store sits in LDS and ONE lane runs the machine -- a loop that tests each entry before it loads the next
// costs an LDS round trip per entry, ~4,000 cycles for a store of forty, and a round that looks twelve points ahead searches twelve times.
// Scanned from the back with a select, the loads are independent of the compares and the first match still wins.)
PCSF_HD inline int store_find(const StoreEntry* e, uint32_t n, double x)
{
    int found = -1;
    for (uint32_t i = n; i-- > 0;) found = (e[i].x == x) ? static_cast<int>(i) : found;
    return found;
}
PCSF_HD inline void store_push(StoreEntry* e, uint32_t& n, double x, double lpr, double anc, bool failed)
{
    if (n >= STORE_ENTRIES) return;   // cannot happen (STORE_ENTRIES); a lost value would only be evaluated again
    e[n].x = x;
    e[n].lpr = lpr;
    e[n].anc = anc;
    e[n].failed = failed ? 1u : 0u;
    e[n].used = 0u;
    ++n;
}

// A search that exhausts its random restarts (or a Brent iteration that runs long) would fill hundreds of entries.  Of the values it has
// CONSUMED it can only ask again for lo, hi and the start (entries 0..2: Brent's set-up re-evaluates the ends and the chosen start) and
// for the restart point it consumed LAST (the only one that can become the chosen start, src/fixed_lik.hpp:494-512): the others go.
// Values evaluated ahead and not yet consumed stay.  Bounds a store by ~40 entries, so that the device keeps it in LDS.
PCSF_HD inline void store_compact(StoreEntry* e, uint32_t& n, uint32_t& tally_used)
{
    if (n <= STORE_COMPACT_AT) return;
    int last_used = -1;
    for (uint32_t i = 0; i < n; ++i)
        if (e[i].used) last_used = static_cast<int>(i);
    uint32_t k = 0;
    for (uint32_t i = 0; i < n; ++i) {
        if (i < 3 || !e[i].used || static_cast<int>(i) == last_used) {
            if (k != i) e[k] = e[i];
            ++k;
        } else {
            ++tally_used;
        }
    }
    n = k;
}

struct Lookahead {  // MaxLik1D::upcoming() of one search for the current round
    int n;   // (set by whoever fills the list: the struct sits in LDS on the device, which takes no initialisers)
    double x[MAX_POINTS_PER_SEARCH];
    PCSF_HD bool seen_before(int rank) const  // the same point earlier in the list (lo, hi, chosen recur when Brent is set up)
    {
        int seen = 0;   // (no early exit, like store_find)
        const double xr = x[rank];
        for (int i = 0; i < MAX_POINTS_PER_SEARCH; ++i) seen |= static_cast<int>(i < rank) & static_cast<int>(x[i] == xr);
        return seen != 0;
    }
};

// ---- MLE ---------------------------------------------------------------------------------------------------------------------
// The reference optimises the coding model, then the non-coding one (src/run.hpp:193-194), drawing the random restarts of both from one
// generator per block.  The two searches are independent otherwise, and a search draws only BEFORE its Brent iteration
// (MaxLik1D::draws_finished): the non-coding search of a block starts as soon as the coding one has drawn its last number, sees the
// same numbers as it would afterwards, and the two advance in the same rounds -- half the rounds, the same results.
struct MleMachine {
    MaxLik1D opt[2];
    uint32_t n_store[2];
    uint32_t draws;        // numbers this block's generator has produced
    uint32_t n_sites;      // codon columns of the block (none: lpr = 0 at every scale)
    uint32_t failed, started1, active;
    uint32_t n_pending;
    uint32_t used, unused;  // diagnostics: stored values the searches consumed / speculative ones they never asked for
    double pend_x[MLE_MAX_REQUESTS];
    uint32_t pend_kind[MLE_MAX_REQUESTS];

    PCSF_HD void init(uint32_t sites)
    {
        opt[0] = MaxLik1D();
        opt[1] = MaxLik1D();
        n_store[0] = n_store[1] = 0;
        draws = 0;
        n_sites = sites;
        failed = started1 = 0;
        active = 1;
        n_pending = 0;
        used = unused = 0;
        opt[0].start(1.0, 1e-2, 10.0);
    }

    // res[i]: the value of pending request i (n_pending of them; none in the first round).  store: two stores of up to STORE_ENTRIES values,
    // store_stride entries apart.
    // max_points (5..MAX_POINTS_PER_SEARCH) / max_restarts: how far a search looks ahead per round -- the five certain points of a Brent set-up
    // always fit; no result depends on either.  Returns the number of requests written to out[2 * max_points]; 0 = the block is finished (or failed).
    PCSF_HD int round(const EvalResult* res, StoreEntry* store, uint32_t store_stride, const DrawTable& draws_table, bool eager, int max_points, int max_restarts, bool compute_anc,
                      EvalRequest* out, Lookahead* ahead /* [2], work space: on the device in LDS -- as a local it is indexed at run time and lives in scratch memory */)
    {
        for (int m = 0; m < 2; ++m) store_compact(store + m * store_stride, n_store[m], used);
        for (uint32_t i = 0; i < n_pending; ++i) {
            const uint32_t m = pend_kind[i];
            store_push(store + m * store_stride, n_store[m], pend_x[i], res[i].lnz, compute_anc ? res[i].anc : 0.0, res[i].failed != 0);
        }
        n_pending = 0;
        // feed every search from its store; blocks without a codon have lpr = 0 at every scale
        for (int model = 0; model < 2 && !failed; ++model) {
            if (model == 1 && !started1) {
                if (!opt[0].draws_finished()) break;
                opt[1].start(1.0, 1e-2, 10.0);
                started1 = 1;
            }
            MaxLik1D& o = opt[model];
            while (!o.done()) {
                if (n_sites == 0) {
                    o.supply(0.0, 0.0, draws_table, draws);
                    continue;
                }
                StoreEntry* st = store + model * store_stride;
                const int at = store_find(st, n_store[model], o.request());
                if (at < 0) break;
                if (st[at].failed) {  // std::runtime_error in PhyloModel_make -> NaN row
                    failed = 1;
                    break;
                }
                st[at].used = 1;
                o.supply(st[at].lpr, st[at].anc, draws_table, draws);
            }
        }
        active = (!failed && (!opt[0].done() || !started1 || !opt[1].done())) ? 1u : 0u;
        if (!active) {
            for (int m = 0; m < 2; ++m)
                for (uint32_t i = 0; i < n_store[m]; ++i) (store[m * store_stride + i].used ? used : unused) += 1;
            return 0;
        }
        // the points the searches will or may need next: rank 0 (certain) of both first
        ahead[0].n = ahead[1].n = 0;
        for (int i = 0; i < MAX_POINTS_PER_SEARCH; ++i) ahead[0].x[i] = ahead[1].x[i] = 0.0;   // (seen_before scans the whole list)
        for (int model = 0; model < 2; ++model)
            if ((model == 0 || started1) && !opt[model].done())
                ahead[model].n = opt[model].upcoming(ahead[model].x, max_points, max_restarts, eager, nullptr, draws_table, draws);
        int n = 0;
        for (int rank = 0; rank < MAX_POINTS_PER_SEARCH; ++rank)
            for (int model = 0; model < 2; ++model) {
                if (rank >= ahead[model].n) continue;
                const double x = ahead[model].x[rank];
                if (store_find(store + model * store_stride, n_store[model], x) >= 0 || ahead[model].seen_before(rank)) continue;
                EvalRequest& r = out[n];
                r.x = r.tree_scale = x;
                r.eig = static_cast<uint32_t>(model);
                r.warm = -1;
                r.kind = static_cast<uint32_t>(model);
                r.new_eigen = 0;
                pend_x[n] = x;
                pend_kind[n] = static_cast<uint32_t>(model);
                ++n;
            }
        n_pending = static_cast<uint32_t>(n);
        return n;
    }
};

// ---- OMEGA -------------------------------------------------------------------------------------------------------------------
// Eigen-systems a block keeps between rounds, by role.  OWN: the model the block is at.  LO / HI: kappa = 1 and kappa = 10, the ends
// every kappa stage evaluates first -- the three kappa stages of a hypothesis ask for the SAME two rate matrices.  P0 / P1: the last two
// interior kappa probes (Brent's steps shrink geometrically: the nearest earlier probe is a far better start for the Jacobi iteration
// than the stage's first point).  A probe whose settings equal a held system is not solved again (a fifth of all solves on the
// configs[2] list), every other probe starts from the nearest held system of its family (same omega, sigma, frequencies).  Roles map
// to physical slots through `slot`, so that the system of a stage's last probe BECOMES the block's own by a swap when the stage ends.
enum EigRole { ROLE_OWN = 0, ROLE_LO = 1, ROLE_HI = 2, ROLE_P0 = 3, ROLE_P1 = 4, N_ROLES = 5 };

PCSF_HD inline bool same_settings(const double* a, const double* b, int from = 0)
{
    // (no early exit: like store_find -- on the device both sides sit in LDS and one lane compares; a loop that tests each pair before it
    // loads the next pays a round trip per setting, and a kappa stage's round compares eleven probes with five held systems)
    int same = 1;
    for (int i = 0; i < 12; ++i) same &= static_cast<int>(i < from) | static_cast<int>(a[i] == b[i]);   // (settings are never NaN)
    return same != 0;
}
PCSF_HD inline void copy_settings(double* dst, const double* src)
{
    for (int i = 0; i < 12; ++i) dst[i] = src[i];
}

struct OmegaMachine {
    double q_settings[12];      // kappa, omega, sigma, nine ratios: the model the block is AT (src/run.hpp:75-79)
    double eig_settings[12];    // settings the block's own eigen-system slot currently holds
    double held[N_ROLES][12];   // settings the role's slot holds (OWN: eig_settings / eig_valid)
    double f3x4[9];             // the block's nucleotide-frequency ratios (src/run.hpp:105-134), set once the first validation has passed
    double tree_scale;
    double lpr_h0, lpr_h1;
    MaxLik1D opt;
    uint32_t slot[N_ROLES];     // physical eigen-system slot of every role
    uint32_t held_valid[N_ROLES];
    uint32_t eig_valid;
    uint32_t next_p;            // which of P0 / P1 the next interior probe overwrites
    uint32_t n_store, draws;
    int32_t stage;              // 0..5: H0 (rho,kappa)x3 ; 6..11: H1 (rho,kappa)x3 ; 12 = finished
    uint32_t prelude;           // 2: the initial model is being validated; 1: the model with the block's frequencies; 0: searching
    uint32_t failed, done, needs_validation, active;
    uint32_t n_pending;
    uint32_t n_eig_new, n_eig_reused, n_eig_adopted, used, unused;   // diagnostics
    double pend_x[OMEGA_MAX_REQUESTS];
    uint32_t pend_kind[OMEGA_MAX_REQUESTS];

    // own_slots[N_ROLES]: the block's eigen-system slots; ratios: its F3x4 ratios; sites == 0: NaN by policy ("OMEGA NaN rows":
    // the reference yields NaN for every block without a complete codon -- GSL's non-symmetric eigen-solver degrades on the uniform matrix)
    PCSF_HD void init(uint32_t sites, const uint32_t* own_slots, const double* ratios)
    {
        opt = MaxLik1D();
        for (int r = 0; r < N_ROLES; ++r) {
            slot[r] = own_slots[r];
            held_valid[r] = 0;
            for (int i = 0; i < 12; ++i) held[r][i] = 0.0;
        }
        for (int i = 0; i < 9; ++i) f3x4[i] = ratios[i];
        q_settings[0] = 2.5;  // kappa, omega, sigma, nine ratios (src/run.hpp:75-79)
        q_settings[1] = 1.0;
        q_settings[2] = 1.0;
        for (int i = 3; i < 12; ++i) q_settings[i] = 1.0;
        for (int i = 0; i < 12; ++i) eig_settings[i] = 0.0;
        tree_scale = 1.0;
        lpr_h0 = lpr_h1 = 0.0;
        eig_valid = 0;
        next_p = 0;
        n_store = draws = 0;
        stage = 0;
        prelude = 2;
        failed = sites == 0 ? 1u : 0u;
        done = 0;
        needs_validation = 1;
        active = failed ? 0u : 1u;
        n_pending = 0;
        n_eig_new = n_eig_reused = n_eig_adopted = used = unused = 0;
    }

    PCSF_HD void clear_store(const StoreEntry* store)
    {
        for (uint32_t i = 0; i < n_store; ++i) (store[i].used ? used : unused) += 1;
        n_store = 0;
    }
    PCSF_HD void start_stage(const StoreEntry* store)
    {
        clear_store(store);
        if ((stage % 2) == 0) opt.start(tree_scale, 0.001, 10.0);
        else opt.start(q_settings[0], 1.0, 10.0);
    }

    // res[i]: the likelihood part of pending request i; prior(i, kind, x): the log prior the objective adds to it (src/omega.hpp:130-149).
    // store: [STORE_ENTRIES]; rho_draws / kappa_draws: the restart points of the two search ranges.
    // Returns the number of requests written to out[OMEGA_MAX_REQUESTS]; 0 = the block is finished (or failed).
    template <class Prior>
    PCSF_HD int round(const EvalResult* res, StoreEntry* store, const DrawTable& rho_draws, const DrawTable& kappa_draws, bool eager, int max_points, int max_restarts,
                      bool held_systems, Prior prior, EvalRequest* out, Lookahead* ahead_buf /* work space, see MleMachine::round */)
    {
        PCSF_MACHINE_STAMP(0);
        store_compact(store, n_store, used);
        for (uint32_t i = 0; i < n_pending; ++i) {
            if (pend_kind[i] == EVAL_VALIDATION) {
                needs_validation = 0;
                if (res[i].failed) {
                    failed = 1;
                } else if (prelude == 2) {
                    // F3x4 nucleotide counts over every non-gap codon of every row (src/run.hpp:105-134): the model is instantiated again
                    for (int k = 0; k < 9; ++k) q_settings[3 + k] = f3x4[k];
                    prelude = 1;
                    needs_validation = 1;
                } else {
                    prelude = 0;
                    start_stage(store);
                }
            } else {
                store_push(store, n_store, pend_x[i], res[i].lnz + prior(i, pend_kind[i], pend_x[i]), 0.0, res[i].failed != 0);
            }
        }
        n_pending = 0;
        PCSF_MACHINE_STAMP(1);
        // (1) feed the search from the store; a finished stage moves the block on
        while (!failed && !done && !needs_validation) {
            const DrawTable& dt = (stage % 2) == 0 ? rho_draws : kappa_draws;
            if (opt.done()) {
                // max_lik_lpr_leaves returned: the model stays at the LAST evaluated point
                if ((stage % 2) == 0) {
                    tree_scale = opt.last_x();
                } else {
                    q_settings[0] = opt.last_x();
                    // the model stays at the stage's LAST probe -- whose eigen-system, if it was kept, becomes the block's own by a
                    // swap of roles instead of being solved once more at the next rho evaluation
                    for (int role = ROLE_LO; role < N_ROLES && held_systems; ++role)
                        if (held_valid[role] && same_settings(held[role], q_settings)) {
                            const uint32_t s = slot[ROLE_OWN];
                            slot[ROLE_OWN] = slot[role];
                            slot[role] = s;
                            double tmp[12];
                            copy_settings(tmp, eig_settings);
                            copy_settings(eig_settings, held[role]);
                            copy_settings(held[role], tmp);
                            const uint32_t v = eig_valid;
                            eig_valid = held_valid[role];
                            held_valid[role] = v;
                            ++n_eig_adopted;
                            break;
                        }
                }
                (stage < 6 ? lpr_h0 : lpr_h1) = opt.last_lpr();
                ++stage;
                if (stage == 12) {
                    done = 1;
                } else if (stage == 6) {  // omega_H1 = 0.2, sigma_H1 = 0.01, keep rho and kappa (src/run.hpp:161-166)
                    q_settings[1] = 0.2;
                    q_settings[2] = 0.01;
                    needs_validation = 1;   // PhyloModel_make can throw: validated by the round that follows
                } else {
                    start_stage(store);
                }
                continue;
            }
            const int at = store_find(store, n_store, opt.request());
            if (at < 0) break;
            if (store[at].failed) {
                failed = 1;
                break;
            }
            store[at].used = 1;
            opt.supply(store[at].lpr, 0.0, dt, draws);
        }
        active = (!failed && !done) ? 1u : 0u;
        if (!active) {
            clear_store(store);
            return 0;
        }
        // A block whose model was re-instantiated has it validated (one evaluation whose flags are read) before its next stage starts.
        if (needs_validation) {
            EvalRequest& r = out[0];
            r.x = r.tree_scale = tree_scale;
            copy_settings(r.qs, q_settings);
            r.eig = slot[ROLE_OWN];
            r.warm = eig_valid ? static_cast<int32_t>(slot[ROLE_OWN]) : -1;
            r.kind = EVAL_VALIDATION;
            r.new_eigen = 1;
            copy_settings(eig_settings, q_settings);
            eig_valid = 1;
            pend_x[0] = tree_scale;
            pend_kind[0] = EVAL_VALIDATION;
            n_pending = 1;
            return 1;
        }
        // (2) the points the search will or may need next
        PCSF_MACHINE_STAMP(2);
        Lookahead& ahead = *ahead_buf;
        ahead.n = 0;
        for (int i = 0; i < MAX_POINTS_PER_SEARCH; ++i) ahead.x[i] = 0.0;   // (seen_before scans the whole list)
        int ahead_certain = 0;
        const bool rho_stage = (stage % 2) == 0;
        ahead.n = opt.upcoming(ahead.x, max_points, max_restarts, eager, &ahead_certain, rho_stage ? rho_draws : kappa_draws, draws);
        PCSF_MACHINE_STAMP(3);
        bool family_known = false;     // kappa stages: family[] / kappa_of[] / qbase are set (see there)
        uint32_t family = 0u;          // bit role: the role's system is of the probes' family
        double k_own = 0.0, k_lo = 0.0, k_hi = 0.0, k_p0 = 0.0, k_p1 = 0.0, qbase[12];   // (scalars and selects: an array indexed by the role lives in scratch memory on the device)
        auto kappa_of = [&](int role) { return role == ROLE_OWN ? k_own : role == ROLE_LO ? k_lo : role == ROLE_HI ? k_hi : role == ROLE_P0 ? k_p0 : k_p1; };
        bool rebuilt = false;          // the block's own eigen-system slot is rebuilt in this round
        uint32_t written = 0;          // bit role: the role's slot is rewritten in this round
        uint32_t n_transient = 0;
        int n = 0;
        for (int rank = 0; rank < ahead.n; ++rank) {
            const double x = ahead.x[rank];
            if (store_find(store, n_store, x) >= 0 || ahead.seen_before(rank)) continue;
            EvalRequest& r = out[n];
            r.x = x;
            r.warm = -1;
            r.new_eigen = 0;
            if (rho_stage) {  // rho: the block's own eigen-system, brought up to date if need be
                r.kind = EVAL_RHO;
                r.tree_scale = x;
                r.eig = slot[ROLE_OWN];
                const bool stale = !eig_valid || !same_settings(eig_settings, q_settings);
                if (stale && !rebuilt) {
                    copy_settings(r.qs, q_settings);
                    r.warm = eig_valid ? static_cast<int32_t>(slot[ROLE_OWN]) : -1;
                    r.new_eigen = 1;
                    copy_settings(eig_settings, q_settings);
                    eig_valid = 1;
                    rebuilt = true;
                    ++n_eig_new;
                } else if (rebuilt) {
                    r.new_eigen = 2;
                }
            } else {  // kappa: a probe model
                r.kind = EVAL_KAPPA;
                r.tree_scale = tree_scale;
                // Every probe of the round is the block's model with another kappa, and a slot written in this round is out of the comparisons
                // for the rest of it: which held systems are of the probes' FAMILY (all settings but kappa equal), and their kappas, are the
                // same for every probe -- looked up once (five comparisons of twelve settings per round instead of ten per probe; on the
                // device one lane walks this against LDS latency, and a stage's first round lists eleven probes).
                if (!family_known) {
                    copy_settings(qbase, q_settings);
                    family = 0u;
                    for (int role = 0; role < N_ROLES; ++role) {
                        const double* hs = role == ROLE_OWN ? eig_settings : held[role];
                        family |= same_settings(hs, q_settings, 1) ? 1u << role : 0u;
                    }
                    k_own = eig_settings[0];
                    k_lo = held[ROLE_LO][0];
                    k_hi = held[ROLE_HI][0];
                    k_p0 = held[ROLE_P0][0];
                    k_p1 = held[ROLE_P1][0];
                    family_known = true;
                }
                double qs[12];
                copy_settings(qs, qbase);
                qs[0] = x;
                // Which probes may change what the block holds: the ends of the range, and the probes of the Brent phase (each asked
                // for only after its predecessor was consumed).  Random restarts never do, and always start from the block's own
                // system: whether a restart is evaluated ahead of need depends on the engine's speculation policy -- no result may.
                const bool certain = rank < ahead_certain;
                const bool brent_phase = opt.draws_finished();
                int hit = -1;
                for (int role = 0; role < N_ROLES && held_systems && hit < 0; ++role) {   // the role's slot holds exactly this model, and is not being rewritten in this round
                    const bool valid = role == ROLE_OWN ? (eig_valid && !rebuilt) : (held_valid[role] && !(written & (1u << role)));
                    if (valid && (family >> role & 1u) && kappa_of(role) == x) hit = role;   // = same_settings(the role's settings, qs)
                }
                if (hit >= 0) {   // solved before (the block's own model, or an end of the kappa range in an earlier stage): no new eigen-system
                    r.eig = slot[hit];
                    ++n_eig_reused;
                } else {
                    // where the new system goes: the ends of the range and the certain interior probes are kept, restarts are not
                    int target = -1;
                    if (held_systems && certain) {
                        if (x == 1.0) target = ROLE_LO;
                        else if (x == 10.0) target = ROLE_HI;
                        else if (brent_phase) {
                            target = ROLE_P0 + static_cast<int>(next_p);
                            next_p ^= 1u;
                        }
                    }
                    // Jacobi starts in the eigenbasis of the nearest held system of the same family (all settings but kappa equal).
                    // The ends only look at the block's own system and at what their slot held before.
                    int32_t warm = -1;
                    double best = 0.0;
                    for (int role = 0; role < N_ROLES; ++role) {
                        if ((!held_systems || target < 0) && role != ROLE_OWN) break;
                        if ((target == ROLE_LO || target == ROLE_HI) && role != ROLE_OWN && role != target) continue;
                        const bool valid = role == ROLE_OWN ? (eig_valid && !rebuilt) : (held_valid[role] && !(written & (1u << role)));
                        if (!valid || !(family >> role & 1u)) continue;   // = same_settings(the role's settings, qs, 1)
                        const double dist = fabs(kappa_of(role) - x);
                        if (warm < 0 || dist < best) {
                            warm = static_cast<int32_t>(slot[role]);
                            best = dist;
                        }
                    }
                    if (warm < 0 && eig_valid && !rebuilt) warm = static_cast<int32_t>(slot[ROLE_OWN]);   // another family: still better than a cold start
                    if (target >= 0 && !(written & (1u << target))) {
                        r.eig = slot[target];
                        copy_settings(held[target], qs);
                        held_valid[target] = 1;
                        written |= 1u << target;
                    } else {
                        r.eig = TRANSIENT_EIG | n_transient++;
                    }
                    copy_settings(r.qs, qs);
                    r.warm = warm;
                    r.new_eigen = 1;
                    ++n_eig_new;
                }
            }
            pend_x[n] = x;
            pend_kind[n] = r.kind;
            ++n;
        }
        PCSF_MACHINE_STAMP(4);
        n_pending = static_cast<uint32_t>(n);
        return n;
    }
};

}  // namespace pcsf
