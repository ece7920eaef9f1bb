// optimizer.hpp -- the 1-D maximiser of the MLE / OMEGA strategies as a resumable state machine, for the host AND the device.
//
// The reference runs, per block, `max_lik_lpr_leaves` (src/fixed_lik.hpp:469-544): a random-restart
// search for a starting point (fit_find_init) followed by GSL's Brent minimiser, every objective
// evaluation being a full likelihood pass.  On the GPU thousands of blocks advance in rounds:
// each block's optimiser is suspended whenever it needs an objective value, the requests of a round
// become one batch of eigen / edge-scale / pruning launches, and the optimisers resume with the results.
// Decision logic (comparison order, the result being the LAST evaluation, Brent's d/e swap on
// entry, tolerances) follows the reference and GSL's published min/brent.c so that the sequence
// of evaluation points is the same.
//
// One source, two compilations: machine.hpp's per-block machines (built on this class) run inside
// chain.hip's `advance_blocks` kernel -- the optimiser loop on the device, no host round trip per round --
// and, unchanged, on the host in blocks.cpp's lock-step engine (kept as the A/B reference: both must give the
// same bits).  Hence no std::mt19937 in here: see DrawTable.
#pragma once

#include <cfloat>
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define PCSF_HD __host__ __device__
#else
#define PCSF_HD
#endif

namespace pcsf {

// Random restarts.  The reference draws x = exp(log(lo) + U(0, log(hi) - log(lo))) from a std::mt19937 that is seeded once per
// block and shared by all of the block's searches (src/fixed_lik.hpp:483-491, src/run.hpp:150-179, seed 42 for every block:
// src/phylocsf++score_msa.hpp:115).  How MANY numbers a block draws depends on its data; WHICH number its k-th draw is does not --
// it is the k-th draw of the generator, the same for every block of a call.  So the generator runs once, on the host, with
// libstdc++'s own uniform_real_distribution and the host's exp / log (make_draw_table in blocks.cpp), and a block only keeps
// a count of its draws: x[k] is the restart point the k-th draw gives for this search range.  Exactly the numbers the host
// optimiser of rounds 1-3 drew, available to device code.
struct DrawTable {
    const double* x = nullptr;   // [n]
    uint32_t n = 0;
};
constexpr uint32_t DRAW_TABLE_ENTRIES = 12 * 250 + 16;   // OMEGA: twelve searches of at most 250 draws, plus the look-ahead

class MaxLik1D {
public:
    // objective value convention: `lpr` = the quantity being MAXIMISED (log-likelihood [+ log prior])
    PCSF_HD void start(double init, double lo, double hi)
    {
        init_ = init;
        lo_ = lo;
        hi_ = hi;
        tries_ = 0;
        state_ = S_LO;
        request_ = lo;
        done_ = false;
        evals_ = 0;
    }
    PCSF_HD bool done() const { return done_; }
    // true once the search will never draw again (the random restarts precede Brent's iteration): a second
    // search sharing the block's draw count may start then and still sees the numbers it would have seen after this one had finished
    PCSF_HD bool draws_finished() const { return done_ || state_ >= S_FINAL; }
    PCSF_HD double request() const { return request_; }  // x at which the objective is wanted next
    PCSF_HD double last_x() const { return last_x_; }
    PCSF_HD double last_lpr() const { return last_lpr_; }  // result of max_lik_lpr_leaves = LAST evaluation
    PCSF_HD double last_anc() const { return last_anc_; }
    PCSF_HD int evaluations() const { return evals_; }

    // Points whose objective value the search will or may ask for next, in the order it would ask, WITHOUT
    // depending on values not yet supplied: xs[0] == request() (certain); the rest is certain up to the next
    // data-dependent decision (lo, hi, init at the start; chosen, lo, hi, chosen, v when Brent is set up) or
    // speculative (the next random restarts: the restart sequence does not depend on the values, only its length
    // does).  A driver may evaluate all of them in one batch, keep the results and feed supply() from that store:
    // the sequence of supply() calls -- hence every decision and the result -- is the same as when evaluating one
    // point at a time.  Returns the number of points written.
    // eager_restarts: also list random restarts BEFORE it is known that the search needs any (lo, hi and the start not yet
    // evaluated).  Most searches need none -- the start beats both ends -- so those evaluations are usually thrown away: a third
    // of all OMEGA evaluations on a 5,088-block list (profiles/r03).  Worth it only while a round is latency-bound (few blocks).
    // n_certain (optional): how many of the points, from the front, the search WILL ask for whatever the values (the rest are
    // restarts that may never be consumed).  next_draw: the block's draw count (not advanced here).
    PCSF_HD int upcoming(double* xs, int max_points, int max_restarts, bool eager_restarts, int* n_certain, const DrawTable& draws, uint32_t next_draw) const
    {
        int n = 0, certain = -1;
        if (done_ || max_points <= 0) return 0;
#define PCSF_PUSH(x_)                       \
    do {                                    \
        if (n < max_points) xs[n++] = (x_); \
    } while (0)
#define PCSF_RESTARTS(already_)                                                                                    \
    do {                                                                                                           \
        certain = n;                                                                                               \
        if (draws.x && max_restarts > 0)                                                                           \
            for (int t = (already_), k = 0; t < 250 && k < max_restarts && n < max_points && next_draw + k < draws.n; ++t, ++k) \
                PCSF_PUSH(draws.x[next_draw + k]);                                                                 \
    } while (0)
        switch (state_) {
            case S_LO:
                PCSF_PUSH(lo_);
                PCSF_PUSH(hi_);
                PCSF_PUSH(init_);
                if (eager_restarts) PCSF_RESTARTS(0);
                break;
            case S_HI:
                PCSF_PUSH(hi_);
                PCSF_PUSH(init_);
                if (eager_restarts) PCSF_RESTARTS(0);
                break;
            case S_INIT:
                PCSF_PUSH(init_);
                if (eager_restarts) PCSF_RESTARTS(0);
                break;
            case S_RESTART:
                PCSF_PUSH(request_);
                PCSF_RESTARTS(tries_ + 1);
                break;
            case S_FINAL:
                PCSF_PUSH(chosen_);
                if (lo_ < chosen_ && chosen_ < hi_) {
                    PCSF_PUSH(lo_);
                    PCSF_PUSH(hi_);
                    PCSF_PUSH(chosen_);
                    PCSF_PUSH(lo_ + GOLDEN * (hi_ - lo_));
                }
                break;
            case B_LO:
                PCSF_PUSH(lo_);
                PCSF_PUSH(hi_);
                PCSF_PUSH(b_min_);
                PCSF_PUSH(b_lo_ + GOLDEN * (b_up_ - b_lo_));
                break;
            case B_HI:
                PCSF_PUSH(hi_);
                PCSF_PUSH(b_min_);
                PCSF_PUSH(b_lo_ + GOLDEN * (b_up_ - b_lo_));
                break;
            case B_MIN:
                PCSF_PUSH(b_min_);
                PCSF_PUSH(b_lo_ + GOLDEN * (b_up_ - b_lo_));
                break;
            default:
                PCSF_PUSH(request_);
                break;
        }
#undef PCSF_RESTARTS
#undef PCSF_PUSH
        if (n_certain) *n_certain = certain < 0 ? n : certain;
        return n;
    }

    // next_draw: the block's draw count, advanced when the search draws a restart point
    PCSF_HD void supply(double lpr, double anc, const DrawTable& draws, uint32_t& next_draw)
    {
        last_x_ = request_;
        last_lpr_ = lpr;
        last_anc_ = anc;
        ++evals_;
        switch (state_) {
            case S_LO:
                flo_ = lpr;
                state_ = S_HI;
                request_ = hi_;
                return;
            case S_HI:
                fhi_ = lpr;
                state_ = S_INIT;
                request_ = init_;
                return;
            case S_INIT:
                x_ = init_;
                fx_ = lpr;
                search_or_finish(draws, next_draw);
                return;
            case S_RESTART:
                fx_ = lpr;
                ++tries_;
                search_or_finish(draws, next_draw);
                return;
            case S_FINAL:
                if (lo_ < chosen_ && chosen_ < hi_) {
                    // gsl_min_fminimizer_set: f(lower), f(upper), f(minimum), then brent_init's f(v)
                    b_lo_ = lo_;
                    b_up_ = hi_;
                    b_min_ = chosen_;
                    state_ = B_LO;
                    request_ = lo_;
                } else {
                    done_ = true;
                }
                return;
            case B_LO:
                f_lo_ = -lpr;
                state_ = B_HI;
                request_ = hi_;
                return;
            case B_HI:
                f_up_ = -lpr;
                state_ = B_MIN;
                request_ = b_min_;
                return;
            case B_MIN:
                f_min_ = -lpr;
                v_ = b_lo_ + GOLDEN * (b_up_ - b_lo_);
                w_ = v_;
                d_ = 0;
                e_ = 0;
                state_ = B_V;
                request_ = v_;
                return;
            case B_V:
                f_v_ = f_w_ = -lpr;
                iter_left_ = 250;
                brent_propose();
                return;
            case B_ITER:
                brent_update(-lpr);
                if (((b_up_ - b_lo_) / b_min_) <= 0.01) {
                    done_ = true;
                    return;
                }
                if (--iter_left_ > 0) brent_propose();
                else done_ = true;
                return;
        }
    }

private:
    static constexpr double GOLDEN = 0.3819660;
    enum State { S_LO, S_HI, S_INIT, S_RESTART, S_FINAL, B_LO, B_HI, B_MIN, B_V, B_ITER };

    PCSF_HD void search_or_finish(const DrawTable& draws, uint32_t& next_draw)
    {
        if (tries_ < 250 && (fx_ <= flo_ || fx_ <= fhi_)) {
            x_ = draws.x[next_draw < draws.n ? next_draw : draws.n - 1];   // (a table always covers a block's draws: DRAW_TABLE_ENTRIES)
            ++next_draw;
            state_ = S_RESTART;
            request_ = x_;
            return;
        }
        chosen_ = (tries_ == 250) ? ((flo_ > fhi_) ? lo_ : hi_) : x_;
        state_ = S_FINAL;  // the reference re-evaluates at the chosen point
        request_ = chosen_;
    }

    PCSF_HD void brent_propose()
    {
        const double z = b_min_;
        double d = e_, e = d_;  // GSL loads state.e into d and state.d into e
        const double w_lower = z - b_lo_, w_upper = b_up_ - z;
        const double tol = 1.4901161193847656e-08 * fabs(z);
        double p = 0, q = 0, r = 0;
        const double mid = 0.5 * (b_lo_ + b_up_);
        if (fabs(e) > tol) {
            r = (z - w_) * (f_min_ - f_v_);
            q = (z - v_) * (f_min_ - f_w_);
            p = (z - v_) * q - (z - w_) * r;
            q = 2 * (q - r);
            if (q > 0) p = -p;
            else q = -q;
            r = e;
            e = d;
        }
        double u;
        if (fabs(p) < fabs(0.5 * q * r) && p < q * w_lower && p < q * w_upper) {
            const double t2 = 2 * tol;
            d = p / q;
            u = z + d;
            if ((u - b_lo_) < t2 || (b_up_ - u) < t2) d = (z < mid) ? tol : -tol;
        } else {
            e = (z < mid) ? b_up_ - z : -(z - b_lo_);
            d = GOLDEN * e;
        }
        if (fabs(d) >= tol) u = z + d;
        else u = z + ((d > 0) ? tol : -tol);
        e_ = e;
        d_ = d;
        u_ = u;
        state_ = B_ITER;
        request_ = u;
    }

    PCSF_HD void brent_update(double f_u)
    {
        const double z = b_min_, u = u_, f_z = f_min_;
        if (f_u <= f_z) {
            if (u < z) { b_up_ = z; f_up_ = f_z; }
            else { b_lo_ = z; f_lo_ = f_z; }
            v_ = w_; f_v_ = f_w_;
            w_ = z; f_w_ = f_z;
            b_min_ = u; f_min_ = f_u;
        } else {
            if (u < z) { b_lo_ = u; f_lo_ = f_u; }
            else { b_up_ = u; f_up_ = f_u; }
            if (f_u <= f_w_ || w_ == z) {
                v_ = w_; f_v_ = f_w_;
                w_ = u; f_w_ = f_u;
            } else if (f_u <= f_v_ || v_ == z || v_ == w_) {
                v_ = u; f_v_ = f_u;
            }
        }
    }

    int state_ = S_LO;
    bool done_ = true;
    double init_ = 0, lo_ = 0, hi_ = 0, request_ = 0;
    double flo_ = 0, fhi_ = 0, fx_ = 0, x_ = 0, chosen_ = 0;
    int tries_ = 0, iter_left_ = 0, evals_ = 0;
    double last_x_ = 0, last_lpr_ = 0, last_anc_ = 0;
    // Brent
    double b_lo_ = 0, b_up_ = 0, b_min_ = 0, f_lo_ = 0, f_up_ = 0, f_min_ = 0;
    double v_ = 0, w_ = 0, d_ = 0, e_ = 0, f_v_ = 0, f_w_ = 0, u_ = 0;
};

// log half-Cauchy(mode 1, scale 0.5) truncated at 0: prior on the tree scale rho (src/omega.hpp:130-141)
// log Gamma(shape 7, scale 0.25) density at kappa - 1 + eps: prior on kappa (src/omega.hpp:143-149)
// Both are evaluated ON THE DEVICE by either engine (chain.hip: log_prior_rho_dev / log_prior_kappa_dev, the same functions of the
// same code object), so that the host engine and the device engine add the same bits.  The host versions below are what rounds 1-3
// used (glibc); they differ from the device's by at most an ulp and remain for the tests that pin the formulas against the oracle.
inline double log_prior_rho(double rho)
{
    const double mode = 1.0, scale = 0.5;
    const double numer = 1.0 / (M_PI * scale * (1.0 + std::pow(((rho - mode) / scale), 2.0)));
    const double cdf0 = std::atan((0.0 - mode) / scale) / M_PI + 0.5;
    return std::log(numer) - std::log(1.0 - cdf0);
}

inline double log_prior_kappa(double kappa)
{
    const double k = kappa - 1.0 + DBL_EPSILON, a = 7.0, b = 0.25;
    const double pdf = (k <= 0.0) ? 0.0 : std::exp((a - 1) * std::log(k / b) - k / b - std::lgamma(a)) / b;
    return std::log(pdf);
}

}  // namespace pcsf
