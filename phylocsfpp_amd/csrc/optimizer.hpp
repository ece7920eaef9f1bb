// optimizer.hpp -- the 1-D maximiser of the MLE / OMEGA strategies as a resumable state machine.
//
// The reference runs, per block, `max_lik_lpr_leaves` (src/fixed_lik.hpp:469-544): a random-restart
// search for a starting point (fit_find_init) followed by GSL's Brent minimiser, every objective
// evaluation being a full likelihood pass.  On the GPU thousands of blocks advance in lock-step:
// each block's optimiser is suspended whenever it needs an objective value, the host batches all
// requests into one expm + pruning launch, and resumes the optimisers with the results.
// Decision logic (comparison order, the result being the LAST evaluation, Brent's d/e swap on
// entry, tolerances) follows the reference and GSL's published min/brent.c so that the sequence
// of evaluation points is the same.
#pragma once

#include <cfloat>
#include <cmath>
#include <random>

namespace pcsf {

class MaxLik1D {
public:
    // objective value convention: `lpr` = the quantity being MAXIMISED (log-likelihood [+ log prior])
    void start(double init, double lo, double hi, std::mt19937* gen)
    {
        init_ = init;
        lo_ = lo;
        hi_ = hi;
        gen_ = gen;
        width_ = std::log(hi) - std::log(lo);
        tries_ = 0;
        state_ = S_LO;
        request_ = lo;
        done_ = false;
        evals_ = 0;
    }
    bool done() const { return done_; }
    // true once the search will never draw from the generator again (the random restarts precede Brent's iteration): a second
    // search sharing the generator may start then and still sees the numbers it would have seen after this one had finished
    bool draws_finished() const { return done_ || state_ >= S_FINAL; }
    double request() const { return request_; }  // x at which the objective is wanted next
    double last_x() const { return last_x_; }
    double last_lpr() const { return last_lpr_; }  // result of max_lik_lpr_leaves = LAST evaluation
    double last_anc() const { return last_anc_; }
    int evaluations() const { return evals_; }

    // Points whose objective value the search will or may ask for next, in the order it would ask, WITHOUT
    // depending on values not yet supplied: xs[0] == request() (certain); the rest is certain up to the next
    // data-dependent decision (lo, hi, init at the start; chosen, lo, hi, chosen, v when Brent is set up) or
    // speculative (the next random restarts, drawn from a COPY of the generator: the restart sequence does not
    // depend on the values, only its length does).  A driver may evaluate all of them in one batch, keep the
    // results and feed supply() from that store: the sequence of supply() calls -- hence every decision and the
    // result -- is the same as when evaluating one point at a time.  Returns the number of points written.
    // eager_restarts: also list random restarts BEFORE it is known that the search needs any (lo, hi and the start not yet
    // evaluated).  Most searches need none -- the start beats both ends -- so those evaluations are usually thrown away: a third
    // of all OMEGA evaluations on a 5,088-block list (profiles/r03).  Worth it only while a round is latency-bound (few blocks).
    // n_certain (optional): how many of the points, from the front, the search WILL ask for whatever the values (the rest are
    // restarts that may never be consumed)
    int upcoming(double* xs, int max_points, int max_restarts, bool eager_restarts = true, int* n_certain = nullptr) const
    {
        int n = 0, certain = -1;
        auto push = [&](double x) {
            if (n < max_points) xs[n++] = x;
        };
        if (done_ || max_points <= 0) return 0;
        auto restarts = [&](int already) {  // the next random starting points
            certain = n;
            if (!gen_ || max_restarts <= 0) return;
            std::mt19937 g = *gen_;
            for (int t = already, k = 0; t < 250 && k < max_restarts && n < max_points; ++t, ++k) {
                std::uniform_real_distribution<> dis(0.0, width_);
                push(std::exp(std::log(lo_) + dis(g)));
            }
        };
        switch (state_) {
            case S_LO:
                push(lo_);
                push(hi_);
                push(init_);
                if (eager_restarts) restarts(0);
                break;
            case S_HI:
                push(hi_);
                push(init_);
                if (eager_restarts) restarts(0);
                break;
            case S_INIT:
                push(init_);
                if (eager_restarts) restarts(0);
                break;
            case S_RESTART:
                push(request_);
                restarts(tries_ + 1);
                break;
            case S_FINAL:
                push(chosen_);
                if (lo_ < chosen_ && chosen_ < hi_) {
                    push(lo_);
                    push(hi_);
                    push(chosen_);
                    push(lo_ + GOLDEN * (hi_ - lo_));
                }
                break;
            case B_LO:
                push(lo_);
                push(hi_);
                push(b_min_);
                push(b_lo_ + GOLDEN * (b_up_ - b_lo_));
                break;
            case B_HI:
                push(hi_);
                push(b_min_);
                push(b_lo_ + GOLDEN * (b_up_ - b_lo_));
                break;
            case B_MIN:
                push(b_min_);
                push(b_lo_ + GOLDEN * (b_up_ - b_lo_));
                break;
            default:
                push(request_);
                break;
        }
        if (n_certain) *n_certain = certain < 0 ? n : certain;
        return n;
    }

    void supply(double lpr, double anc)
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
                search_or_finish();
                return;
            case S_RESTART:
                fx_ = lpr;
                ++tries_;
                search_or_finish();
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

    void search_or_finish()
    {
        if (tries_ < 250 && (fx_ <= flo_ || fx_ <= fhi_)) {
            std::uniform_real_distribution<> dis(0.0, width_);
            const double r = dis(*gen_);
            x_ = std::exp(std::log(lo_) + r);
            state_ = S_RESTART;
            request_ = x_;
            return;
        }
        chosen_ = (tries_ == 250) ? ((flo_ > fhi_) ? lo_ : hi_) : x_;
        state_ = S_FINAL;  // the reference re-evaluates at the chosen point
        request_ = chosen_;
    }

    void brent_propose()
    {
        const double z = b_min_;
        double d = e_, e = d_;  // GSL loads state.e into d and state.d into e
        const double w_lower = z - b_lo_, w_upper = b_up_ - z;
        const double tol = 1.4901161193847656e-08 * std::fabs(z);
        double p = 0, q = 0, r = 0;
        const double mid = 0.5 * (b_lo_ + b_up_);
        if (std::fabs(e) > tol) {
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
        if (std::fabs(p) < std::fabs(0.5 * q * r) && p < q * w_lower && p < q * w_upper) {
            const double t2 = 2 * tol;
            d = p / q;
            u = z + d;
            if ((u - b_lo_) < t2 || (b_up_ - u) < t2) d = (z < mid) ? tol : -tol;
        } else {
            e = (z < mid) ? b_up_ - z : -(z - b_lo_);
            d = GOLDEN * e;
        }
        if (std::fabs(d) >= tol) u = z + d;
        else u = z + ((d > 0) ? tol : -tol);
        e_ = e;
        d_ = d;
        u_ = u;
        state_ = B_ITER;
        request_ = u;
    }

    void brent_update(double f_u)
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

    State state_ = S_LO;
    bool done_ = true;
    std::mt19937* gen_ = nullptr;
    double init_ = 0, lo_ = 0, hi_ = 0, width_ = 0, request_ = 0;
    double flo_ = 0, fhi_ = 0, fx_ = 0, x_ = 0, chosen_ = 0;
    int tries_ = 0, iter_left_ = 0, evals_ = 0;
    double last_x_ = 0, last_lpr_ = 0, last_anc_ = 0;
    // Brent
    double b_lo_ = 0, b_up_ = 0, b_min_ = 0, f_lo_ = 0, f_up_ = 0, f_min_ = 0;
    double v_ = 0, w_ = 0, d_ = 0, e_ = 0, f_v_ = 0, f_w_ = 0, u_ = 0;
};

// log half-Cauchy(mode 1, scale 0.5) truncated at 0: prior on the tree scale rho (src/omega.hpp:130-141)
inline double log_prior_rho(double rho)
{
    const double mode = 1.0, scale = 0.5;
    const double numer = 1.0 / (M_PI * scale * (1.0 + std::pow(((rho - mode) / scale), 2.0)));
    const double cdf0 = std::atan((0.0 - mode) / scale) / M_PI + 0.5;
    return std::log(numer) - std::log(1.0 - cdf0);
}

// log Gamma(shape 7, scale 0.25) density at kappa - 1 + eps: prior on kappa (src/omega.hpp:143-149)
inline double log_prior_kappa(double kappa)
{
    const double k = kappa - 1.0 + DBL_EPSILON, a = 7.0, b = 0.25;
    const double pdf = (k <= 0.0) ? 0.0 : std::exp((a - 1) * std::log(k / b) - k / b - std::lgamma(a)) / b;
    return std::log(pdf);
}

}  // namespace pcsf
