// hostmath.hpp -- host-side numerics of the scoring core (product code, C++17, no dependencies).
//
// What the reference does with GSL once per run_tracks()/run() call is done here once per
// context (FIXED) or once per kappa evaluation (OMEGA):
//   rate matrix        src/instance.hpp:648-685 (ECM), src/omega.hpp:8-128 (kappa/omega F3x4)
//   eigen-system       src/instance.hpp:309-434  (gsl_eigen_nonsymmv + complex LU inverse)
//   equilibrium prior  src/fixed_lik.hpp:281-360
//   P(t) + validation  src/instance.hpp:449-646  (host version; the batched device version
//                      lives in kernels.hip and is checked against this one)
// Both rate matrices are reversible (q_ij = s_ij * pi_j with s symmetric), so the eigen-system
// is obtained from the symmetric matrix D^{1/2} Q D^{-1/2} with a cyclic Jacobi iteration
// instead of a non-symmetric QR.
#pragma once

#include <array>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <vector>

namespace pcsf {

constexpr int NS = 64;  // codon states
using Mat = std::array<double, NS * NS>;
using Vec = std::array<double, NS>;

struct EigenSystem {
    Vec lambda;   // eigenvalues
    Mat right;    // right eigenvectors in columns  (reference: eig.r_s)
    Mat left;     // left eigenvectors in rows = right^-1 (reference: eig.r_s2)
    Vec prior;    // equilibrium distribution, sums to 1
};

// q_ij = S_ij * pi_j, q_ii = -sum_j q_ij, normalised to one expected substitution per unit time
inline void rate_matrix_ecm(const double* S, const double* pi, Mat& Q)
{
    double norm = 0.0;
    for (int i = 0; i < NS; ++i) {
        double diag = 0.0;
        for (int j = 0; j < NS; ++j) {
            const double q = S[i * NS + j] * pi[j];
            Q[i * NS + j] = q;
            diag -= q;
        }
        Q[i * NS + i] = diag;
        norm -= diag * pi[i];
    }
    for (double& q : Q) q = q / norm;
}

// genetic code in A,C,G,T order (codon = 16*n1 + 4*n2 + n3); '*' = stop
inline char amino_acid(int codon)
{
    static const char code[] = "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*Y*YSSSS*CWCLFLF";
    return code[codon];
}

// q_settings = [kappa, omega, sigma, 9 nucleotide-frequency ratios]  (src/run.hpp:75-79)
inline void rate_matrix_omega(const double* qs, Mat& Q, Vec& pi)
{
    // F3x4 weight of a codon: product of three per-position factors, each divided separately
    auto weight_ref = [&](int codon) {
        const int d0 = codon >> 4, d1 = (codon >> 2) & 3, d2 = codon & 3;
        const double f1 = ((d0 == 3) ? 1.0 : qs[3 + d0]) / (1.0 + qs[3] + qs[4] + qs[5]);
        const double f2 = ((d1 == 3) ? 1.0 : qs[6 + d1]) / (1.0 + qs[6] + qs[7] + qs[8]);
        const double f3 = ((d2 == 3) ? 1.0 : qs[9 + d2]) / (1.0 + qs[9] + qs[10] + qs[11]);
        return f1 * f2 * f3;
    };
    const double sigma = qs[2];
    const double stop_mass = weight_ref(48) + weight_ref(50) + weight_ref(56);  // TAA, TAG, TGA
    const double denom = 1.0 - ((1.0 - sigma) * stop_mass);
    for (int c = 0; c < NS; ++c) pi[c] = weight_ref(c) / denom;

    const double kappa = qs[0], omega = qs[1];
    for (int i = 0; i < NS; ++i) {
        for (int j = 0; j < NS; ++j) {
            double rate = 0.0;
            const int x = i ^ j;  // differing nucleotide positions show up as non-zero 2-bit fields
            const int ndiff = ((x & 0x30) != 0) + ((x & 0x0c) != 0) + ((x & 0x03) != 0);
            if (ndiff == 1) {
                const int shift = (x & 0x30) ? 4 : ((x & 0x0c) ? 2 : 0);
                const int a = (i >> shift) & 3, b = (j >> shift) & 3;
                const bool transition = (a + b == 2) || (a + b == 4);  // A<->G or C<->T
                rate = transition ? kappa : 1.0;
                const char ai = amino_acid(i), aj = amino_acid(j);
                rate *= (ai != '*' && aj != '*' && ai != aj) ? omega : 1.0;
                rate *= pi[j];
            }
            Q[i * NS + j] = rate;
        }
    }
    for (int i = 0; i < NS; ++i) {
        double diag = 0.0;
        for (int j = 0; j < NS; ++j)
            if (j != i) diag -= Q[i * NS + j];
        Q[i * NS + i] = diag;
    }
    double norm = 0.0;
    for (int i = 0; i < NS; ++i) norm -= pi[i] * Q[i * NS + i];
    for (double& q : Q) q = q / norm;
}

// Cyclic Jacobi for a symmetric NSxNS matrix held in `a` (overwritten); columns of `vecs`
// receive the orthonormal eigenvectors.  Only rotations, so tiny eigenvalues keep their
// relative accuracy, which the equilibrium extraction relies on.
inline void symmetric_eigen(Mat& a, Vec& vals, Mat& vecs)
{
    vecs.fill(0.0);
    for (int i = 0; i < NS; ++i) vecs[i * NS + i] = 1.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, total = 0.0;
        for (int r = 0; r < NS; ++r)
            for (int c = 0; c < NS; ++c) (r == c ? total : off) += a[r * NS + c] * a[r * NS + c];
        if (off <= 1e-58 * total) break;
        for (int p = 0; p + 1 < NS; ++p)
            for (int q = p + 1; q < NS; ++q) {
                const double g = a[p * NS + q];
                if (std::fabs(g) < 1e-290) {
                    a[p * NS + q] = a[q * NS + p] = 0.0;
                    continue;
                }
                const double zeta = (a[q * NS + q] - a[p * NS + p]) / (2.0 * g);
                const double t = (std::fabs(zeta) > 1e150)
                                     ? 0.5 / zeta
                                     : std::copysign(1.0, zeta) / (std::fabs(zeta) + std::sqrt(1.0 + zeta * zeta));
                const double cs = 1.0 / std::sqrt(1.0 + t * t), sn = t * cs;
                for (int k = 0; k < NS; ++k) {  // columns p,q
                    const double x = a[k * NS + p], y = a[k * NS + q];
                    a[k * NS + p] = cs * x - sn * y;
                    a[k * NS + q] = sn * x + cs * y;
                }
                for (int k = 0; k < NS; ++k) {  // rows p,q
                    const double x = a[p * NS + k], y = a[q * NS + k];
                    a[p * NS + k] = cs * x - sn * y;
                    a[q * NS + k] = sn * x + cs * y;
                }
                a[p * NS + q] = a[q * NS + p] = 0.0;
                for (int k = 0; k < NS; ++k) {
                    const double x = vecs[k * NS + p], y = vecs[k * NS + q];
                    vecs[k * NS + p] = cs * x - sn * y;
                    vecs[k * NS + q] = sn * x + cs * y;
                }
            }
    }
    for (int i = 0; i < NS; ++i) vals[i] = a[i * NS + i];
}

// Q = right * diag(lambda) * left for a rate matrix reversible with respect to `pi`
// (pi need not be normalised).  Returns false if pi has a non-positive entry.
inline bool eigen_reversible(const Mat& Q, const double* pi, EigenSystem& es)
{
    Vec root, iroot;
    for (int i = 0; i < NS; ++i) {
        if (!(pi[i] > 0.0)) return false;
        root[i] = std::sqrt(pi[i]);
        iroot[i] = 1.0 / root[i];
    }
    Mat sym, U;
    for (int i = 0; i < NS; ++i)
        for (int j = 0; j < NS; ++j) sym[i * NS + j] = root[i] * Q[i * NS + j] * iroot[j];
    for (int i = 0; i < NS; ++i)
        for (int j = i + 1; j < NS; ++j)
            sym[i * NS + j] = sym[j * NS + i] = 0.5 * (sym[i * NS + j] + sym[j * NS + i]);
    symmetric_eigen(sym, es.lambda, U);
    for (int i = 0; i < NS; ++i)
        for (int k = 0; k < NS; ++k) {
            es.right[i * NS + k] = iroot[i] * U[i * NS + k];
            es.left[k * NS + i] = U[i * NS + k] * root[i];
        }
    // equilibrium: left eigenvector of the eigenvalue closest to zero, scaled to sum 1
    int z = 0;
    for (int k = 1; k < NS; ++k)
        if (std::fabs(es.lambda[k]) < std::fabs(es.lambda[z])) z = k;
    double mass = 0.0;
    for (int j = 0; j < NS; ++j) mass += es.left[z * NS + j];
    for (int j = 0; j < NS; ++j) es.prior[j] = es.left[z * NS + j] / mass;
    return true;
}

// branch length actually used at tree scale x: the reference keeps branch lengths as float and
// multiplies by the double scale, storing the product back into a float
inline double scaled_branch(float branch_len, double x)
{
    return static_cast<double>(static_cast<float>(static_cast<double>(branch_len) * x));
}

constexpr double P_TOLERANCE = 1e-6;

// Post-processing the reference applies to every P(t): negative entries within tolerance are
// clamped to zero, the diagonal is replaced by one minus the off-diagonal row sum; an entry
// below -tol or a raw row sum off by more than tol invalidates the model.
inline bool finalize_transition_matrix(double* P)
{
    for (int i = 0; i < NS; ++i) {
        double raw_sum = 0.0, diag = 1.0;
        double* row = P + i * NS;
        for (int j = 0; j < NS; ++j) {
            const double v = row[j];
            raw_sum += v;
            if (v < 0.0) {
                if (std::fabs(v) > P_TOLERANCE) return false;
                row[j] = 0.0;
            }
            if (j != i) diag -= row[j];
        }
        if (!(std::fabs(raw_sum - 1.0) <= P_TOLERANCE)) return false;
        row[i] = diag;
    }
    return true;
}

// P = right * diag(exp(lambda t)) * left, then finalize.  Row = parent state, column = child.
inline bool transition_matrix(const EigenSystem& es, double t, double* P)
{
    Mat scaled_left;
    for (int k = 0; k < NS; ++k) {
        const double e = std::exp(es.lambda[k] * t);
        for (int j = 0; j < NS; ++j) scaled_left[k * NS + j] = es.left[k * NS + j] * e;
    }
    for (int i = 0; i < NS; ++i) {
        double* row = P + i * NS;
        for (int j = 0; j < NS; ++j) row[j] = 0.0;
        for (int k = 0; k < NS; ++k) {
            const double r = es.right[i * NS + k];
            const double* l = &scaled_left[k * NS];
            for (int j = 0; j < NS; ++j) row[j] += r * l[j];
        }
    }
    return finalize_transition_matrix(P);
}

}  // namespace pcsf
