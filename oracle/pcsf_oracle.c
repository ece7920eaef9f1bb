/*
 * pcsf_oracle.c -- TEST INFRASTRUCTURE (see pcsf_oracle.h).  CPU restatement of the
 * PhyloCSF++ scoring path; plain loops, fp64, no GPU, no product code.
 * All file:line citations are into the reference checkout (cpockrandt/PhyloCSFpp v1.2.0).
 */
#include "pcsf_oracle.h"

#include <ctype.h>
#include <float.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define K PO_K
#define KK (PO_K * PO_K)

/* ------------------------------------------------------------------------------------------
 * model input
 * ---------------------------------------------------------------------------------------- */

/* src/ecm.hpp:20-67: line l (1-based, l<=63) holds l lower-triangle values S[0..l-1][l];
 * line 65 holds the 64 codon frequencies; the matrix is symmetric with zero diagonal. */
int po_ecm_open(const char *path, double *S, double *pi)
{
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    size_t cap = 1 << 16;
    char *line = (char *)malloc(cap);
    memset(S, 0, sizeof(double) * KK);
    memset(pi, 0, sizeof(double) * K);
    int line_id = 1, ok = 1, have_freq = 0;
    while (fgets(line, (int)cap, f)) {
        const char *p = line;
        char *end;
        int field = 0;
        if (line_id <= 63) {
            for (double v = strtod(p, &end); p != end; v = strtod(p, &end)) {
                p = end;
                if (field < line_id) {
                    S[field * K + line_id] = v;
                    S[line_id * K + field] = v;
                }
                ++field;
            }
            if (field != line_id) ok = 0;
        } else if (line_id == 65) {
            for (double v = strtod(p, &end); p != end; v = strtod(p, &end)) {
                p = end;
                if (field < K) pi[field] = v;
                ++field;
            }
            if (field != K) ok = 0;
            have_freq = 1;
        }
        ++line_id;
    }
    free(line);
    fclose(f);
    return (ok && have_freq) ? 0 : -2;
}

/* recursive-descent newick reader for strictly binary trees.  Semantics of
 * src/newick.hpp:31-91: labels lower-cased, lengths are [0-9.]+ read with strtod, the root has
 * no length.  Numbering of src/newick.hpp:100-118,170-229. */
typedef struct pnode {
    struct pnode *l, *r, *parent;
    double len;
    char label[64];
    int id;
} pnode;

static pnode *pn_new(pnode *parent)
{
    pnode *n = (pnode *)calloc(1, sizeof(pnode));
    n->parent = parent;
    n->id = -1;
    return n;
}

static void pn_free(pnode *n)
{
    if (!n) return;
    pn_free(n->l);
    pn_free(n->r);
    free(n);
}

static int parse_len(const char **s, double *out)
{
    char buf[64];
    int k = 0;
    while ((isdigit((unsigned char)**s) || **s == '.') && k < 63) buf[k++] = *(*s)++;
    buf[k] = 0;
    if (k == 0) return -1;
    *out = strtod(buf, NULL);
    return 0;
}

static int parse_node(const char **s, pnode *n)
{
    if (**s == '(') {
        ++*s;
        n->l = pn_new(n);
        if (parse_node(s, n->l)) return -1;
        if (**s != ',') return -1;
        ++*s;
        n->r = pn_new(n);
        if (parse_node(s, n->r)) return -1;
        if (**s != ')') return -1; /* non-binary trees are not supported by the reference either */
        ++*s;
    } else {
        int k = 0;
        while (**s && **s != '(' && **s != ',' && **s != ')' && **s != ':' && **s != ';') {
            if (k < 63) n->label[k++] = (char)tolower((unsigned char)**s);
            ++*s;
        }
        n->label[k] = 0;
        if (k == 0) return -1;
    }
    if (**s == ':') {
        ++*s;
        if (parse_len(s, &n->len)) return -1;
    }
    return 0;
}

static int count_leaves(const pnode *n) { return n->l ? count_leaves(n->l) + count_leaves(n->r) : 1; }

static void annotate(pnode *n, int *leaf_id, int *inner_id)
{
    if (n->l) {
        annotate(n->l, leaf_id, inner_id);
        annotate(n->r, leaf_id, inner_id);
        n->id = (*inner_id)++;
    } else {
        n->id = (*leaf_id)++;
    }
}

static void flatten(const pnode *n, po_tree *t)
{
    int i = n->id;
    t->child1[i] = n->l ? (int16_t)n->l->id : -1;
    t->child2[i] = n->r ? (int16_t)n->r->id : -1;
    t->parent[i] = n->parent ? (int16_t)n->parent->id : -1;
    t->sibling[i] = -1;
    if (n->parent) t->sibling[i] = (int16_t)((n->parent->l == n) ? n->parent->r->id : n->parent->l->id);
    t->branch_len[i] = (float)n->len; /* newick_elem::branch_length is float (newick.hpp:27,188,211) */
    t->branch_len_d[i] = n->len;
    memcpy(t->label[i], n->label, sizeof n->label);
    if (n->l) {
        flatten(n->l, t);
        flatten(n->r, t);
    }
}

int po_tree_parse(const char *text, po_tree *t)
{
    size_t len = strlen(text);
    char *buf = (char *)malloc(len + 1);
    size_t k = 0;
    for (size_t i = 0; i < len; ++i)
        if (!isspace((unsigned char)text[i])) buf[k++] = text[i]; /* newick.hpp:157 */
    buf[k] = 0;
    const char *s = buf;
    pnode *root = pn_new(NULL);
    int rc = parse_node(&s, root);
    if (rc == 0 && !(*s == 0 || *s == ';')) rc = -1;
    if (rc == 0) {
        int nl = count_leaves(root);
        int n = 2 * nl - 1;
        if (n > PO_MAX_NODES || nl < 2) rc = -2;
        else {
            memset(t, 0, sizeof(*t));
            t->n = n;
            t->nl = nl;
            int leaf_id = 0, inner_id = nl;
            annotate(root, &leaf_id, &inner_id);
            flatten(root, t);
        }
    }
    pn_free(root);
    free(buf);
    return rc;
}

int po_tree_open(const char *path, po_tree *t)
{
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *buf = (char *)malloc((size_t)sz + 1);
    size_t got = fread(buf, 1, (size_t)sz, f);
    buf[got] = 0;
    fclose(f);
    int rc = po_tree_parse(buf, t);
    free(buf);
    return rc;
}

int po_model_open(const char *prefix, po_model *m)
{
    char path[4096];
    snprintf(path, sizeof path, "%s_coding.ECM", prefix); /* src/models.hpp:1773-1775 */
    if (po_ecm_open(path, m->S_c, m->pi_c)) return -1;
    snprintf(path, sizeof path, "%s_noncoding.ECM", prefix);
    if (po_ecm_open(path, m->S_nc, m->pi_nc)) return -2;
    snprintf(path, sizeof path, "%s.nh", prefix);
    if (po_tree_open(path, &m->tree)) return -3;
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * rate matrices
 * ---------------------------------------------------------------------------------------- */

/* src/instance.hpp:648-685 */
void po_q_from_ecm(const double *S, const double *pi, double *Q)
{
    double scale = 0.0;
    for (int i = 0; i < K; ++i) {
        double sum = 0.0;
        for (int j = 0; j < K; ++j) {
            const double v = S[i * K + j] * pi[j];
            Q[i * K + j] = v;
            sum -= v;
        }
        Q[i * K + i] = sum;
        scale -= sum * pi[i];
    }
    for (int i = 0; i < KK; ++i) Q[i] = Q[i] / scale;
}

/* genetic code, A0 C1 G2 T3, codon = 16*n1+4*n2+n3 (src/translation.hpp:5-27) */
static const char po_aa_table[65] = "KNKNTTTTRSRSIIMIQHQHPPPPRRRRLLLLEDEDAAAAGGGGVVVV*Y*YSSSS*CWCLFLF";

/* src/omega.hpp:8-19 */
static double pi_expr_sc(const double *v, int codon)
{
    const int i1 = codon / 16, i2 = (codon - 16 * i1) / 4, i3 = codon - 16 * i1 - 4 * i2;
    const double f1 = ((i1 == 3) ? 1.0 : v[3 + i1]) / (1.0 + v[3] + v[4] + v[5]);
    const double f2 = ((i2 == 3) ? 1.0 : v[6 + i2]) / (1.0 + v[6] + v[7] + v[8]);
    const double f3 = ((i3 == 3) ? 1.0 : v[9 + i3]) / (1.0 + v[9] + v[10] + v[11]);
    return f1 * f2 * f3;
}

/* src/omega.hpp:21-128: pi_expr, comp_q_p14n, comp_q_scale, compute_q_..._omega */
void po_q_from_omega(const double *v, double *Q, double *pi)
{
    const double sigma = v[2];
    const double denom = 1.0 - ((1.0 - sigma) * (pi_expr_sc(v, 48) + pi_expr_sc(v, 50) + pi_expr_sc(v, 56)));
    for (int i = 0; i < K; ++i) pi[i] = pi_expr_sc(v, i) / denom;

    const double kappa = v[0], omega = v[1];
    for (int i = 0; i < K; ++i) {
        const int i1 = i / 16, i2 = (i / 4) % 4, i3 = i % 4;
        const char iaa = po_aa_table[i];
        for (int j = 0; j < K; ++j) {
            const int j1 = j / 16, j2 = (j / 4) % 4, j3 = j % 4;
            double val = 0.0;
            const int changes = (i1 != j1) + (i2 != j2) + (i3 != j3);
            if (changes == 1) {
                int transition = 0;
                if (i1 != j1 && (i1 + j1 == 2 || i1 + j1 == 4)) transition = 1;
                if (i2 != j2 && (i2 + j2 == 2 || i2 + j2 == 4)) transition = 1;
                if (i3 != j3 && (i3 + j3 == 2 || i3 + j3 == 4)) transition = 1;
                val = transition ? kappa : 1.0;
                const char jaa = po_aa_table[j];
                val *= (iaa != '*' && jaa != '*' && iaa != jaa) ? omega : 1.0;
                val *= pi[j];
            }
            Q[i * K + j] = val;
        }
    }
    for (int i = 0; i < K; ++i) {
        double val = 0.0;
        for (int j = 0; j < K; ++j)
            if (i != j) val -= Q[i * K + j];
        Q[i * K + i] = val;
    }
    double factor = 0.0;
    for (int i = 0; i < K; ++i) factor -= pi[i] * Q[i * K + i];
    for (int i = 0; i < KK; ++i) Q[i] = Q[i] / factor;
}

/* ------------------------------------------------------------------------------------------
 * eigen-system.  The reference calls gsl_eigen_nonsymmv + complex LU inverse
 * (src/instance.hpp:331,344-345).  Both rate matrices on the path are reversible
 * (q_ij = s_ij pi_j with s symmetric, src/instance.hpp:669, src/omega.hpp:70-74), so
 * B = D^{1/2} Q D^{-1/2}, D = diag(pi), is symmetric: B = U L U^T by cyclic Jacobi, and
 * Sr = D^{-1/2} U, Sinv = U^T D^{1/2}.
 * ---------------------------------------------------------------------------------------- */
static inline int perturbation(void);   /* control experiments, see po_set_perturbation */
static void jacobi_sym(double *a /*KxK, destroyed*/, double *w, double *v)
{
    for (int i = 0; i < K; ++i)
        for (int j = 0; j < K; ++j) v[i * K + j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 64; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int p = 0; p < K; ++p) {
            diag += a[p * K + p] * a[p * K + p];
            for (int q = p + 1; q < K; ++q) off += a[p * K + q] * a[p * K + q];
        }
        if (off <= 1e-60 * diag || off == 0.0) break;
        /* control experiment only (po_set_perturbation bit 32): the same sweeps with the pairs visited in the opposite order -- an equally
         * valid cyclic Jacobi whose eigen-system differs from this one's the way two solvers' do */
        const int reversed = (perturbation() & 32) != 0;
        for (int pp = 0; pp < K - 1; ++pp) {
            const int p = reversed ? K - 2 - pp : pp;
            for (int qq = p + 1; qq < K; ++qq) {
                const int q = reversed ? K - 1 - (qq - (p + 1)) : qq;
                const double apq = a[p * K + q];
                if (apq == 0.0) continue;
                const double app = a[p * K + p], aqq = a[q * K + q];
                if (fabs(apq) < 1e-300) { a[p * K + q] = a[q * K + p] = 0.0; continue; }
                const double theta = (aqq - app) / (2.0 * apq);
                double t;
                if (fabs(theta) > 1e150) t = 0.5 / theta;
                else t = ((theta >= 0.0) ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                /* A <- J^T A J with J = rotation in the (p,q) plane */
                for (int k = 0; k < K; ++k) {
                    const double akp = a[k * K + p], akq = a[k * K + q];
                    a[k * K + p] = c * akp - s * akq;
                    a[k * K + q] = s * akp + c * akq;
                }
                for (int k = 0; k < K; ++k) {
                    const double apk = a[p * K + k], aqk = a[q * K + k];
                    a[p * K + k] = c * apk - s * aqk;
                    a[q * K + k] = s * apk + c * aqk;
                }
                a[p * K + q] = a[q * K + p] = 0.0;
                for (int k = 0; k < K; ++k) {
                    const double vkp = v[k * K + p], vkq = v[k * K + q];
                    v[k * K + p] = c * vkp - s * vkq;
                    v[k * K + q] = s * vkp + c * vkq;
                }
            }
        }
    }
    for (int i = 0; i < K; ++i) w[i] = a[i * K + i];
}

void po_eig(const double *Q, const double *pi, po_qdiag *qd)
{
    double B[KK], U[KK], sq[K], isq[K];
    memcpy(qd->Q, Q, sizeof(double) * KK);
    for (int i = 0; i < K; ++i) {
        sq[i] = sqrt(pi[i]);
        isq[i] = 1.0 / sq[i];
    }
    for (int i = 0; i < K; ++i)
        for (int j = 0; j < K; ++j) B[i * K + j] = sq[i] * Q[i * K + j] * isq[j];
    for (int i = 0; i < K; ++i)
        for (int j = i + 1; j < K; ++j) {
            const double m = 0.5 * (B[i * K + j] + B[j * K + i]);
            B[i * K + j] = B[j * K + i] = m;
        }
    jacobi_sym(B, qd->lam, U);
    for (int i = 0; i < K; ++i)
        for (int k = 0; k < K; ++k) {
            qd->Sr[i * K + k] = isq[i] * U[i * K + k];
            qd->Sinv[k * K + i] = U[i * K + k] * sq[i];
        }
    /* equilibrium = left eigenvector of the smallest-|lambda| eigenvalue, normalised to sum 1
     * (src/fixed_lik.hpp:323-346) */
    int mn = 0;
    double best = fabs(qd->lam[0]);
    for (int i = 1; i < K; ++i)
        if (fabs(qd->lam[i]) < best) { best = fabs(qd->lam[i]); mn = i; }
    double mass = 0.0;
    for (int j = 0; j < K; ++j) mass += qd->Sinv[mn * K + j];
    for (int j = 0; j < K; ++j) qd->prior[j] = qd->Sinv[mn * K + j] / mass;
}

/* src/instance.hpp:529-549 (expm) and :602-640 (clamp / validate / diagonal).
 * target_clones only widens the SIMD lanes of the j loops (no FMA contraction: the build uses
 * -ffp-contract=off), so every clone returns bit-identical results on every host. */
__attribute__((target_clones("avx2", "default")))
int po_make_P(const po_qdiag *qd, double t, double *P)
{
    double e[K], D[KK];
    for (int k = 0; k < K; ++k) e[k] = exp(qd->lam[k] * t);
    for (int k = 0; k < K; ++k)
        for (int j = 0; j < K; ++j) D[k * K + j] = qd->Sinv[k * K + j] * e[k];
    /* reference-BLAS dgemm loop order: i, k, j */
    for (int i = 0; i < K; ++i) {
        double *row = P + i * K;
        for (int j = 0; j < K; ++j) row[j] = 0.0;
        for (int k = 0; k < K; ++k) {
            const double a = qd->Sr[i * K + k];
            if (a != 0.0)
                for (int j = 0; j < K; ++j) row[j] += a * D[k * K + j];
        }
    }
    const double tol = 1e-6;
    for (int i = 0; i < K; ++i) {
        double total = 0.0, smii = 1.0;
        for (int j = 0; j < K; ++j) {
            const double cell = P[i * K + j];
            total += cell;
            if (cell < 0.0) {
                if (fabs(cell) > tol) return -1;
                P[i * K + j] = 0.0;
            }
            if (i != j) smii -= P[i * K + j];
        }
        if (!(fabs(total - 1.0) <= tol)) return -1;
        P[i * K + i] = smii;
    }
    return 0;
}

int po_make_all_P(const po_qdiag *qd, const po_tree *tree, double x, double *P)
{
    for (int b = 0; b < tree->n - 1; ++b) {
        /* instantiate_tree: float *= double, stored back as float (src/instance.hpp:299-307) */
        const float tf = (float)((double)tree->branch_len[b] * x);
        if (po_make_P(qd, (double)tf, P + (size_t)b * KK)) return -1;
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * pruning
 * ---------------------------------------------------------------------------------------- */

/* CONTROL EXPERIMENT ONLY (tests/golden/make_omega_control.py, profiles/r04/omega_control.txt): the same algorithm with the terms of
 * some sums added in another order -- every objective value moves by a few ulp, nothing else changes.  Shows how far the END results of
 * the MLE / OMEGA procedures move under a perturbation of that size, i.e. how much of the difference between two implementations
 * (this oracle and the HIP path; two builds of the reference) the procedure itself accounts for.  Bits: 1 = the 64-term sums of
 * internal-node and gap messages run from state 63 down to 0; 2 = the root sum runs backwards; 4 = a block's site log-likelihoods are
 * added from the last site to the first.  0 (the default, and what every parity test uses) = the reference's order.
 * Round 5, bits 8 and 16: a perturbation of the SIZE measured between the HIP path and this oracle instead of an ulp -- every site's
 * ln z is multiplied by 1 + r 2^-43 (|r| <= 1, a hash of the site, of the bits of the tree scale and of the bit set: two independent
 * draws), i.e. moved by up to 1.1e-13 relative; per-codon scores of the two implementations agree to 6e-14 (tests/test_gpu_fixed.py),
 * objective values to 1.3e-13 (layer 2).  The few-ulp control understates how often an MLE search ends a Brent
 * step apart between two IMPLEMENTATIONS (profiles/r05/mle_control_full.txt).
 * Bit 32 (OMEGA): the cyclic Jacobi sweeps of the eigen-solver visit their pairs in the opposite order -- another valid solver.  At small
 * tree scales the objective sees the solver's rounding (the reference clamps noise-level entries of P to zero and rebuilds the diagonal):
 * two solvers agree to ~1e-7 relative at a tree scale of 0.02, not to 1e-13, which no other bit reproduces.
 * PO_PERTURB=<bits> in the environment sets it for the command-line driver. */
static int g_perturb = -1;
void po_set_perturbation(int bits) { g_perturb = bits; }
static inline int perturbation(void)
{
    if (g_perturb < 0) {
        const char *e = getenv("PO_PERTURB");
        g_perturb = e ? atoi(e) : 0;
    }
    return g_perturb;
}

/* src/fixed_lik.hpp:105-123 */
static inline double child_msg(const double *Prow, int is_leaf, uint8_t code, const double *alpha, int reversed)
{
    if (!is_leaf) {
        double d = 0.0;
        if (reversed)
            for (int b = K - 1; b >= 0; --b) d += Prow[b] * alpha[b];
        else
            for (int b = 0; b < K; ++b) d += Prow[b] * alpha[b];
        return d;
    }
    if (code == PO_GAP) {
        double d = 0.0;
        if (reversed)
            for (int b = K - 1; b >= 0; --b) d += Prow[b];
        else
            for (int b = 0; b < K; ++b) d += Prow[b];
        return d;
    }
    return Prow[code];
}

/* src/fixed_lik.hpp:125-164 */
double po_prune_site(const po_tree *tree, const double *P, const double *prior, const uint8_t *codons,
                     size_t leaf_stride, size_t site, double *alpha_root)
{
    const int n = tree->n, nl = tree->nl;
    const int pert = perturbation(), rev = pert & 1;
    double alpha[(PO_MAX_NODES / 2 + 1) * K];
    for (int i = nl; i < n; ++i) {
        const int lc = tree->child1[i], rc = tree->child2[i];
        const double *ls = P + (size_t)lc * KK, *rs = P + (size_t)rc * KK;
        const int l_leaf = lc < nl, r_leaf = rc < nl;
        const uint8_t lcode = l_leaf ? codons[(size_t)lc * leaf_stride + site] : 0;
        const uint8_t rcode = r_leaf ? codons[(size_t)rc * leaf_stride + site] : 0;
        const double *la = l_leaf ? NULL : alpha + (size_t)(lc - nl) * K;
        const double *ra = r_leaf ? NULL : alpha + (size_t)(rc - nl) * K;
        double *out = alpha + (size_t)(i - nl) * K;
        for (int a = 0; a < K; ++a) {
            const double r1 = child_msg(ls + a * K, l_leaf, lcode, la, rev);
            const double r2 = child_msg(rs + a * K, r_leaf, rcode, ra, rev);
            out[a] = r1 * r2;
        }
    }
    const double *ar = alpha + (size_t)(n - 1 - nl) * K;
    double z = 0.0;
    if (pert & 2)
        for (int a = K - 1; a >= 0; --a) z += prior[a] * ar[a];
    else
        for (int a = 0; a < K; ++a) z += prior[a] * ar[a];
    if (alpha_root) memcpy(alpha_root, ar, sizeof(double) * K);
    return z;
}

static long g_eval_count = 0;
#ifdef _OPENMP
#pragma omp threadprivate(g_eval_count)
#endif
long po_last_eval_count(void) { return g_eval_count; }

/* src/fixed_lik.hpp:362-449 */
int po_lpr_leaves(const po_qdiag *qd, const po_tree *tree, double x, const uint8_t *codons, size_t n_sites,
                  size_t leaf_stride, int compute_anc, double *lpr, double *elpr_anc, double *per_codon)
{
    double *P = (double *)malloc(sizeof(double) * KK * (size_t)(tree->n - 1));
    ++g_eval_count;
    if (po_make_all_P(qd, tree, x, P)) {
        free(P);
        return -1;
    }
    double lprior[K], ar[K];
    if (compute_anc)
        for (int i = 0; i < K; ++i) lprior[i] = log(qd->prior[i]);
    double l = 0.0, anc = 0.0;
    const int backwards = (perturbation() & 4) != 0;   /* control experiment only, see po_set_perturbation */
    for (size_t k = 0; k < n_sites; ++k) {
        const size_t s = backwards ? n_sites - 1 - k : k;
        const double z = po_prune_site(tree, P, qd->prior, codons, leaf_stride, s, compute_anc ? ar : NULL);
        double lz = log(z);
        if (perturbation() & 24) {   /* control experiment only: an implementation-sized relative difference, see po_set_perturbation */
            uint64_t h, xb;
            memcpy(&xb, &x, sizeof xb);
            h = (xb ^ (uint64_t)s * 0x9E3779B97F4A7C15ull ^ (uint64_t)(perturbation() & 24) << 56) * 0xD6E8FEB86659FD93ull;
            h ^= h >> 32;
            h *= 0xD6E8FEB86659FD93ull;
            h ^= h >> 32;
            lz *= 1.0 + ((double)(int64_t)h / 9223372036854775808.0) * 0x1p-43;
        }
        l += lz;
        if (per_codon) per_codon[s] = lz;
        if (compute_anc) {
            /* node_posterior(root), src/fixed_lik.hpp:215-246 */
            for (int k = 0; k < K; ++k) {
                const double pr = (z == 0.0) ? 0.0 : ar[k] * qd->prior[k] / z;
                anc += lprior[k] * pr;
            }
        }
    }
    free(P);
    *lpr = l;
    if (elpr_anc) *elpr_anc = anc;
    return 0;
}

/* PhyloCSFModel_make (src/instance.hpp:687-712): Q from the ECM, eigen-system, and a first
 * PhyloModel_make at scale 1 whose only observable effect is that it can throw. */
static int ecm_instance(const double *S, const double *pi, const po_tree *tree, po_qdiag *qd)
{
    double Q[KK];
    po_q_from_ecm(S, pi, Q);
    po_eig(Q, pi, qd);
    double *P = (double *)malloc(sizeof(double) * KK * (size_t)(tree->n - 1));
    const int rc = po_make_all_P(qd, tree, 1.0, P);
    free(P);
    return rc;
}

/* src/run.hpp:35-55 */
int po_run_tracks(const po_model *m, const uint8_t *codons, size_t n_sites, size_t leaf_stride, double *out)
{
    po_qdiag *qc = (po_qdiag *)malloc(sizeof(po_qdiag)), *qn = (po_qdiag *)malloc(sizeof(po_qdiag));
    int rc = 0;
    double *nc = (double *)malloc(sizeof(double) * (n_sites ? n_sites : 1));
    double lc, ln_;
    if (ecm_instance(m->S_c, m->pi_c, &m->tree, qc) || ecm_instance(m->S_nc, m->pi_nc, &m->tree, qn)) rc = -1;
    if (!rc && po_lpr_leaves(qc, &m->tree, 1.0, codons, n_sites, leaf_stride, 0, &lc, NULL, out)) rc = -1;
    if (!rc && po_lpr_leaves(qn, &m->tree, 1.0, codons, n_sites, leaf_stride, 0, &ln_, NULL, nc)) rc = -1;
    if (!rc)
        for (size_t i = 0; i < n_sites; ++i) out[i] = 10.0 * (out[i] - nc[i]) / log(10.0);
    free(nc);
    free(qc);
    free(qn);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * std::mt19937 + libstdc++ std::uniform_real_distribution<double> (src/fixed_lik.hpp:482-487)
 * ---------------------------------------------------------------------------------------- */
void po_mt_seed(po_mt19937 *g, uint32_t seed)
{
    g->mt[0] = seed;
    for (int i = 1; i < 624; ++i) g->mt[i] = 1812433253u * (g->mt[i - 1] ^ (g->mt[i - 1] >> 30)) + (uint32_t)i;
    g->idx = 624;
}

uint32_t po_mt_next(po_mt19937 *g)
{
    if (g->idx >= 624) {
        for (int i = 0; i < 624; ++i) {
            const uint32_t y = (g->mt[i] & 0x80000000u) | (g->mt[(i + 1) % 624] & 0x7fffffffu);
            g->mt[i] = g->mt[(i + 397) % 624] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
        }
        g->idx = 0;
    }
    uint32_t y = g->mt[g->idx++];
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

/* generate_canonical<double,53>: two 32-bit draws, low word first */
double po_uniform_real(po_mt19937 *g, double a, double b)
{
    double sum = 0.0, tmp = 1.0;
    for (int k = 0; k < 2; ++k) {
        sum += (double)po_mt_next(g) * tmp;
        tmp *= 4294967296.0;
    }
    double r = sum / tmp;
    if (r >= 1.0) r = nextafter(1.0, 0.0);
    return r * (b - a) + a;
}

/* ------------------------------------------------------------------------------------------
 * 1-D maximiser: fit_find_init + GSL Brent (src/fixed_lik.hpp:469-544; GSL min/brent.c)
 * ---------------------------------------------------------------------------------------- */
typedef struct objective {
    double (*f)(double x, struct objective *self); /* returns -lpr like the minimizer_* functions */
    double x, lpr, elpr_anc;                       /* minimizer_params_t: state of the LATEST call */
    int failed;                                    /* a P matrix failed validation */
    /* context */
    const po_tree *tree;
    const uint8_t *codons;
    size_t n_sites, leaf_stride;
    int compute_anc;
    po_qdiag *qd;
    /* omega model state (instance_t::q_settings / tree_settings) */
    double q_settings[12];
    double tree_settings;
} objective;

typedef struct {
    double v, w, d, e, f_v, f_w;
    double x_min, f_min, x_lo, f_lo, x_up, f_up;
} brent_t;

static void brent_set(brent_t *b, objective *o, double x_min, double lo, double hi)
{
    /* gsl_min_fminimizer_set: f(lower), f(upper), f(minimum) in this order, then brent_init */
    b->x_lo = lo;
    b->x_up = hi;
    b->x_min = x_min;
    b->f_lo = o->f(lo, o);
    b->f_up = o->f(hi, o);
    b->f_min = o->f(x_min, o);
    const double golden = 0.3819660;
    b->v = lo + golden * (hi - lo);
    b->w = b->v;
    b->d = 0;
    b->e = 0;
    b->f_v = o->f(b->v, o);
    b->f_w = b->f_v;
}

static void brent_iterate(brent_t *b, objective *o)
{
    const double x_left = b->x_lo, x_right = b->x_up, z = b->x_min;
    double d = b->e, e = b->d; /* sic: GSL loads them swapped */
    double u, f_u;
    const double v = b->v, w = b->w, f_v = b->f_v, f_w = b->f_w, f_z = b->f_min;
    const double golden = 0.3819660;
    const double w_lower = z - x_left, w_upper = x_right - z;
    const double tolerance = 1.4901161193847656e-08 * fabs(z); /* GSL_SQRT_DBL_EPSILON */
    double p = 0, q = 0, r = 0;
    const double midpoint = 0.5 * (x_left + x_right);

    if (fabs(e) > tolerance) {
        r = (z - w) * (f_z - f_v);
        q = (z - v) * (f_z - f_w);
        p = (z - v) * q - (z - w) * r;
        q = 2 * (q - r);
        if (q > 0) p = -p;
        else q = -q;
        r = e;
        e = d;
    }
    if (fabs(p) < fabs(0.5 * q * r) && p < q * w_lower && p < q * w_upper) {
        const double t2 = 2 * tolerance;
        d = p / q;
        u = z + d;
        if ((u - x_left) < t2 || (x_right - u) < t2) d = (z < midpoint) ? tolerance : -tolerance;
    } else {
        e = (z < midpoint) ? x_right - z : -(z - x_left);
        d = golden * e;
    }
    if (fabs(d) >= tolerance) u = z + d;
    else u = z + ((d > 0) ? tolerance : -tolerance);
    b->e = e;
    b->d = d;
    f_u = o->f(u, o);
    if (f_u <= f_z) {
        if (u < z) { b->x_up = z; b->f_up = f_z; }
        else { b->x_lo = z; b->f_lo = f_z; }
        b->v = w; b->f_v = f_w;
        b->w = z; b->f_w = f_z;
        b->x_min = u; b->f_min = f_u;
    } else {
        if (u < z) { b->x_lo = u; b->f_lo = f_u; }
        else { b->x_up = u; b->f_up = f_u; }
        if (f_u <= f_w || w == z) {
            b->v = w; b->f_v = f_w;
            b->w = u; b->f_w = f_u;
        } else if (f_u <= f_v || v == z || v == w) {
            b->v = u; b->f_v = f_u;
        }
    }
}

/* returns the lpr of the LAST evaluation (src/fixed_lik.hpp:542-543); -1 in *failed on throw */
static void max_lik(objective *o, double init, double lo, double hi, po_mt19937 *gen, double *lpr, double *elpr_anc)
{
    /* fit_find_init, src/fixed_lik.hpp:469-509 */
    const double width = log(hi) - log(lo);
    const double flo = -o->f(lo, o);
    if (o->failed) return;
    const double fhi = -o->f(hi, o);
    if (o->failed) return;
    double x = init;
    double fx = -o->f(init, o);
    if (o->failed) return;
    int i = 0;
    const int max_tries = 250;
    while (i < max_tries && (fx <= flo || fx <= fhi)) {
        const double r = po_uniform_real(gen, 0.0, width);
        x = exp(log(lo) + r);
        fx = -o->f(x, o);
        if (o->failed) return;
        ++i;
    }
    if (i == max_tries) {
        if (flo > fhi) { o->x = lo; o->lpr = flo; }
        else { o->x = hi; o->lpr = fhi; }
    }
    o->f(o->x, o);
    if (o->failed) return;
    /* max_lik_lpr_leaves, src/fixed_lik.hpp:511-544 */
    if (lo < o->x && o->x < hi) {
        brent_t b;
        brent_set(&b, o, o->x, lo, hi);
        if (o->failed) return;
        int max_iter = 250;
        do {
            brent_iterate(&b, o);
            if (o->failed) return;
            if (((b.x_up - b.x_lo) / b.x_min) <= 0.01) break;
            --max_iter;
        } while (max_iter > 0);
    }
    *lpr = o->lpr;
    if (elpr_anc) *elpr_anc = o->elpr_anc;
}

/* test hook: max_lik on a caller-supplied objective (to be maximised); records evaluation points */
typedef struct { double (*fn)(double, void *); void *user; double *xs; int cap, n; } dbg_ctx;
static dbg_ctx *g_dbg;
static double obj_debug(double x, objective *o)
{
    o->x = x;
    if (g_dbg->xs && g_dbg->n < g_dbg->cap) g_dbg->xs[g_dbg->n] = x;
    g_dbg->n++;
    o->lpr = g_dbg->fn(x, g_dbg->user);
    return -o->lpr;
}
int po_debug_max_lik(double (*fn)(double, void *), void *user, double init, double lo, double hi, uint32_t seed,
                     int skip_draws, double *xs_out, int cap, int *n_evals, double *last_lpr)
{
    po_mt19937 gen;
    po_mt_seed(&gen, seed);
    for (int i = 0; i < skip_draws; ++i) (void)po_mt_next(&gen);
    dbg_ctx d = {fn, user, xs_out, cap, 0};
    g_dbg = &d;
    objective o;
    memset(&o, 0, sizeof o);
    o.f = obj_debug;
    double lpr = 0;
    max_lik(&o, init, lo, hi, &gen, &lpr, NULL);
    *n_evals = d.n;
    *last_lpr = lpr;
    return 0;
}

/* po_run_trace: every objective evaluation of a run, in order (tests replay them on the device) */
static __thread po_trace_entry *g_trace = NULL;
static __thread int g_trace_cap = 0, g_trace_n = 0, g_trace_kind = 0;

static void trace_add(int kind, double x, const objective *o, double lpr_raw, int failed)
{
    if (!g_trace) return;
    if (g_trace_n < g_trace_cap) {
        po_trace_entry *e = &g_trace[g_trace_n];
        e->kind = kind;
        e->failed = failed;
        e->x = x;
        e->tree_scale = o->tree_settings;
        memcpy(e->q_settings, o->q_settings, sizeof e->q_settings);
        e->lpr = lpr_raw;
    }
    ++g_trace_n;
}

/* minimizer_lpr_leaves, src/fixed_lik.hpp:460-467 */
static double obj_tree_scale(double x, objective *o)
{
    o->x = x;
    if (po_lpr_leaves(o->qd, o->tree, x, o->codons, o->n_sites, o->leaf_stride, o->compute_anc, &o->lpr,
                      &o->elpr_anc, NULL)) {
        o->failed = 1;
        trace_add(g_trace_kind, x, o, NAN, 1);
        return NAN;
    }
    trace_add(g_trace_kind, x, o, o->lpr, 0);
    return -o->lpr;
}

/* src/omega.hpp:130-141 */
double po_lpr_rho(double rho)
{
    const double mode = 1.0, scale = 0.5;
    const double numer = 1.0 / (M_PI * scale * (1.0 + pow(((rho - mode) / scale), 2.0)));
    const double cauchy_cdf = atan((0.0 - mode) / scale) / M_PI + 0.5;
    const double denom = 1.0 - cauchy_cdf;
    return log(numer) - log(denom);
}

/* src/omega.hpp:143-149 with gsl_ran_gamma_pdf(x; a=7, b=0.25) =
 * exp((a-1) log(x/b) - x/b - lgamma(a)) / b for x > 0, 0 for x <= 0 (a != 1) */
double po_lpr_kappa(double kappa)
{
    const double k = kappa - 1.0 + DBL_EPSILON;
    const double a = 7.0, b = 0.25;
    double g;
    if (k < 0) g = 0.0;
    else if (k == 0) g = 0.0;
    else g = exp((a - 1) * log(k / b) - k / b - lgamma(a)) / b;
    return log(g);
}

/* lpr_leaves_omega (src/omega.hpp:151-202) with P at the CURRENT tree scale */
static int omega_lik(objective *o, double *lpr)
{
    return po_lpr_leaves(o->qd, o->tree, o->tree_settings, o->codons, o->n_sites, o->leaf_stride, 0, lpr, NULL, NULL);
}

static int omega_requantise(objective *o)
{
    double Q[KK], pi[K];
    po_q_from_omega(o->q_settings, Q, pi);
    po_eig(Q, pi, o->qd);
    return 0;
}

/* minimizer_lpr_leaves_rho, src/omega.hpp:204-218 */
static double obj_rho(double x, objective *o)
{
    o->x = x;
    o->tree_settings = x;
    if (omega_lik(o, &o->lpr)) {
        o->failed = 1;
        trace_add(2, x, o, NAN, 1);
        return NAN;
    }
    trace_add(2, x, o, o->lpr, 0);
    o->lpr += po_lpr_rho(x);
    if (getenv("PO_TRACE")) fprintf(stderr, "TRACE omega rho x %.17g lpr %.17g\n", x, o->lpr);
    return -o->lpr;
}

/* minimizer_lpr_leaves_kappa, src/omega.hpp:220-234 */
static double obj_kappa(double x, objective *o)
{
    o->x = x;
    o->q_settings[0] = x;
    omega_requantise(o);
    if (omega_lik(o, &o->lpr)) {
        o->failed = 1;
        trace_add(3, x, o, NAN, 1);
        return NAN;
    }
    trace_add(3, x, o, o->lpr, 0);
    o->lpr += po_lpr_kappa(x);
    if (getenv("PO_TRACE")) fprintf(stderr, "TRACE omega kappa x %.17g lpr %.17g\n", x, o->lpr);
    return -o->lpr;
}

/* src/run.hpp:57-211 */
int po_run(const po_model *m, int algo, const uint8_t *codons, size_t n_sites, size_t leaf_stride,
           int compute_anc, uint32_t seed, float *phylo, float *anc)
{
    const po_tree *tree = &m->tree;
    po_mt19937 gen;
    po_mt_seed(&gen, seed);
    g_eval_count = 0;
    *phylo = NAN;
    if (anc) *anc = NAN;
    int rc = 0;

    if (algo == PO_OMEGA) {
        /* POLICY: the reference yields NaN for every block without a complete
         * codon because GSL's non-symmetric solver degrades on the uniform-F3x4 matrix; the
         * symmetric solver would return 0.0.  Emit NaN to match the shipped goldens. */
        if (n_sites == 0) return -1;
        po_qdiag *qd = (po_qdiag *)malloc(sizeof(po_qdiag));
        double *P = (double *)malloc(sizeof(double) * KK * (size_t)(tree->n - 1));
        objective o;
        memset(&o, 0, sizeof o);
        o.tree = tree; o.codons = codons; o.n_sites = n_sites; o.leaf_stride = leaf_stride; o.qd = qd;
        /* src/run.hpp:75-102 */
        o.q_settings[0] = 2.5; o.q_settings[1] = 1.0; o.q_settings[2] = 1.0;
        for (int i = 3; i < 12; ++i) o.q_settings[i] = 1.0;
        o.tree_settings = 1.0;
        omega_requantise(&o);
        if (po_make_all_P(qd, tree, 1.0, P)) rc = -1;
        /* update_f3x4, src/run.hpp:105-140 */
        if (!rc) {
            double counts[3][4];
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 4; ++j) counts[i][j] = 1.0;
            for (int l = 0; l < tree->nl; ++l)
                for (size_t s = 0; s < n_sites; ++s) {
                    const uint8_t c = codons[(size_t)l * leaf_stride + s];
                    if (c != PO_GAP) {
                        counts[0][c / 16] += 1;
                        counts[1][(c / 4) % 4] += 1;
                        counts[2][c % 4] += 1;
                    }
                }
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) o.q_settings[3 + 3 * i + j] = counts[i][j] / counts[i][3];
            omega_requantise(&o);
            if (po_make_all_P(qd, tree, 1.0, P)) rc = -1;
        }
        double lpr_H0 = 0.0, lpr_H1 = 0.0;
        if (!rc) {
            for (int i = 0; i < 3 && !o.failed; ++i) { /* src/run.hpp:150-157 */
                o.f = obj_rho;
                max_lik(&o, o.tree_settings, 0.001, 10.0, &gen, &lpr_H0, NULL);
                if (o.failed) break;
                o.f = obj_kappa;
                max_lik(&o, o.q_settings[0], 1.0, 10.0, &gen, &lpr_H0, NULL);
            }
            if (o.failed) rc = -1;
        }
        if (!rc) { /* src/run.hpp:161-176 */
            o.q_settings[1] = 0.2;
            o.q_settings[2] = 0.01;
            omega_requantise(&o);
            /* tree_p14n still holds the last evaluated rho (src/omega.hpp:210) */
            if (po_make_all_P(qd, tree, o.tree_settings, P)) rc = -1;
        }
        if (!rc) {
            for (int i = 0; i < 3 && !o.failed; ++i) {
                o.f = obj_rho;
                max_lik(&o, o.tree_settings, 0.001, 10.0, &gen, &lpr_H1, NULL);
                if (o.failed) break;
                o.f = obj_kappa;
                max_lik(&o, o.q_settings[0], 1.0, 10.0, &gen, &lpr_H1, NULL);
            }
            if (o.failed) rc = -1;
        }
        if (!rc) *phylo = (float)(10.0 * (lpr_H1 - lpr_H0) / log(10.0));
        free(P);
        free(qd);
        return rc;
    }

    po_qdiag *qc = (po_qdiag *)malloc(sizeof(po_qdiag)), *qn = (po_qdiag *)malloc(sizeof(po_qdiag));
    double lpr_c = 0, lpr_nc = 0, anc_c = 0, anc_nc = 0;
    if (ecm_instance(m->S_c, m->pi_c, tree, qc) || ecm_instance(m->S_nc, m->pi_nc, tree, qn)) rc = -1;
    if (!rc && algo == PO_MLE) { /* src/run.hpp:191-195 */
        objective o;
        memset(&o, 0, sizeof o);
        o.tree = tree; o.codons = codons; o.n_sites = n_sites; o.leaf_stride = leaf_stride;
        o.compute_anc = compute_anc; o.f = obj_tree_scale;
        o.qd = qc;
        g_trace_kind = 0;
        max_lik(&o, 1.0, 1e-2, 10.0, &gen, &lpr_c, &anc_c);
        if (!o.failed) {
            o.qd = qn;
            g_trace_kind = 1;
            max_lik(&o, 1.0, 1e-2, 10.0, &gen, &lpr_nc, &anc_nc);
        }
        if (o.failed) rc = -1;
    } else if (!rc) { /* FIXED, src/run.hpp:196-204 */
        if (po_lpr_leaves(qc, tree, 1.0, codons, n_sites, leaf_stride, compute_anc, &lpr_c, &anc_c, NULL)) rc = -1;
        if (!rc && po_lpr_leaves(qn, tree, 1.0, codons, n_sites, leaf_stride, compute_anc, &lpr_nc, &anc_nc, NULL)) rc = -1;
    }
    if (!rc) {
        *phylo = (float)(10.0 * (lpr_c - lpr_nc) / log(10.0));
        if (anc) *anc = (float)(10.0 * (anc_c - anc_nc) / log(10.0));
    }
    free(qc);
    free(qn);
    return rc;
}

/* po_run that also records every objective evaluation (kind 0/1: MLE tree scale under the coding / non-coding ECM,
 * 2: OMEGA rho, 3: OMEGA kappa; x, the model's q_settings and tree scale DURING the evaluation, and the
 * log-likelihood WITHOUT the prior term).  *n_out = number of evaluations (may exceed cap; only cap are stored). */
int po_run_trace(const po_model *m, int algo, const uint8_t *codons, size_t n_sites, size_t leaf_stride, int compute_anc,
                 uint32_t seed, float *phylo, float *anc, po_trace_entry *out, int cap, int *n_out)
{
    g_trace = out;
    g_trace_cap = cap;
    g_trace_n = 0;
    const int rc = po_run(m, algo, codons, n_sites, leaf_stride, compute_anc, seed, phylo, anc);
    *n_out = g_trace_n;
    g_trace = NULL;
    return rc;
}

/* score-msa's block loop (src/phylocsf++score_msa.hpp:94-128): every block alone, seeded alike */
int po_run_blocks(const po_model *m, int algo, const uint8_t *codons, const uint64_t *off, size_t n_blocks,
                  size_t leaf_stride, int compute_anc, uint32_t seed, float *phylo, float *anc, int threads)
{
    int nfail = 0;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1) reduction(+ : nfail)
#endif
    for (long b = 0; b < (long)n_blocks; ++b) {
        float a = NAN;
        if (po_run(m, algo, codons + off[b], (size_t)(off[b + 1] - off[b]), leaf_stride, compute_anc, seed,
                   &phylo[b], &a))
            ++nfail;
        if (anc) anc[b] = a;
    }
    (void)threads;
    return nfail;
}

/* ------------------------------------------------------------------------------------------
 * branch-length score (src/additional_scores.hpp:5-84) */
static int po_overlap_size(const po_tree *t, int node, const unsigned char *in_set)
{
    if (node < t->nl) return in_set[node] ? 1 : 0;
    return po_overlap_size(t, t->child1[node], in_set) + po_overlap_size(t, t->child2[node], in_set);
}

/* newick_sum_branch_lengths (src/additional_scores.hpp:5-41) */
static double po_sum_branch_lengths(const po_tree *t, int node, const unsigned char *in_set, int arrived_at_lca, int overlap_parent)
{
    if (node < t->nl) return t->branch_len_d[node];
    if (overlap_parent == -1) overlap_parent = po_overlap_size(t, node, in_set);
    const int overlap_left = po_overlap_size(t, t->child1[node], in_set);
    const int overlap_right = overlap_parent - overlap_left;
    double bl = 0.0;
    if (arrived_at_lca) bl = t->branch_len_d[node];
    if (overlap_left > 0 && overlap_right > 0) arrived_at_lca = 1;
    if (overlap_left > 0) bl += po_sum_branch_lengths(t, t->child1[node], in_set, arrived_at_lca, overlap_left);
    if (overlap_right > 0) bl += po_sum_branch_lengths(t, t->child2[node], in_set, arrived_at_lca, overlap_right);
    return bl;
}

/* compute_bls_score (src/additional_scores.hpp:44-84) */
double po_bls(const po_tree *t, const char *nucl, size_t row_stride, size_t n_cols, double *per_base)
{
    unsigned char in_set[PO_MAX_NODES];
    for (int l = 0; l < t->nl; ++l) in_set[l] = 1;
    const double all = po_sum_branch_lengths(t, t->n - 1, in_set, 0, -1);
    double total = 0.0;
    for (size_t i = 0; i < n_cols; ++i) {
        int cnt = 0;
        for (int l = 0; l < t->nl; ++l) {
            const char ch = nucl[(size_t)l * row_stride + i];
            int id;
            switch (ch) { /* get_dna_id, src/translation.hpp:29-88 */
                case 'A': case 'a': id = 0; break;
                case 'C': case 'c': id = 1; break;
                case 'G': case 'g': id = 2; break;
                case 'T': case 't': id = 3; break;
                case 'N': case 'n': case '-': case '.': id = 4; break;
                default: return -1.0;
            }
            in_set[l] = id <= 3;
            cnt += id <= 3;
        }
        if (cnt >= 2) {
            const double bl = po_sum_branch_lengths(t, t->n - 1, in_set, 0, -1);
            total += bl;
            if (per_base) per_base[i] = bl / all;
        } else if (per_base) {
            per_base[i] = 0.0;
        }
    }
    return total / (all * (double)n_cols);
}

/* ------------------------------------------------------------------------------------------
 * reference-structured CPU baseline for bench.py
 * ---------------------------------------------------------------------------------------- */
int po_baseline_run_tracks(const po_model *m, const uint8_t *codons, const uint64_t *off, size_t n_aln,
                           size_t leaf_stride, double *out, int threads)
{
    int rc = 0;
#ifdef _OPENMP
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel for num_threads(threads) schedule(dynamic, 1) reduction(| : rc)
#endif
    for (long a = 0; a < (long)n_aln; ++a) {
        const size_t s0 = off[a], s1 = off[a + 1];
        if (po_run_tracks(m, codons + s0, s1 - s0, leaf_stride, out + s0)) rc |= 1;
    }
    (void)threads;
    return rc ? -1 : 0;
}
