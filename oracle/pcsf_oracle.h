/*
 * pcsf_oracle.h -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C, single-threaded-per-call CPU restatement of the PhyloCSF++ scoring hot path
 * (cpockrandt/PhyloCSFpp v1.2.0).  It exists only so that tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg can check / time the HIP path against an independent
 * implementation of the reference's algorithm.  Nothing under phylocsfpp_amd/ may include,
 * link or call it.
 *
 * Parity status: PINNED against the reference's own golden outputs (tests/golden/, generated
 * by tests/golden/make_fixtures.py from /root/reference/test): the 7 raw/power wigs of the
 * build-tracks test and the FIXED/MLE .scores files of the score-msa test -- see
 * tests/test_oracle_golden.py.  The reference's likelihood core cannot be compiled in this image
 * (GSL and libBigWig are absent and un-vendored); oracle/_ref holds only ref_harness, built from
 * the reference's GSL-free headers (oracle/ref_harness.cpp: MAF reader, frames, branch-length
 * score, trees, HMM), which pins the code either side of this restatement.
 *
 * GSL (un-pinned version, CMakeLists.txt:43) is replaced as follows:
 *   gsl_eigen_nonsymmv + complex LU inverse  -> cyclic Jacobi on the pi-symmetrised rate matrix
 *   gsl_blas_dgemm / ddot                    -> plain loops in the reference BLAS loop order
 *   gsl_min_fminimizer_brent                 -> restatement of GSL's published min/brent.c
 *   gsl_ran_gamma_pdf, gsl_sf_exp            -> closed forms with libm
 *
 * Every function cites the reference file:line it follows.
 */
#ifndef PCSF_ORACLE_H
#define PCSF_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PO_K 64          /* codon states, stop codons included (src/ecm.hpp:17-18) */
#define PO_GAP 64        /* "marginalise" leaf code (src/translation.hpp:80-88) */
#define PO_MAX_NODES 512

enum { PO_MLE = 0, PO_FIXED = 1, PO_OMEGA = 2 }; /* src/run.hpp:16-21 */

/* Tree in newick_flatten order (src/newick.hpp:170-229): ids 0..nl-1 leaves (left-to-right),
 * nl..n-1 internal nodes in post-order, root = n-1.  branch_len is FLOAT like newick_elem. */
typedef struct {
    int32_t n, nl;
    int16_t child1[PO_MAX_NODES], child2[PO_MAX_NODES];
    int16_t parent[PO_MAX_NODES], sibling[PO_MAX_NODES];
    float branch_len[PO_MAX_NODES];
    double branch_len_d[PO_MAX_NODES]; /* newick_node::branch_length (double), used by BLS */
    char label[PO_MAX_NODES][64];      /* lower-cased (src/newick.hpp:37) */
} po_tree;

typedef struct {
    po_tree tree;
    double S_c[PO_K * PO_K], pi_c[PO_K];   /* coding ECM: symmetric exchangeabilities + freqs */
    double S_nc[PO_K * PO_K], pi_nc[PO_K]; /* non-coding ECM */
} po_model;

/* --- model input (src/ecm.hpp:20-67, src/newick.hpp:31-229) --- */
int po_ecm_open(const char *path, double *S /*[64*64]*/, double *pi /*[64]*/);
int po_tree_parse(const char *newick_text, po_tree *t);
int po_tree_open(const char *path, po_tree *t);
int po_model_open(const char *prefix, po_model *m); /* <prefix>_coding.ECM, _noncoding.ECM, .nh */

/* --- rate matrix, eigen-system, P(t) (src/instance.hpp:309-434, 449-646, 648-685) --- */
typedef struct {
    double Q[PO_K * PO_K];
    double lam[PO_K];
    double Sr[PO_K * PO_K];   /* right eigenvectors in columns (r_s)  */
    double Sinv[PO_K * PO_K]; /* left eigenvectors in rows (r_s2) = Sr^-1 */
    double prior[PO_K];       /* equilibrium, src/fixed_lik.hpp:323-346 */
} po_qdiag;

void po_q_from_ecm(const double *S, const double *pi, double *Q);
void po_q_from_omega(const double *q_settings /*[12]*/, double *Q, double *pi_out /*[64]*/);
void po_eig(const double *Q, const double *pi_sym, po_qdiag *qd);
/* P = Sr diag(exp(lam t)) Sinv with the reference's clamp / diagonal fix / validation.
 * returns 0, or -1 where the reference throws std::runtime_error (src/instance.hpp:618,635) */
int po_make_P(const po_qdiag *qd, double t, double *P /*[64*64], row = parent state*/);
/* all n-1 branch matrices at tree scale x: t_b = (double)(float)((double)bl_b * x)
 * (src/instance.hpp:299-307,498) */
int po_make_all_P(const po_qdiag *qd, const po_tree *tree, double x, double *P /*[(n-1)*4096]*/);

/* --- Felsenstein pruning (src/fixed_lik.hpp:86-164) ---
 * codons: leaf-major [nl][leaf_stride]; returns z for one site; alpha_root optional [64] */
double po_prune_site(const po_tree *tree, const double *P, const double *prior,
                     const uint8_t *codons, size_t leaf_stride, size_t site, double *alpha_root);

/* lpr_leaves (src/fixed_lik.hpp:362-449) at tree scale x with a ready eigen-system.
 * per_codon may be NULL.  returns 0 / -1 (P validation failure) */
int po_lpr_leaves(const po_qdiag *qd, const po_tree *tree, double x, const uint8_t *codons,
                  size_t n_sites, size_t leaf_stride, int compute_anc, double *lpr,
                  double *elpr_anc, double *per_codon);

/* run_tracks (src/run.hpp:35-55): per-codon 10*(ln z_c - ln z_nc)/ln 10 */
int po_run_tracks(const po_model *m, const uint8_t *codons, size_t n_sites, size_t leaf_stride,
                  double *out_decibans);

/* run (src/run.hpp:57-211): block score for FIXED / MLE / OMEGA; seed = 42 in score-msa
 * (src/phylocsf++score_msa.hpp:115).  On model failure both outputs are NaN and -1 is returned
 * (score-msa swallows the exception, src/phylocsf++score_msa.hpp:124-127). */
int po_run(const po_model *m, int algo, const uint8_t *codons, size_t n_sites,
           size_t leaf_stride, int compute_anc, uint32_t seed, float *phylo, float *anc);

/* po_run plus a record of every objective evaluation in order, for replaying the optimiser's path elsewhere:
 * kind 0/1 = MLE tree scale under the coding / non-coding ECM, 2 = OMEGA rho, 3 = OMEGA kappa; x = the point;
 * q_settings / tree_scale = the model DURING the evaluation; lpr = the log-likelihood WITHOUT the prior term. */
typedef struct {
    int32_t kind, failed;
    double x, tree_scale, q_settings[12], lpr;
} po_trace_entry;
int po_run_trace(const po_model *m, int algo, const uint8_t *codons, size_t n_sites, size_t leaf_stride,
                 int compute_anc, uint32_t seed, float *phylo, float *anc, po_trace_entry *out, int cap, int *n_out);

/* batch of independent blocks (CSR offsets into the leaf-major codon matrix), OpenMP over
 * blocks; returns the number of blocks that failed (NaN rows). threads<=0: all cores */
int po_run_blocks(const po_model *m, int algo, const uint8_t *codons, const uint64_t *site_offsets,
                  size_t n_blocks, size_t leaf_stride, int compute_anc, uint32_t seed, float *phylo,
                  float *anc, int threads);

/* control experiment only (pcsf_oracle.c, "CONTROL EXPERIMENT"): other summation orders; 0 = the reference's.  Not thread-safe: set
 * it before any run. */
void po_set_perturbation(int bits);

/* number of objective evaluations done by the last po_run on this thread (diagnostics) */
long po_last_eval_count(void);

/* --- small pieces exposed for unit tests --- */
/* max_lik_lpr_leaves (fit_find_init + GSL Brent) on a caller-supplied objective to be maximised */
int po_debug_max_lik(double (*fn)(double, void *), void *user, double init, double lo, double hi, uint32_t seed,
                     int skip_draws, double *xs_out, int cap, int *n_evals, double *last_lpr);
typedef struct { uint32_t mt[624]; int idx; } po_mt19937;
void po_mt_seed(po_mt19937 *g, uint32_t seed);
uint32_t po_mt_next(po_mt19937 *g);
double po_uniform_real(po_mt19937 *g, double a, double b); /* libstdc++ uniform_real_distribution */
double po_lpr_rho(double rho);     /* src/omega.hpp:130-141 */
double po_lpr_kappa(double kappa); /* src/omega.hpp:143-149 */

/* --- branch-length score (src/additional_scores.hpp:5-84) ---
 * compute_bls_score<true>: per column of the leaf-major ASCII rows nucl[nl][row_stride], S = the species with
 * one of ACGTacgt; per_base[col] = bl(S) / bl(all species), 0 when |S| < 2 (per_base may be NULL); returns the
 * block score sum bl(S) / (bl(all) * n_cols).  bl is newick_sum_branch_lengths, as a recursion over the
 * flattened tree with the reference's order of additions.  -1.0 on a character outside ACGTNacgtn-. */
double po_bls(const po_tree *t, const char *nucl, size_t row_stride, size_t n_cols, double *per_base);

/* reference-structured CPU baseline used by bench.py (cpu_baseline.kind = "port"):
 * scores n_aln independent alignments exactly like the OpenMP loop of build-tracks does --
 * the model is re-instantiated for every alignment x model (src/run.hpp:38-49,
 * src/fixed_lik.hpp:370-372) and every codon goes through the scalar pruning loop.
 * site_offsets[n_aln+1] delimit alignments in the leaf-major codon matrix. threads<=0: all. */
int po_baseline_run_tracks(const po_model *m, const uint8_t *codons, const uint64_t *site_offsets,
                           size_t n_aln, size_t leaf_stride, double *out, int threads);

#ifdef __cplusplus
}
#endif
#endif
