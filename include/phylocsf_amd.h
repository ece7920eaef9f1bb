/*
 * phylocsf_amd.h -- C-ABI of the MI355X (gfx950) PhyloCSF scoring core.
 *
 * Drop-in boundary: the reference (cpockrandt/PhyloCSFpp v1.2.0) has no FFI; its scoring core is
 * reached through two free functions called from inside the OpenMP job loops:
 *
 *   void run_tracks(Data&, const Model&, const alignment_t&, std::vector<double>&)   src/run.hpp:35
 *        callers: src/phylocsf++build_tracks.hpp:172, src/phylocsf++score_msa.hpp:149
 *   std::tuple<float,float> run(Data&, const Model&, const alignment_t&, algorithm_t,
 *                               std::mt19937&, bool)                                src/run.hpp:57
 *        caller:  src/phylocsf++score_msa.hpp:116
 *
 * The entry points below replace exactly those two calls (batched, because one call per
 * alignment x frame is too fine for a GPU) plus the model set-up they repeat on every call
 * (PhyloCSFModel_make, src/instance.hpp:687-712).  INTEGRATION.md shows the binding a
 * maintainer of the reference would add.
 *
 * Conventions: plain C types only; every function returns 0 on success or a negative
 * pcsf_status; no exception crosses the boundary; the caller owns every buffer; a context is
 * bound to one GPU and may be used by one host thread at a time; outputs are in input order.
 * There is NO CPU fallback: if no gfx950 device is present pcsf_create fails with
 * PCSF_ERR_NO_DEVICE.
 */
#ifndef PHYLOCSF_AMD_H
#define PHYLOCSF_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCSF_STATES 64     /* codon states incl. stops (src/ecm.hpp:17-18) */
#define PCSF_GAP_CODON 64  /* "marginalise" leaf code (src/translation.hpp:80-88) */

typedef enum {
    PCSF_OK = 0,
    PCSF_ERR_INVALID_ARG = -1,
    PCSF_ERR_NO_DEVICE = -2,   /* no HIP device / not gfx950 / extension not usable */
    PCSF_ERR_HIP = -3,         /* a HIP runtime call failed; see pcsf_last_error */
    PCSF_ERR_MODEL = -4,       /* P(t) failed the reference's validation (src/instance.hpp:618,635) */
    PCSF_ERR_UNSUPPORTED = -5
} pcsf_status;

/* algorithm_t, src/run.hpp:16-21 (same numeric values) */
typedef enum { PCSF_MLE = 0, PCSF_FIXED = 1, PCSF_OMEGA = 2 } pcsf_algo;

/* What the path reads from `Model` (src/models.hpp:1742-1755):
 *   phylo_array  -> the flattened tree exactly as newick_flatten numbers it
 *                   (src/newick.hpp:170-229): ids 0..nl-1 leaves left-to-right, nl..n-1 internal
 *                   nodes in post-order, root = n-1; child ids -1 for leaves; FLOAT branch lengths
 *   c_model / nc_model -> empirical_codon_model::matrix (row-major 64x64, symmetric, zero
 *                   diagonal) and ::codon_freq (src/ecm.hpp:15-19) */
typedef struct {
    int32_t n_nodes;
    const int16_t *child1;     /* [n_nodes] */
    const int16_t *child2;     /* [n_nodes] */
    const float *branch_len;   /* [n_nodes] (root entry ignored) */
    const double *S_coding;    /* [64*64] */
    const double *pi_coding;   /* [64] */
    const double *S_noncoding; /* [64*64] */
    const double *pi_noncoding;/* [64] */
} pcsf_model_desc;

typedef struct pcsf_ctx pcsf_ctx;

/* ---- load_model (src/models.hpp:1757-1857): the model files behind the arrays above ----------
 * <prefix>_coding.ECM / <prefix>_noncoding.ECM (get_ecm: 63 lower-triangle rows, blank line, 64 frequencies;
 * src/ecm.hpp:20-67) and <prefix>.nh (newick_parse + newick_flatten, src/newick.hpp:31-229).
 * prefix: like the reference's model_name_or_path (src/models.hpp:1766-1783) -- a prefix that names existing files, or, failing
 * that, the bare NAME of a model that ships with the library ("53birds", "58mammals", "100vertebrates": the reference compiles its
 * models in, src/models.hpp:1443-1455; here they are files in `models/` next to the library, $PCSF_MODEL_DIR first).
 * alias_file: "<tree label>\t<other names...>" per line, stands in for the name table compiled into the
 * reference (src/models.hpp:1240-1706); NULL = species_aliases.tsv next to the model files if present, "" = none.
 * species_csv: --species (tree reduced like newick_reduce, src/newick.hpp:286-363), NULL/"" = whole tree.
 * mapping_file: --mapping (update_sequence_name_mapping, src/models.hpp:1709-1740), NULL/"" = none.
 * Host-only (no GPU needed).  On failure pcsf_last_error(NULL) holds the message. */
typedef struct pcsf_model pcsf_model;
int pcsf_model_open(pcsf_model **out, const char *prefix, const char *alias_file, const char *species_csv,
                    const char *mapping_file);
void pcsf_model_close(pcsf_model *model);
/* fills *desc with pointers INTO the model (valid until pcsf_model_close): what pcsf_create takes */
int pcsf_model_describe(const pcsf_model *model, pcsf_model_desc *desc);
/* [n_nodes] newick_node::branch_length in double, for pcsf_set_bls_branch_lengths */
const double *pcsf_model_branch_len_double(const pcsf_model *model);
/* [n_nodes] parent ids (-1 for the root) */
const int16_t *pcsf_model_parents(const pcsf_model *model);
/* lower-cased label of a node ("" for internal nodes); NULL if node is out of range */
const char *pcsf_model_label(const pcsf_model *model, int32_t node);
/* leaf index of a species name as it appears in alignments (tree labels and their other names), -1 if unknown */
int32_t pcsf_model_leaf_of(const pcsf_model *model, const char *species_name);

/* Replaces PhyloCSFModel_make x2 (src/run.hpp:38-39): builds Q, its eigen-system and the
 * 2*(n-1) branch matrices once and keeps them resident in HBM.  device_id: HIP ordinal.
 * Also made here, once: the subtree tables of tracks mode -- for every cherry of the tree (and every node over a cherry and a leaf) the
 * message it sends its parent, per codon pair (triple), computed by the pruning kernel's own product: per-codon scoring then runs one
 * matrix product less per such node and model with the same bits (58mammals: 30 instead of 56; 2.6 GB of device memory).  Environment,
 * read here: PCSF_CHERRY_TABLES=0 (none), PCSF_TABLE_LEAVES=2|3|4 (default 3), PCSF_TABLE_BYTES (default 8 GiB); pcsf_tracks_program_info
 * reports what a context uses. */
int pcsf_create(pcsf_ctx **out, const pcsf_model_desc *model, int device_id);
void pcsf_destroy(pcsf_ctx *ctx);
/* message of the last failure on this context (or of the last failed pcsf_create if ctx==NULL) */
const char *pcsf_last_error(const pcsf_ctx *ctx);
int pcsf_n_leaves(const pcsf_ctx *ctx);

/* ---- run_tracks (src/run.hpp:35-55): per-codon 10*(ln z_coding - ln z_noncoding)/ln 10 ------
 * codons: the reference's alignment_t::peptides, leaf-major uint8 [n_leaves][leaf_stride],
 * values 0..63 or 64 (= marginalise).  Sites of any number of alignments/frames may be
 * concatenated into one call; every site is independent.  out: [n_sites] fp64. */
int pcsf_score_sites_fixed(pcsf_ctx *ctx, const uint8_t *codons, size_t n_sites, size_t leaf_stride,
                           double *out_decibans);

/* Same, but codons/out are DEVICE pointers on ctx's GPU and the work is enqueued on
 * `hip_stream` (a hipStream_t, NULL = default stream) without synchronising.
 * out_lnz may be NULL, else device [2][n_sites]: ln z under the coding (row 0) and the
 * non-coding (row 1) model (the lpr_per_codon vectors of src/fixed_lik.hpp:431-433). */
int pcsf_score_sites_fixed_dev(pcsf_ctx *ctx, const uint8_t *d_codons, size_t n_sites, size_t leaf_stride,
                               double *d_out_decibans, double *d_out_lnz, void *hip_stream);

/* ---- build-tracks front end: the six frames of whole alignments ------------------------------
 * Replaces, per alignment, the loop of src/phylocsf++build_tracks.hpp:162-227: for strand in {+,-}, frame 1..3:
 * alignment_t::update_seqs (frame offset + translation, src/parallel_file_reader.hpp:61-113), run_tracks,
 * reversal of the '-' strand scores, reverse complement between the strands.
 * nucleotides: the rows of alignment_t::seqs (reference gaps already removed), ASCII, leaf-major
 * [n_leaves][row_stride]; alignment k occupies columns [col_offset, col_offset+len) of every row, start_pos is
 * its 1-based reference position and chrom_len the reference chromosome length (alignment_t::start_pos /
 * ::chrom_len).  N n - . make a codon a gap; any other character outside ACGTacgt fails the call ("unexpected character":
 * the reference exits on it, src/translation.hpp:29-88) -- in this entry point for the characters a codon touches.
 * Output: for alignment k and frame index f = 0..5 (+1,+2,+3,-1,-2,-3) n_codons[6k+f] scores at
 * out_scores[out_offsets[6k+f] ...] in FORWARD genome order; skip_bases (may be NULL) = alignment_t::skip_bases. */
typedef struct {
    uint64_t col_offset;
    uint32_t len;
    uint64_t start_pos;
    uint64_t chrom_len;
} pcsf_alignment_desc;
int pcsf_score_alignments_6frames(pcsf_ctx *ctx, const char *nucleotides, size_t row_stride,
                                  const pcsf_alignment_desc *alignments, size_t n_alignments, double *out_scores,
                                  size_t out_capacity, uint64_t *out_offsets, uint32_t *n_codons, uint32_t *skip_bases);

/* Optional: gets the context ready for six-frame calls of up to max_columns alignment columns in total (device scratch
 * reserved, kernels run once on a dummy alignment so that their code is loaded).  A short run -- one 1 Mb file is three
 * calls -- otherwise pays for both inside its first call.  Never changes a result. */
int pcsf_prepare_tracks(pcsf_ctx *ctx, size_t max_columns);

/* ---- branch-length score (compute_bls_score, src/additional_scores.hpp:44-84) ---------------------
 * Per alignment column: the branch lengths of the subtree spanned by the species that have one of
 * ACGTacgt there (newick_sum_branch_lengths, src/additional_scores.hpp:5-41: own branch, then left, then
 * right subtree, in double) over the same sum for all species; 0 when fewer than two species have a base.
 * The reference works on newick_node's DOUBLE branch lengths (the likelihood uses the float ones), so they
 * are handed over separately: branch_len[n_nodes] in the node order of pcsf_model_desc.  Up to 128 species.
 * A character outside ACGTNacgtn-. fails the call ("unexpected character", the reference exits on it,
 * src/translation.hpp:29-88). */
int pcsf_set_bls_branch_lengths(pcsf_ctx *ctx, const double *branch_len);
/* out_bls[n_columns] for the leaf-major rows nucleotides[n_leaves][row_stride] */
int pcsf_branch_length_scores(pcsf_ctx *ctx, const char *nucleotides, size_t row_stride, size_t n_columns,
                              double *out_bls);
/* pcsf_score_alignments_6frames plus, from the same uploaded rows, out_bls[max(col_offset + len)]: the per-base
 * branch-length scores build-tracks thresholds the codons with and writes to PhyloCSFpower.wig
 * (src/phylocsf++build_tracks.hpp:139-158,201-205); columns outside every alignment get 0.  out_bls may be NULL. */
int pcsf_score_alignments_6frames_bls(pcsf_ctx *ctx, const char *nucleotides, size_t row_stride,
                                      const pcsf_alignment_desc *alignments, size_t n_alignments, double *out_scores,
                                      size_t out_capacity, uint64_t *out_offsets, uint32_t *n_codons,
                                      uint32_t *skip_bases, double *out_bls);
/* The same call for a caller that only WRITES the results: the scores and the per-base branch-length scores stay on the device and
 * the text of build-tracks' seven wig files is formatted there (csrc/wig.hip) -- the output loop of
 * src/phylocsf++build_tracks.hpp:139-156 (PhyloCSFpower.wig: every codon of frame 3 of the + strand, "%.4f" trimmed, one header per
 * alignment) and :183-217 (PhyloCSFRaw+1..-3: a codon whose three per-base scores sum to less than 3 x power_threshold is left out,
 * :201-205; a "fixedStep chrom=... start=... step=3 span=3" line wherever a run of written codons starts, :207-210; "%.3f" trimmed,
 * src/common.hpp:48-68).  chrom_names[n_alignments]: the alignments' chromosome names.  out_bytes[7]: bytes of text per file of THIS
 * call (0: power track, 1..6: the frames in the order +1 +2 +3 -1 -2 -3); want_power / want_raw = 0 leaves the file(s) empty.
 * text[text_capacity] (may be NULL / 0): receives the seven files' text back to back -- file f at offset out_bytes[0] + .. +
 * out_bytes[f - 1] -- IF the sum of out_bytes is at most text_capacity; the copy is queued behind the kernels before the sizes are
 * known, so give it pcsf_wig_text_estimate() bytes of page-locked memory (pcsf_host_alloc_on).  A caller whose buffer was too small
 * (or who passed none) fetches the text with pcsf_fetch_wig_text.  The text is what the host formatter writes, byte for byte
 * (tests/test_gpu_wig.py, tests/test_gpu_cli.py).  PCSF_ERR_UNSUPPORTED: a finite |value| >= 9e14 turned up (never a score): format
 * this batch from pcsf_score_alignments_6frames_bls instead.  Needs pcsf_set_bls_branch_lengths. */
int pcsf_score_alignments_6frames_wig(pcsf_ctx *ctx, const char *nucleotides, size_t row_stride,
                                      const pcsf_alignment_desc *alignments, size_t n_alignments, const char *const *chrom_names,
                                      float power_threshold, int want_power, int want_raw, char *text, size_t text_capacity,
                                      uint64_t *out_bytes, size_t *n_codon_frames);
/* bytes of text buffer that hold the text of a call over n_columns alignment columns in n_alignments alignments unless its lines are
 * unusually long (12 bytes per line where a line is ~8; no device needed) */
size_t pcsf_wig_text_estimate(size_t n_columns, size_t n_alignments);
/* copies the text of the last pcsf_score_alignments_6frames_wig call of this context: dst[f] has room for out_bytes[f] (page-locked
 * memory from pcsf_host_alloc_on keeps the copy off the host threads); a file without text may have dst[f] = NULL */
int pcsf_fetch_wig_text(pcsf_ctx *ctx, char *const *dst);

/* ---- build-tracks as one process per GPU: the output wig stream gathered over RCCL --------------------------
 * The reference runs its byte-range jobs in ONE parallel loop, every job writing files of its own, and then merges the jobs' files in
 * job order (src/phylocsf++build_tracks.hpp:88 "#pragma omp parallel for", :93-124 "PhyloCSFpower.wig.<job_id>" ...,
 * merge_job_output_files :27-53, called :246-259).  With one process ("rank") per GPU the jobs' ranges of the file go to the ranks in
 * rank order, a rank's text stays in its GPU's memory (pcsf_wig_retain_*), and ONE grouped send/receive over RCCL (xGMI between the
 * GPUs of a node) brings every rank's text to the root's GPU, where the root copies it down file by file, rank by rank -- the merge.
 * librccl.so is loaded when pcsf_comm_unique_id / pcsf_comm_create is first called (a process that never calls them does not map it).
 *
 * pcsf_comm_unique_id: rank 0 makes the 128-byte identifier (ncclGetUniqueId) and hands it to the other processes out of band (a file,
 * a pipe).  pcsf_comm_create: collective over the `world` processes, each with a context on a GPU of its own (ncclCommInitRank on the
 * context's device).  One communicator per context; pcsf_destroy releases it. */
#define PCSF_COMM_ID_BYTES 128
int pcsf_comm_unique_id(void *id);
int pcsf_comm_create(pcsf_ctx *ctx, int rank, int world, const void *id);
int pcsf_comm_destroy(pcsf_ctx *ctx);
/* Keeps the text of this context's LAST pcsf_score_alignments_6frames_wig call on the device (a device-to-device copy into the
 * context's arena) under the caller's key `order`; pcsf_wig_retain_host adds text formatted on the host the same way (text[f] may be
 * NULL where bytes[f] is 0).  The gather sends a context's kept texts in ascending `order`, whatever order they were kept in.
 * pcsf_wig_retained_bytes: bytes kept per file so far.  pcsf_wig_retain_reset drops them (the gather does that itself). */
int pcsf_wig_retain_last(pcsf_ctx *ctx, uint64_t order);
int pcsf_wig_retain_host(pcsf_ctx *ctx, uint64_t order, const char *const *text, const uint64_t *bytes);
int pcsf_wig_retained_bytes(pcsf_ctx *ctx, uint64_t *bytes /* [7] */);
int pcsf_wig_retain_reset(pcsf_ctx *ctx);
/* Collective.  Every rank but the root -- and the root too if include_root is set (a rehearsal on one GPU: the root's text then
 * travels through RCCL to itself) -- sends what it kept: its seven files' texts, each file's kept pieces joined in `order`.  An
 * all-gather of the 7 sizes per rank first (bytes[world][7] is filled on every rank), then one ncclSend per sender and one ncclRecv per
 * sender on the root inside one group.  Afterwards the root's device holds every sender's text and pcsf_comm_fetch_gathered copies the
 * text of `file` sent by `rank` to dst (bytes[rank][file] bytes; page-locked memory keeps the copy off the host threads).  The kept
 * texts of the calling context are dropped. */
int pcsf_comm_gather_wig(pcsf_ctx *ctx, int root, int include_root, uint64_t *bytes);
int pcsf_comm_fetch_gathered(pcsf_ctx *ctx, int rank, int file, char *dst);

/* ---- the smoothing pass of build-tracks (--output-phylo / --output-regions) --------------------------------
 * hmm::state_posterior_probabilities and the forward half of get_best_path_by_viterbi (src/create_tracks.hpp:25-159,
 * called per run of a raw track by process_scores, src/create_tracks.hpp:226-314) for a batch of runs at once.
 * init[4] / trans[4][4] (row-major, trans[from][to]): get_coding_hmm (src/create_tracks.hpp:163-200), state 0 = coding.
 * emissions[n]: the coding state's emission at every position, pow(10, score / 10) as the caller's libm gives it (the
 * other states emit 1); run r owns positions [run_offsets[r], run_offsets[r + 1]), n = run_offsets[n_runs].
 * out_posterior[n]: posterior probability of the coding state (the reference's scaled forward-backward, same operation
 * order: the values equal a sequential evaluation bit for bit -- csrc/hmm.hip says how).  Optionally the Viterbi
 * recursion: out_back[n] holds, for every position but a run's first, the best predecessor of each of the four states
 * (2 bits per state, state s in bits 2s..2s+1) and out_last[n_runs][4] the rescaled probabilities at each run's last
 * position -- the caller picks the final state (first maximum) and walks back.  out_back and out_last are both NULL or
 * both given.  stats (may be NULL): how many chunks the runs were cut into and how many had to be evaluated again. */
typedef struct pcsf_hmm_stats {
    uint32_t chunks;
    uint32_t forward_reruns, forward_passes, backward_reruns, backward_passes, viterbi_reruns, viterbi_passes;
} pcsf_hmm_stats;
int pcsf_hmm_smooth_runs(pcsf_ctx *ctx, const double *init, const double *trans, const double *emissions,
                         const uint64_t *run_offsets, size_t n_runs, double *out_posterior, uint8_t *out_back,
                         double *out_last, pcsf_hmm_stats *stats);
/* The same for a caller that only PRINTS a monotone function of the posterior, as process_scores does (log-odds as
 * "%.3f", src/create_tracks.hpp:226-260, src/common.hpp:48-68): instead of the posterior, out_bins[n] = how many of
 * the ascending thresholds[n_thresholds] (fewer than 65,534) are <= the posterior -- with the thresholds at the values
 * where the printed text changes that IS the text, in 2 bytes per position and without a log10 per value on the host.
 * A NaN posterior has no bin (out_bins 0xffff, or 0xfffe when its sign bit is set): position and value are appended to exc_index / exc_value (up to
 * exc_capacity entries; *n_exceptions counts all of them). */
int pcsf_hmm_smooth_runs_binned(pcsf_ctx *ctx, const double *init, const double *trans, const double *emissions,
                                const uint64_t *run_offsets, size_t n_runs, const double *thresholds,
                                size_t n_thresholds, uint16_t *out_bins, uint64_t *exc_index, double *exc_value,
                                size_t exc_capacity, size_t *n_exceptions, uint8_t *out_back, double *out_last,
                                pcsf_hmm_stats *stats);
/* The two calls above keep their device buffers (~60 bytes per value of the largest call so far) on the context for the next call;
 * this frees them (a caller that has finished its smoothing pass).  They are also freed by pcsf_destroy. */
int pcsf_hmm_release(pcsf_ctx *ctx);

/* ---- run (src/run.hpp:57-211): whole-block scores -----------------------------------------
 * A batch of blocks in CSR form: block b owns sites [site_offsets[b], site_offsets[b+1]) of the
 * leaf-major codon matrix.  algo FIXED / MLE / OMEGA; seed is what score-msa passes to
 * gen.seed() per block (42, src/phylocsf++score_msa.hpp:115).  out_phylo/out_anc: [n_blocks]
 * float like the reference's std::tuple<float,float>; a block whose model fails validation gets
 * NaN in both (score-msa swallows the exception, src/phylocsf++score_msa.hpp:124-127); out_anc
 * may be NULL; it is NaN in OMEGA mode (src/run.hpp:181).  When the call fails (return != PCSF_OK)
 * the contents of out_phylo / out_anc are undefined: long lists are processed in groups, and the
 * groups that had finished before the failure have written their rows. */
int pcsf_score_blocks(pcsf_ctx *ctx, int algo, const uint8_t *codons, const uint64_t *site_offsets,
                      size_t n_blocks, size_t leaf_stride, int compute_anc, uint32_t seed,
                      float *out_phylo, float *out_anc);
/* Which kernel prunes the blocks of MLE / OMEGA calls on this context.  The engine has two -- prune_tiles for dense blocks, prune_factored
 * for the short sparse blocks of real alignments -- whose sums round differently (objective values differ by ~1e-13, OMEGA end results by
 * more: DESIGN.md section 5).  PCSF_BLOCK_KERNEL_AUTO (the default) lets every pcsf_score_blocks call pick one from ITS list (the mean
 * number of species with a codon per block), so a list scored whole and the same list cut into several calls -- or across several
 * devices -- may use different kernels.  A caller that cuts a list and wants the reference's guarantee that parallelisation does not
 * change a result (src/phylocsf++score_msa.hpp:115) decides ONCE, with pcsf_choose_block_kernel on the whole list, and pins the answer
 * on every context it uses (the command line's score-msa does).  The environment variable PCSF_BLOCK_KERNEL=tiles|factored overrides both. */
typedef enum { PCSF_BLOCK_KERNEL_AUTO = 0, PCSF_BLOCK_KERNEL_TILES = 1, PCSF_BLOCK_KERNEL_FACTORED = 2 } pcsf_block_kernel;
int pcsf_set_block_kernel(pcsf_ctx *ctx, int kernel);
/* Host-only (no GPU needed): the AUTO rule evaluated on a list -- PCSF_BLOCK_KERNEL_TILES or PCSF_BLOCK_KERNEL_FACTORED; a negative
 * pcsf_status on bad arguments.  n_leaves: rows of the codon matrix. */
int pcsf_choose_block_kernel(int32_t n_leaves, const uint8_t *codons, const uint64_t *site_offsets, size_t n_blocks,
                             size_t leaf_stride);
/* Optional, no reference counterpart: a caller that WILL call pcsf_score_blocks with MLE or OMEGA says so early (the
 * command line: before it reads the alignment file).  Returns at once; the device queues the lock-step engine works on
 * are created on a thread of the library's meanwhile (~11 ms each the first time, 35 ms of a 55 ms MLE call on 516
 * blocks otherwise).  No effect on any result; pcsf_score_blocks without it creates them when it needs them. */
int pcsf_prepare_block_scoring(pcsf_ctx *ctx, int algo);

/* One objective evaluation of the MLE / OMEGA strategies per block at caller-chosen parameters, without
 * the prior term: sum over the block's sites of ln z with the branch matrices at tree scale
 * tree_scale[b] (minimizer_lpr_leaves, src/fixed_lik.hpp:460-467; lpr_leaves_omega, src/omega.hpp:151-202).
 * model 0 = coding ECM, 1 = non-coding ECM, 2 = the kappa/omega F3x4 model with
 * q_settings[n_blocks][12] = kappa, omega, sigma and the nine ratios (src/run.hpp:75-79, else NULL).
 * out_failed[b] (may be NULL) = 1 where a branch matrix fails the reference's validation. */
int pcsf_block_loglik(pcsf_ctx *ctx, int model, const double *q_settings, const double *tree_scale,
                      const uint8_t *codons, const uint64_t *site_offsets, size_t n_blocks, size_t leaf_stride,
                      double *out_lpr, int32_t *out_failed);

/* The same evaluation with CALLER-SUPPLIED eigen-systems Q = right diag(lambda) left (the reference's eig.r_s,
 * eig.r_l, eig.r_s2 of src/instance.hpp:309-434) and root priors: right[n_eig][64][64] eigenvectors in columns,
 * left[n_eig][64][64] = right^-1 in rows, lambda[n_eig][64], prior[n_eig][64]; n_eig = 1 (shared) or n_blocks.
 * P(t) with the reference's clamp / diagonal / validation (src/instance.hpp:529-646) and the pruning run on the
 * device as in every other call; only the eigen-solver is the caller's.  The parity tests feed the oracle's
 * eigen-systems through this to hold the device's P(t) + pruning to 1e-9 at every parameter the optimisers visit. */
int pcsf_block_loglik_eig(pcsf_ctx *ctx, const double *right, const double *left, const double *lambda,
                          const double *prior, size_t n_eig, const double *tree_scale, const uint8_t *codons,
                          const uint64_t *site_offsets, size_t n_blocks, size_t leaf_stride, double *out_lpr,
                          int32_t *out_failed);

/* ---- diagnostics ---------------------------------------------------------------------------*/
/* eigen-systems of n kappa/omega F3x4 rate matrices (q_settings[n][12]) from the device solver that OMEGA uses at
 * every kappa evaluation (replaces gsl_eigen_nonsymmv + complex LU of src/instance.hpp:309-434):
 * right[n][64][64] (eigenvectors in columns), left[n][64][64] (= right^-1, in rows), lambda[n][64],
 * prior[n][64] (equilibrium, sums to 1), failed[n] = 1 where a codon frequency is not positive.
 * warm_settings (may be NULL): [n][12] settings whose eigenbasis the iteration starts from, as OMEGA does for its probes. */
int pcsf_debug_eigen_omega(pcsf_ctx *ctx, const double *q_settings, const double *warm_settings, size_t n, double *right,
                           double *left, double *lambda, double *prior, int32_t *failed);
/* host-only (works without a GPU): static facts about the ops of pcsf_build_schedule's program, for trees of at most 128 leaves
 * (PCSF_ERR_UNSUPPORTED otherwise) -- leaves[2 i], leaves[2 i + 1]: the leaves whose messages op i's result contains (bit l of
 * word l >> 6); meta[i] = program slot the op starts with | stack depth << 14 | (pops) index of the op that parked the tile << 18.
 * The factored block-mode kernel skips the ops of all-gap subtrees with them (tests/test_schedule.py). */
int pcsf_debug_program_facts(int32_t n_nodes, const int16_t *child1, const int16_t *child2, uint64_t *leaves, uint32_t *meta,
                             int32_t cap_ops);
/* host-only (works without a GPU): the pair schedule of the eigen-solver kernel (csrc/eigen.hip, register-resident variant) --
 * pairs[63][32][2]: the index pairs every step of one sweep rotates; located[15][16]: where each group of four indices
 * sits in each round (super-group | half << 3).  For tests/test_schedule.py: every pair exactly once, disjoint per step. */
void pcsf_debug_eigen_schedule(int32_t *pairs, int32_t *located);
/* host-only: the seven-wave form of the same kernel (the default; PCSF_EIGEN_FORM=2) keeps the schedule on the seats of quads of lanes: a round's
 * end re-seats B through a staged look-up (rr::shift_source) and V by register moves and a four-column exchange.  This replays both on indices over
 * two sweeps -- pairs[63][32][2] as the seats give them (inside steps: seat e with e ^ X; cross steps: seat k with 4 + ((k + t) & 3)) -- and returns
 * the number of places where a round's end and the seat -> index map disagree (0). */
int pcsf_debug_eigen_schedule_quads(int32_t *pairs);
/* milliseconds spent in the pruning kernel by the last *_fixed / *_fixed_dev call that was
 * timed; pcsf_set_timing(ctx,1) brackets each launch with HIP events ON THE LAUNCH STREAM. */
int pcsf_set_timing(pcsf_ctx *ctx, int enable);
double pcsf_last_kernel_ms(const pcsf_ctx *ctx);
/* in-kernel cycle stamps of one work-group (needs the diagnostic kernel build, PCSF_PRUNE_VARIANT=2; tools/stamps.py):
 * out == NULL arms the stamps for work-group `block` and returns the number of values; otherwise copies them out
 * (eight per-segment cycle sums, then eight counts). */
int pcsf_debug_stamps(pcsf_ctx *ctx, int block, unsigned long long *out, int cap);
/* the leaf-major codon matrix [n_leaves][*n_sites] of the last six-frame call (the separate extract kernel's; with
 * PCSF_DUMP_CODONS=1 at pcsf_create, what the fused front end formed inside the pruning kernel) */
int pcsf_debug_codon_matrix(pcsf_ctx *ctx, uint8_t *out, size_t capacity, size_t *n_sites);
/* copies the branch matrices the context holds back to the host: [2][n_nodes-1][64][64],
 * row = parent state (for parity tests of the P(t) stage) */
int pcsf_get_branch_matrices(pcsf_ctx *ctx, double *out);
/* host-only (works without a GPU): the straight-line pruning program the kernel interprets for
 * a tree, op = code<<28 | a<<14 | b (codes in phylocsfpp_amd/csrc/schedule.hpp), and for every
 * MATMUL slot the node whose branch matrix it uses.  program_cap >= 4*n_nodes is always enough;
 * slot_node needs n_nodes entries. */
int pcsf_build_schedule(int32_t n_nodes, const int16_t *child1, const int16_t *child2, uint32_t *program,
                        int32_t program_cap, int32_t *slot_node, int32_t *n_ops, int32_t *n_slots,
                        int32_t *stack_depth);
/* what this context's tracks-mode kernel runs: out[0] = products per site and model of the full program (internal non-root nodes), out[1] = of
 * the program it runs (subtree tables: one less per cherry, one more less per node over a cherry and a leaf), out[2] = cherries with a pair
 * table (0: the full program), out[3] = rows of 64 doubles per table set (leaves' rows + pair rows + triple rows + quad rows), out[4] = triples,
 * out[5] = quads (nodes over two cherries or over a triple and a leaf; PCSF_TABLE_LEAVES=4).  out needs 6 entries. */
int pcsf_tracks_program_info(const pcsf_ctx *ctx, int64_t *out);
/* host-only: the program tracks mode runs when the tree has cherries (csrc/schedule.hpp reduce_cherries: every cherry below the root is a
 * leaf whose "codon" is the pair of its leaves' codons; its message comes from a table of 65 x 65 rows made once per context; with
 * triples != NULL also every node over a cherry and a leaf, 65^3 rows).  An operand below 8192 is a leaf of the tree, 8192 + p the p-th
 * cherry, 12288 + t the t-th triple; pairs[3 p ..]: its leaves and its node; triples[3 t ..]: its pair, its third leaf, its node;
 * slot_node: the ORIGINAL tree's nodes.  *n_pairs = 0 and nothing else written when the tree has no such cherry.  pairs and triples
 * need 3 (n_nodes + 1) / 2 entries each. */
int pcsf_debug_cherry_schedule(int32_t n_nodes, const int16_t *child1, const int16_t *child2, uint32_t *program,
                               int32_t program_cap, int32_t *slot_node, int32_t *n_ops, int32_t *n_slots,
                               int32_t *stack_depth, int32_t *pairs, int32_t *n_pairs, int32_t *triples, int32_t *n_triples);
/* host-only: the same for any table depth (max_leaves 2, 3 or 4): operands[2 k], operands[2 k + 1] = an operand id >= 8192 that the program
 * names and the node of the tree whose message to its parent that operand's table holds (operands needs n_nodes + 1 entries). */
int pcsf_debug_table_schedule(int32_t n_nodes, const int16_t *child1, const int16_t *child2, int32_t max_leaves, uint32_t *program,
                              int32_t program_cap, int32_t *slot_node, int32_t *n_ops, int32_t *n_slots, int32_t *stack_depth,
                              int32_t *operands, int32_t *n_operands);
/* host-only: runs the lock-step optimiser (fit_find_init + Brent, src/fixed_lik.hpp:469-544) on a
 * caller-supplied objective (to be MAXIMISED), mt19937 seeded with `seed` and advanced by
 * `skip_draws` 32-bit draws; returns the evaluation points in order and the value of the LAST one. */
int pcsf_debug_maxlik(double (*objective)(double x, void *user), void *user, double init, double lo, double hi,
                      uint32_t seed, int skip_draws, double *xs_out, int cap, int *n_evals, double *last_lpr);
/* The same search ON THE DEVICE (the state machine the MLE / OMEGA engine runs per block, csrc/chain.hip), fed with a RECORDED sequence of
 * objective values: the points it asks for go to xs_out (at most cap), *n_requests = how many it asked for, or -1 when it wanted more
 * values than n_values (xs_out[n_values] is then the point it is waiting for).  A search that takes the decisions of the one that produced `values` asks for exactly its points. */
int pcsf_debug_maxlik_device(pcsf_ctx *ctx, double init, double lo, double hi, uint32_t seed, int skip_draws, const double *values,
                             int n_values, double *xs_out, int cap, int *n_requests, double *last_lpr);
/* Test hook: the number formatter of csrc/wig.hip (print_trimmed, src/common.hpp:48-68: "%.3f" / "%.4f" without trailing zeros, one decimal kept)
 * on n values.  text[32 * i ...]: value i's characters; lens[i]: their number, or 0xffffffff for a finite |value| >= 9e14 (left to the host).
 * decimals: 3 or 4. */
int pcsf_debug_wig_format(pcsf_ctx *ctx, const float *values, size_t n, int decimals, char *text, uint32_t *lens);
const char *pcsf_version(void);

/* Optional: page-locked host memory for the buffers handed to the scoring calls.  Transfers from/to such buffers
 * run at DMA speed and do not occupy host threads (a pageable 58 MB nucleotide matrix costs ~6 ms of CPU-driven
 * staging per call); any other memory works too.  NULL on failure. */
void *pcsf_host_alloc(size_t bytes);
/* the same through ctx's GPU (hipSetDevice first): threads that never touched a device do not create a context on GPU 0 */
void *pcsf_host_alloc_on(pcsf_ctx *ctx, size_t bytes);
void pcsf_host_free(void *p);

#ifdef __cplusplus
}
#endif
#endif
