// backend_hip.hpp -- the product's scoring backend: forwards to the C-ABI (libphylocsf_amd.so).
#pragma once

#include <algorithm>
#include <cinttypes>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/phylocsf_amd.h"
#include "drivers.hpp"

namespace pcsf_host {

class HipBackend : public ScoringBackend {
public:
    HipBackend(const Model& m, int device)
    {
        pcsf_model_desc d{};
        d.n_nodes = m.n_nodes;
        d.child1 = m.child1.data();
        d.child2 = m.child2.data();
        d.branch_len = m.branch_len.data();
        d.S_coding = m.S_c.data();
        d.pi_coding = m.pi_c.data();
        d.S_noncoding = m.S_nc.data();
        d.pi_noncoding = m.pi_nc.data();
        const int rc = pcsf_create(&ctx_, &d, device);
        if (rc != PCSF_OK) throw std::runtime_error(std::string("pcsf_create failed: ") + pcsf_last_error(nullptr));
        // branch-length scores on the device: the tree's double branch lengths (up to 128 species)
        device_bls_ = m.n_leaves <= 128 && pcsf_set_bls_branch_lengths(ctx_, m.branch_len_d.data()) == PCSF_OK;
    }
    bool device_bls() const override { return device_bls_; }
    int contexts_per_device() const override { return 2; }
    void prepare_tracks(size_t max_columns) override { check(pcsf_prepare_tracks(ctx_, max_columns)); }
    ~HipBackend() override { pcsf_destroy(ctx_); }
    void score_sites_fixed(const uint8_t* codons, size_t n_sites, size_t leaf_stride, double* out) override
    {
        check(pcsf_score_sites_fixed(ctx_, codons, n_sites, leaf_stride, out));
    }
    void score_blocks(int algo, const uint8_t* codons, const uint64_t* offsets, size_t n_blocks, size_t leaf_stride, bool compute_anc,
                      uint32_t seed, float* phylo, float* anc) override
    {
        check(pcsf_score_blocks(ctx_, algo, codons, offsets, n_blocks, leaf_stride, compute_anc ? 1 : 0, seed, phylo, anc));
    }
    void prepare_blocks(int algo) override { check(pcsf_prepare_block_scoring(ctx_, algo)); }
    int choose_block_kernel(const uint8_t* codons, const uint64_t* offsets, size_t n_blocks, size_t leaf_stride) override
    {
        const int k = pcsf_choose_block_kernel(pcsf_n_leaves(ctx_), codons, offsets, n_blocks, leaf_stride);
        if (k < 0) throw std::runtime_error("phylocsf_amd: pcsf_choose_block_kernel: bad argument");
        return k;
    }
    void set_block_kernel(int kernel) override { check(pcsf_set_block_kernel(ctx_, kernel)); }
    void hmm_release() override { (void)pcsf_hmm_release(ctx_); }
    bool device_hmm() const override { return std::getenv("PCSF_HOST_HMM") == nullptr; }   // PCSF_HOST_HMM=1: the host code (A/B, tests)
    void hmm_smooth_runs(const Hmm& h, const double* emissions, const uint64_t* run_offsets, size_t n_runs, double* posterior, uint8_t* back, double* last) override
    {
        pcsf_hmm_stats st{};
        check(pcsf_hmm_smooth_runs(ctx_, h.init, &h.trans[0][0], emissions, run_offsets, n_runs, posterior, back, last, &st));
        if (std::getenv("PCSF_HMM_STATS"))
            fprintf(stderr, "hmm_smooth_runs: %zu runs, %" PRIu64 " values in %u chunks; evaluated again: forward %u (%u passes), backward %u (%u), viterbi %u (%u)\n", n_runs,
                    run_offsets[n_runs], st.chunks, st.forward_reruns, st.forward_passes, st.backward_reruns, st.backward_passes, st.viterbi_reruns, st.viterbi_passes);
    }
    size_t hmm_smooth_runs_binned(const Hmm& h, const double* emissions, const uint64_t* run_offsets, size_t n_runs, const double* thresholds, size_t n_thresholds,
                                  uint16_t* bins, uint64_t* exc_index, double* exc_value, size_t exc_capacity, uint8_t* back, double* last) override
    {
        pcsf_hmm_stats st{};
        size_t n_exc = 0;
        check(pcsf_hmm_smooth_runs_binned(ctx_, h.init, &h.trans[0][0], emissions, run_offsets, n_runs, thresholds, n_thresholds, bins, exc_index, exc_value, exc_capacity,
                                          &n_exc, back, last, &st));
        if (std::getenv("PCSF_HMM_STATS"))
            fprintf(stderr, "hmm_smooth_runs: %zu runs, %" PRIu64 " values in %u chunks; evaluated again: forward %u (%u passes), backward %u (%u), viterbi %u (%u); %zu without a posterior\n",
                    n_runs, run_offsets[n_runs], st.chunks, st.forward_reruns, st.forward_passes, st.backward_reruns, st.backward_passes, st.viterbi_reruns, st.viterbi_passes, n_exc);
        return n_exc;
    }

    size_t score_alignments_6frames(int, const char* nucl, size_t row_stride, const std::vector<uint64_t>& col0, const std::vector<uint32_t>& len,
                                    const std::vector<uint64_t>& start_pos, const std::vector<uint64_t>& chrom_len, double* out, size_t out_capacity,
                                    std::vector<uint64_t>& offsets, std::vector<uint32_t>& n_codons, double* bls_out) override
    {
        const size_t n = col0.size();
        std::vector<pcsf_alignment_desc> d(n);
        for (size_t k = 0; k < n; ++k) d[k] = pcsf_alignment_desc{col0[k], len[k], start_pos[k], chrom_len[k]};
        offsets.assign(n * 6, 0);
        n_codons.assign(n * 6, 0);
        check(pcsf_score_alignments_6frames_bls(ctx_, nucl, row_stride, d.data(), n, out, out_capacity, offsets.data(), n_codons.data(), nullptr, bls_out));
        size_t total = 0;
        for (size_t i = 0; i < n * 6; ++i) total += n_codons[i];
        return total;
    }

    // PCSF_DEVICE_WIG=0: the host formats the tracks (the A/B switch of tests/test_gpu_cli.py and tools/config4_one_gpu.sh)
    bool device_wig() const override
    {
        const char* v = std::getenv("PCSF_DEVICE_WIG");
        return device_bls_ && !(v && v[0] == '0');
    }
    bool score_alignments_6frames_wig(const char* nucl, size_t row_stride, const std::vector<uint64_t>& col0, const std::vector<uint32_t>& len,
                                      const std::vector<uint64_t>& start_pos, const std::vector<uint64_t>& chrom_len, const std::vector<const char*>& chroms,
                                      float power_threshold, bool want_power, bool want_raw, char* text, size_t text_capacity, uint64_t* bytes,
                                      size_t* n_codon_frames) override
    {
        const size_t n = col0.size();
        std::vector<pcsf_alignment_desc> d(n);
        for (size_t k = 0; k < n; ++k) d[k] = pcsf_alignment_desc{col0[k], len[k], start_pos[k], chrom_len[k]};
        const int rc = pcsf_score_alignments_6frames_wig(ctx_, nucl, row_stride, d.data(), n, chroms.data(), power_threshold, want_power ? 1 : 0, want_raw ? 1 : 0,
                                                         text, text_capacity, bytes, n_codon_frames);
        if (rc == PCSF_ERR_UNSUPPORTED) return false;
        check(rc);
        return true;
    }
    size_t wig_text_estimate(size_t n_columns, size_t n_alignments) override { return pcsf_wig_text_estimate(n_columns, n_alignments); }
    void fetch_wig_text(char* const* dst) override { check(pcsf_fetch_wig_text(ctx_, dst)); }

    char* staging_alloc(size_t bytes) override
    {
        // on this context's GPU (no primary context on device 0 for a run on another device); never zero bytes
        char* p = static_cast<char*>(pcsf_host_alloc_on(ctx_, bytes ? bytes : 1));
        if (!p) throw std::runtime_error("pcsf_host_alloc_on failed");
        return p;
    }
    void staging_free(char* p) override { pcsf_host_free(p); }

private:
    void check(int rc)
    {
        if (rc != PCSF_OK) throw std::runtime_error(std::string("phylocsf_amd: ") + pcsf_last_error(ctx_));
    }
    pcsf_ctx* ctx_ = nullptr;
    bool device_bls_ = false;
};

}  // namespace pcsf_host
