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
    void retain_wig_text(uint64_t order) override { check(pcsf_wig_retain_last(ctx_, order)); }
    pcsf_ctx* context() const { return ctx_; }

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

// The ranks' gather of the product: RCCL between the GPUs (csrc/comm.cpp).  A rank's text is formatted on its GPU, kept there, and moves
// to rank 0's GPU in one grouped send / receive; rank 0 copies it down file by file.  The communicator's identifier travels through
// a file in `rendezvous` (a directory the launcher made): rank 0 writes it, the others wait for it.  The communicator is set up on a
// thread of its own while the rank scores.
class RcclGather : public RankGather {
public:
    RcclGather(HipBackend& backend, int rank, int world, const std::string& rendezvous, int threads)
        : ctx_(backend.context()), rank_(rank), world_(world), threads_(threads)
    {
        auto init = [this, rendezvous]() {
            StopWatch set_up;
            char id[PCSF_COMM_ID_BYTES];
            const std::string path = rendezvous + "/rccl_id";
            if (rank_ == 0) {
                if (pcsf_comm_unique_id(id) != PCSF_OK) throw std::runtime_error("pcsf_comm_unique_id failed (is librccl.so there?)");
                const std::string tmp = path + ".tmp";
                FILE* f = fopen(tmp.c_str(), "wb");
                if (!f || fwrite(id, 1, sizeof id, f) != sizeof id || fclose(f) != 0 || rename(tmp.c_str(), path.c_str()) != 0)
                    throw std::runtime_error("cannot write the communicator's identifier to '" + path + "'");
            } else {
                const char* t = std::getenv("PCSF_RENDEZVOUS_TIMEOUT");
                const double limit = t ? atof(t) : 300.0;
                StopWatch waited;
                double total = 0.0;
                for (;;) {
                    if (FILE* f = fopen(path.c_str(), "rb")) {
                        const size_t got = fread(id, 1, sizeof id, f);
                        fclose(f);
                        if (got == sizeof id) break;
                    }
                    if ((total += waited.lap()) > limit) throw std::runtime_error("rank 0 did not publish the communicator's identifier in '" + path + "'");
                    std::this_thread::sleep_for(std::chrono::milliseconds(2));
                }
            }
            if (pcsf_comm_create(ctx_, rank_, world_, id) != PCSF_OK) throw std::runtime_error(std::string("pcsf_comm_create: ") + pcsf_last_error(ctx_));
            setup_seconds_ = set_up.lap();
        };
        const char* sync = std::getenv("PCSF_COMM_INIT_SYNC");
        if (sync && sync[0] == '1') {
            init();
        } else {
            ready_ = std::async(std::launch::async, init);
        }
    }
    ~RcclGather() override
    {
        try {
            if (ready_.valid()) ready_.get();
        } catch (...) {
        }
        for (char* p : staging_)
            if (p) pcsf_host_free(p);
    }
    int rank() const override { return rank_; }
    int world() const override { return world_; }
    double setup_seconds() const override { return setup_seconds_; }
    void begin() override { check(pcsf_wig_retain_reset(ctx_)); }
    void keep_backend_text(ScoringBackend& backend, uint64_t order) override { backend.retain_wig_text(order); }
    void keep_host_text(uint64_t order, const char* const* text, const uint64_t* bytes) override { check(pcsf_wig_retain_host(ctx_, order, text, bytes)); }
    void gather(bool include_root, const std::function<void(int, const char*, size_t)>& sink) override
    {
        if (ready_.valid()) ready_.get();   // rethrows what the set-up threw
        std::vector<uint64_t> bytes(static_cast<size_t>(world_) * 7, 0);
        check(pcsf_comm_gather_wig(ctx_, 0, include_root ? 1 : 0, bytes.data()));
        if (rank_ != 0) return;
        // the seven files side by side: each copies its senders' pieces down in rank order through a page-locked buffer of its own
        staging_.resize(7, nullptr);
        staging_bytes_.resize(7, 0);
        parallel_for(7, threads_, [&](size_t f) {
            uint64_t largest = 0;
            for (int r = 0; r < world_; ++r) largest = std::max(largest, bytes[static_cast<size_t>(r) * 7 + f]);
            if (!largest) return;
            if (staging_bytes_[f] < largest) {
                if (staging_[f]) pcsf_host_free(staging_[f]);
                staging_[f] = static_cast<char*>(pcsf_host_alloc_on(ctx_, largest));
                staging_bytes_[f] = staging_[f] ? largest : 0;
                if (!staging_[f]) throw std::runtime_error("pcsf_host_alloc_on failed");
            }
            for (int r = 0; r < world_; ++r) {
                const uint64_t n = bytes[static_cast<size_t>(r) * 7 + f];
                if (!n) continue;
                if (pcsf_comm_fetch_gathered(ctx_, r, static_cast<int>(f), staging_[f]) != PCSF_OK)
                    throw std::runtime_error(std::string("phylocsf_amd: ") + pcsf_last_error(ctx_));
                sink(static_cast<int>(f), staging_[f], static_cast<size_t>(n));
            }
        });
    }

private:
    void check(int rc)
    {
        if (rc != PCSF_OK) throw std::runtime_error(std::string("phylocsf_amd: ") + pcsf_last_error(ctx_));
    }
    pcsf_ctx* ctx_;
    int rank_, world_, threads_;
    double setup_seconds_ = 0.0;   // identifier exchange + ncclCommInitRank
    std::future<void> ready_;
    std::vector<char*> staging_;
    std::vector<uint64_t> staging_bytes_;
};

}  // namespace pcsf_host
