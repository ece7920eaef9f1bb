// backend_hip.hpp -- the product's scoring backend: forwards to the C-ABI (libphylocsf_amd.so).
#pragma once

#include <algorithm>
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
    void warm_up(size_t window_bytes, bool staging) override
    {
        check(pcsf_prepare_tracks(ctx_, std::max<size_t>(window_bytes / 64, 4096)));  // ~90 bytes of text per column
        if (!staging) return;
        // Page-locked buffers for the first four windows of a run (1/4, 1/2, 1, 1 window; drivers.hpp): rows ~0.75 byte per
        // byte of MAF text, results ~0.33.  Page-locking costs milliseconds per 10 MB, and doing it while the first
        // window is parsed slows the parser down (both work on the process's page tables); later windows recycle.
        for (size_t w : {window_bytes / 4, window_bytes / 2, window_bytes, window_bytes})
            for (size_t bytes : {w - w / 4, w / 3}) {
                char* p = static_cast<char*>(pcsf_host_alloc_on(ctx_, std::max<size_t>(bytes, 1)));
                if (p) cache_.push_back({p, bytes, false});
            }
    }
    ~HipBackend() override
    {
        for (const Cached& c : cache_) pcsf_host_free(c.ptr);
        pcsf_destroy(ctx_);
    }
    void score_sites_fixed(const uint8_t* codons, size_t n_sites, size_t leaf_stride, double* out) override
    {
        check(pcsf_score_sites_fixed(ctx_, codons, n_sites, leaf_stride, out));
    }
    void score_blocks(int algo, const uint8_t* codons, const uint64_t* offsets, size_t n_blocks, size_t leaf_stride, bool compute_anc,
                      uint32_t seed, float* phylo, float* anc) override
    {
        check(pcsf_score_blocks(ctx_, algo, codons, offsets, n_blocks, leaf_stride, compute_anc ? 1 : 0, seed, phylo, anc));
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

    char* staging_alloc(size_t bytes) override
    {
        {  // smallest fitting buffer of the warm-up cache
            std::lock_guard<std::mutex> g(cache_lock_);
            Cached* best = nullptr;
            for (Cached& c : cache_)
                if (!c.in_use && c.bytes >= bytes && (!best || c.bytes < best->bytes)) best = &c;
            if (best) {
                best->in_use = true;
                return best->ptr;
            }
        }
        // on this context's GPU (no primary context on device 0 for a run on another device); never zero bytes
        char* p = static_cast<char*>(pcsf_host_alloc_on(ctx_, bytes ? bytes : 1));
        if (!p) throw std::runtime_error("pcsf_host_alloc_on failed");
        return p;
    }
    void staging_free(char* p) override
    {
        {
            std::lock_guard<std::mutex> g(cache_lock_);
            for (Cached& c : cache_)
                if (c.ptr == p) {
                    c.in_use = false;  // stays page-locked for the next run of this process
                    return;
                }
        }
        pcsf_host_free(p);
    }

private:
    void check(int rc)
    {
        if (rc != PCSF_OK) throw std::runtime_error(std::string("phylocsf_amd: ") + pcsf_last_error(ctx_));
    }
    struct Cached {
        char* ptr;
        size_t bytes;
        bool in_use;
    };
    std::vector<Cached> cache_;
    std::mutex cache_lock_;
    pcsf_ctx* ctx_ = nullptr;
    bool device_bls_ = false;
};

}  // namespace pcsf_host
