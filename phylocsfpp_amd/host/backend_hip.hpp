// backend_hip.hpp -- the product's scoring backend: forwards to the C-ABI (libphylocsf_amd.so).
#pragma once

#include <stdexcept>
#include <string>

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
    }
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

private:
    void check(int rc)
    {
        if (rc != PCSF_OK) throw std::runtime_error(std::string("phylocsf_amd: ") + pcsf_last_error(ctx_));
    }
    pcsf_ctx* ctx_ = nullptr;
};

}  // namespace pcsf_host
