// oracle_cli.cpp -- TEST INFRASTRUCTURE: the host driver of phylocsfpp_amd/host/ (MAF reader,
// frames, BLS, wig/.scores writers) run on top of the CPU oracle instead of the HIP C-ABI.
// It exists to pin (host layer + oracle) against the reference's golden output files
// (tests/test_oracle_golden.py) and is never used by the product.
#include <cmath>

#include "../phylocsfpp_amd/host/cli_common.hpp"
#include "pcsf_oracle.h"

namespace {

class OracleBackend : public pcsf_host::ScoringBackend {
public:
    explicit OracleBackend(const pcsf_host::Model& m) : model_(new po_model)
    {
        po_tree& t = model_->tree;
        memset(&t, 0, sizeof t);
        t.n = m.n_nodes;
        t.nl = m.n_leaves;
        for (int i = 0; i < m.n_nodes; ++i) {
            t.child1[i] = m.child1[i];
            t.child2[i] = m.child2[i];
            t.parent[i] = m.parent[i];
            t.branch_len[i] = m.branch_len[i];
            t.branch_len_d[i] = m.branch_len_d[i];
        }
        std::copy(m.S_c.begin(), m.S_c.end(), model_->S_c);
        std::copy(m.pi_c.begin(), m.pi_c.end(), model_->pi_c);
        std::copy(m.S_nc.begin(), m.S_nc.end(), model_->S_nc);
        std::copy(m.pi_nc.begin(), m.pi_nc.end(), model_->pi_nc);
    }
    void score_sites_fixed(const uint8_t* codons, size_t n_sites, size_t leaf_stride, double* out) override
    {
        // one "alignment" per 4096 sites so that OpenMP has work; results do not depend on the split
        std::vector<uint64_t> off;
        for (size_t s = 0; s < n_sites; s += 4096) off.push_back(s);
        off.push_back(n_sites);
        if (po_baseline_run_tracks(model_.get(), codons, off.data(), off.size() - 1, leaf_stride, out, pcsf_host::effective_threads()))  // cgroup quota / PCSF_THREADS
            throw std::runtime_error("oracle: run_tracks failed");
    }
    void score_blocks(int algo, const uint8_t* codons, const uint64_t* offsets, size_t n_blocks, size_t leaf_stride, bool compute_anc,
                      uint32_t seed, float* phylo, float* anc) override
    {
        po_run_blocks(model_.get(), algo, codons, offsets, n_blocks, leaf_stride, compute_anc ? 1 : 0, seed, phylo, anc, 0);
    }

private:
    std::unique_ptr<po_model> model_;
};

}  // namespace

int main(int argc, char** argv)
{
    return pcsf_host::cli_main(argc, argv, [](const pcsf_host::Model& m, int) {
        return std::unique_ptr<pcsf_host::ScoringBackend>(new OracleBackend(m));
    });
}
