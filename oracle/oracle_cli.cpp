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
    explicit OracleBackend(const pcsf_host::Model& m) : host_model_(&m), model_(new po_model)
    {
        // tests: the driver's "the backend computes the branch-length scores from the uploaded rows" form -- which also switches its
        // stage 1 to measuring the alignments and parsing them straight into the batch matrices -- on the CPU
        const char* v = std::getenv("PCSF_ORACLE_BACKEND_BLS");
        bls_from_rows_ = v && v[0] == '1';
        // tests of the driver's pipeline alone: a "score" that is a cheap checksum of the codon column (every byte of every batch matrix
        // still decides the output files; a 1 GB input takes seconds)
        const char* f = std::getenv("PCSF_ORACLE_BACKEND_FAKE");
        fake_scores_ = f && f[0] == '1';
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
        if (fake_scores_) {
            const size_t nl = static_cast<size_t>(model_->tree.nl);
            for (size_t s = 0; s < n_sites; ++s) {
                uint64_t h = 1469598103934665603ull;
                for (size_t l = 0; l < nl; ++l) h = (h ^ codons[l * leaf_stride + s]) * 1099511628211ull;
                out[s] = static_cast<double>(static_cast<int64_t>(h % 200001) - 100000) / 1000.0;
            }
            return;
        }
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

    bool device_bls() const override { return bls_from_rows_; }
    size_t score_alignments_6frames(int n_leaves, const char* nucl, size_t row_stride, const std::vector<uint64_t>& col0, const std::vector<uint32_t>& len,
                                    const std::vector<uint64_t>& start_pos, const std::vector<uint64_t>& chrom_len, double* out, size_t out_capacity,
                                    std::vector<uint64_t>& offsets, std::vector<uint32_t>& n_codons, double* bls_out) override
    {
        if (bls_out) {   // the host's own branch-length score on the rows of every alignment (like the power track's loop in the driver)
            std::fill(bls_out, bls_out + row_stride, 0.0);
            pcsf_host::BranchLengthScore bls(*host_model_);
            pcsf_host::Alignment aln;
            std::vector<double> v;
            for (size_t k = 0; k < col0.size(); ++k) {
                aln.seqs.assign(static_cast<size_t>(n_leaves), std::string());
                for (int l = 0; l < n_leaves; ++l) aln.seqs[static_cast<size_t>(l)].assign(nucl + static_cast<size_t>(l) * row_stride + col0[k], len[k]);
                bls.per_base(aln, &v);
                std::copy(v.begin(), v.end(), bls_out + col0[k]);
            }
        }
        return pcsf_host::ScoringBackend::score_alignments_6frames(n_leaves, nucl, row_stride, col0, len, start_pos, chrom_len, out, out_capacity, offsets, n_codons, nullptr);
    }

private:
    const pcsf_host::Model* host_model_;
    bool bls_from_rows_ = false, fake_scores_ = false;
    std::unique_ptr<po_model> model_;
};

// --ranks on the CPU: the ranks' text meets in files of the launcher's directory (the product's gather is RCCL between GPUs,
// phylocsfpp_amd/host/backend_hip.hpp).  What the tests pin with it: the byte ranges of the ranks, the order keys of the batches and the
// order of the merge -- the driver's half of the multi-process mode, which is the same code over either gather.
class FileGather : public pcsf_host::RankGather {
public:
    FileGather(int rank, int world, const std::string& dir) : rank_(rank), world_(world), dir_(dir) {}
    int rank() const override { return rank_; }
    int world() const override { return world_; }
    void begin() override
    {
        kept_.clear();
        ++round_;
    }
    void keep_backend_text(pcsf_host::ScoringBackend&, uint64_t) override { throw std::runtime_error("FileGather: the oracle backend formats nothing"); }
    void keep_host_text(uint64_t order, const char* const* text, const uint64_t* bytes) override
    {
        std::lock_guard<std::mutex> g(lock_);
        Kept k;
        k.order = order;
        for (int f = 0; f < 7; ++f) k.text[f].assign(text[f] ? text[f] : "", static_cast<size_t>(bytes[f]));
        kept_.push_back(std::move(k));
    }
    void gather(bool include_root, const std::function<void(int, const char*, size_t)>& sink) override
    {
        const std::string stem = dir_ + "/text." + std::to_string(round_) + ".";
        if (rank_ != 0 || include_root) {
            std::stable_sort(kept_.begin(), kept_.end(), [](const Kept& a, const Kept& b) { return a.order < b.order; });
            for (int f = 0; f < 7; ++f) {
                const std::string path = stem + std::to_string(rank_) + "." + std::to_string(f);
                FILE* out = fopen((path + ".tmp").c_str(), "wb");
                if (!out) throw std::runtime_error("FileGather: cannot create '" + path + "'");
                for (const Kept& k : kept_)
                    if (!k.text[f].empty() && fwrite(k.text[f].data(), 1, k.text[f].size(), out) != k.text[f].size()) throw std::runtime_error("FileGather: write failed");
                if (fclose(out) != 0 || rename((path + ".tmp").c_str(), path.c_str()) != 0) throw std::runtime_error("FileGather: write failed");
            }
            kept_.clear();
        }
        if (rank_ != 0) return;
        for (int f = 0; f < 7; ++f)
            for (int r = include_root ? 0 : 1; r < world_; ++r) {
                const std::string path = stem + std::to_string(r) + "." + std::to_string(f);
                FILE* in = nullptr;
                for (int spins = 0; !(in = fopen(path.c_str(), "rb")); ++spins) {
                    if (spins > 300000) throw std::runtime_error("FileGather: rank " + std::to_string(r) + " never delivered '" + path + "'");
                    std::this_thread::sleep_for(std::chrono::milliseconds(2));
                }
                std::vector<char> buf(1 << 20);
                for (size_t n; (n = fread(buf.data(), 1, buf.size(), in)) > 0;) sink(f, buf.data(), n);
                fclose(in);
                unlink(path.c_str());
            }
    }

private:
    struct Kept {
        uint64_t order = 0;
        std::string text[7];
    };
    int rank_, world_;
    std::string dir_;
    int round_ = 0;
    std::mutex lock_;
    std::vector<Kept> kept_;
};

}  // namespace

int main(int argc, char** argv)
{
    return pcsf_host::cli_main(argc, argv, [](const pcsf_host::Model& m, int) {
        return std::unique_ptr<pcsf_host::ScoringBackend>(new OracleBackend(m));
    }, /*fast_exit=*/false,
    [](pcsf_host::ScoringBackend&, int rank, int world, const std::string& rendezvous, int) {
        return std::unique_ptr<pcsf_host::RankGather>(new FileGather(rank, world, rendezvous));
    });
}
