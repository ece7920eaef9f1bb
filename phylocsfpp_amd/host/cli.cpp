// phylocsf-amd: build-tracks / score-msa with the scoring core on an MI355X (through the C-ABI).
#include "backend_hip.hpp"
#include "cli_common.hpp"

int main(int argc, char** argv)
{
    return pcsf_host::cli_main(argc, argv, [](const pcsf_host::Model& m, int device) {
        return std::unique_ptr<pcsf_host::ScoringBackend>(new pcsf_host::HipBackend(m, device));
    }, /*fast_exit=*/true,
    // --ranks: the ranks' text gathered over RCCL between their GPUs (csrc/comm.cpp)
    [](pcsf_host::ScoringBackend& b, int rank, int world, const std::string& rendezvous, int threads) {
        pcsf_host::HipBackend* hip = dynamic_cast<pcsf_host::HipBackend*>(&b);
        if (!hip) throw std::runtime_error("--ranks needs the HIP backend");
        return std::unique_ptr<pcsf_host::RankGather>(new pcsf_host::RcclGather(*hip, rank, world, rendezvous, threads));
    });
}
