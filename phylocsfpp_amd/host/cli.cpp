// phylocsf-amd: build-tracks / score-msa with the scoring core on an MI355X (through the C-ABI).
#include "backend_hip.hpp"
#include "cli_common.hpp"

int main(int argc, char** argv)
{
    return pcsf_host::cli_main(argc, argv, [](const pcsf_host::Model& m, int device) {
        return std::unique_ptr<pcsf_host::ScoringBackend>(new pcsf_host::HipBackend(m, device));
    }, /*fast_exit=*/true);
}
