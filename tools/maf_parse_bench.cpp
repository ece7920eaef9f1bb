// Host-only micro-benchmark of the MAF reader: MB/s with T threads on byte-range jobs, and a checksum of what it yields.
//   g++ -O2 -std=c++17 -pthread -o /tmp/maf_parse_bench tools/maf_parse_bench.cpp && /tmp/maf_parse_bench <model prefix> <maf> [threads...]
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "../phylocsfpp_amd/host/maf_reader.hpp"
#include "../phylocsfpp_amd/host/model_io.hpp"

int main(int argc, char** argv)
{
    using namespace pcsf_host;
    if (argc < 3) return 1;
    const std::string prefix = argv[1];
    const Model model = load_model(prefix, prefix.substr(0, prefix.find_last_of('/')) + "/species_aliases.tsv");
    MafFile file(argv[2]);
    std::vector<int> counts;
    for (int i = 3; i < argc; ++i) counts.push_back(atoi(argv[i]));
    if (counts.empty()) counts = {1, 1};
    for (int T : counts) {
        const auto t0 = std::chrono::steady_clock::now();
        const size_t jobs = static_cast<size_t>(T) * 4;
        std::vector<uint64_t> sums(jobs, 0), cols(jobs, 0);
        std::atomic<size_t> next{0};
        auto work = [&]() {
            for (size_t j = next.fetch_add(1); j < jobs; j = next.fetch_add(1)) {
                MafReader rd(file, file.size() * j / jobs, file.size() * (j + 1) / jobs, model.name_to_leaf, model.n_leaves, true, false);
                Alignment aln;
                uint64_t h = 0;
                while (rd.next(aln)) {
                    cols[j] += aln.length();
                    h += aln.start_pos * 1099511628211ull;
                    for (const auto& r : aln.seqs)
                        for (size_t i = 0; i < r.size(); i += 97) h += (static_cast<unsigned char>(r[i]) ^ (r.size() << 8)) * 0x9e3779b97f4a7c15ull;
                }
                sums[j] = h;
            }
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < T; ++t) pool.emplace_back(work);
        work();
        for (auto& t : pool) t.join();
        uint64_t h = 0, c = 0;
        for (size_t j = 0; j < jobs; ++j) { h += sums[j]; c += cols[j]; }
        const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("%2d threads: %llu columns, checksum %016llx: %.3f s = %.0f MB/s\n", T, (unsigned long long)c, (unsigned long long)h, s,
               file.size() / s / 1e6);
    }
    return 0;
}
