// Host-only micro-benchmark of stage 1 of build-tracks on text input (host/drivers.hpp prepare_window, direct form): every byte-range job
// MEASURES its alignments (MafReader::measure: reference rows only), then the same readers FILL the rows into a matrix (MafReader::fill).
//   g++ -O2 -std=c++17 -pthread -o tools/_build/maf_direct_bench tools/maf_direct_bench.cpp
//   tools/_build/maf_direct_bench <model prefix> <maf> [threads [window MB]]
// Prints seconds per phase (wall, all threads) and a checksum of the filled rows.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>

#include "../phylocsfpp_amd/host/maf_reader.hpp"
#include "../phylocsfpp_amd/host/model_io.hpp"
#include "../phylocsfpp_amd/host/parallel.hpp"

using namespace pcsf_host;

template <class F>
static void run_parallel(size_t n, int threads, F&& f)
{
    parallel_for(n, threads, f);   // the driver's own (host/parallel.hpp)
}

int main(int argc, char** argv)
{
    if (argc < 3) return 1;
    const std::string prefix = argv[1];
    const Model model = load_model(prefix, prefix.substr(0, prefix.find_last_of('/')) + "/species_aliases.tsv");
    MafFile file(argv[2]);
    const int T = argc > 3 ? atoi(argv[3]) : 8;
    const size_t window = (argc > 4 ? static_cast<size_t>(atoi(argv[4])) : 32) << 20;
    const size_t nl = static_cast<size_t>(model.n_leaves);
    double t_measure = 0.0, t_fill = 0.0;
    uint64_t checksum = 0, columns = 0, alignments = 0;
    std::vector<char> matrix;
    for (size_t w0 = 0; w0 < file.size(); w0 += window) {
        const size_t w1 = std::min(file.size(), w0 + window);
        const size_t n_jobs = std::max<size_t>(1, std::min<size_t>(static_cast<size_t>(T) * 4, (w1 - w0 + (1 << 16) - 1) >> 16));
        std::vector<std::unique_ptr<MafReader>> readers(n_jobs);
        std::vector<std::vector<AlignmentSpan>> spans(n_jobs);
        auto t0 = std::chrono::steady_clock::now();
        std::vector<double> job_init(n_jobs, 0.0), job_meas(n_jobs, 0.0);
        run_parallel(n_jobs, T, [&](size_t j) {
            const size_t b = w0 + (w1 - w0) * j / n_jobs, e = w0 + (w1 - w0) * (j + 1) / n_jobs;
            const auto a0 = std::chrono::steady_clock::now();
            readers[j].reset(new MafReader(file, b, e, model.name_to_leaf, model.n_leaves, true, false));
            const auto a1 = std::chrono::steady_clock::now();
            AlignmentSpan sp;
            while (readers[j]->measure(sp)) spans[j].push_back(sp);
            job_init[j] = std::chrono::duration<double>(a1 - a0).count();
            job_meas[j] = std::chrono::duration<double>(std::chrono::steady_clock::now() - a1).count();
        });
        auto t1 = std::chrono::steady_clock::now();
        if (getenv("MAF_BENCH_VERBOSE") && w0 < 3 * window) {
            double si = 0, sm = 0, mi = 0, mm = 0;
            for (size_t j = 0; j < n_jobs; ++j) { si += job_init[j]; sm += job_meas[j]; mi = std::max(mi, job_init[j]); mm = std::max(mm, job_meas[j]); }
            printf("  window at %zu MB: %zu jobs; reader set-up (look-back + the alignment before) sum %.3f ms, max %.3f ms; measure sum %.3f ms, max %.3f ms; wall %.3f ms\n",
                   w0 >> 20, n_jobs, si * 1e3, mi * 1e3, sm * 1e3, mm * 1e3, std::chrono::duration<double>(t1 - t0).count() * 1e3);
        }
        size_t stride = 0;
        std::vector<std::vector<size_t>> col0(n_jobs);
        for (size_t j = 0; j < n_jobs; ++j)
            for (const AlignmentSpan& sp : spans[j]) {
                col0[j].push_back(stride);
                stride += sp.len;
                ++alignments;
            }
        stride = std::max<size_t>(stride, 1);
        if (matrix.size() < nl * stride) matrix.resize(nl * stride);
        auto t2 = std::chrono::steady_clock::now();
        run_parallel(n_jobs, T, [&](size_t j) {
            for (size_t k = 0; k < spans[j].size(); ++k) readers[j]->fill(spans[j][k], matrix.data() + col0[j][k], stride);
        });
        auto t3 = std::chrono::steady_clock::now();
        t_measure += std::chrono::duration<double>(t1 - t0).count();
        t_fill += std::chrono::duration<double>(t3 - t2).count();
        columns += stride;
        for (size_t l = 0; l < nl; ++l)
            for (size_t i = 0; i < stride; i += 61) checksum += (static_cast<unsigned char>(matrix[l * stride + i]) ^ (i << 8)) * 0x9e3779b97f4a7c15ull;
    }
    printf("%d threads, %zu MB windows: %llu alignments, %llu columns, checksum %016llx; measure %.3f s, fill %.3f s (%.0f MB/s overall)\n", T, window >> 20,
           (unsigned long long)alignments, (unsigned long long)columns, (unsigned long long)checksum, t_measure, t_fill, file.size() / (t_measure + t_fill) / 1e6);
    return 0;
}
