// cli_common.hpp -- argument handling shared by the product CLI and the oracle's test CLI.
// Same sub-commands, option names and defaults as the reference (`phylocsf++ build-tracks`,
// `phylocsf++ score-msa`: src/phylocsf++build_tracks.hpp:367-507, src/phylocsf++score_msa.hpp:245-385)
// for the options that reach the scoring path; options must precede the positionals.
#pragma once

#include <dirent.h>
#include <signal.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <cstdio>
#include <cstring>
#include <chrono>
#include <functional>
#include <limits>
#include <memory>
#include <string>
#include <vector>

#include "drivers.hpp"

namespace pcsf_host {

inline bool parse_bool(const std::string& v) { return v == "1" || v == "true" || v == "one"; }

inline int usage(const char* prog)
{
    fprintf(stderr,
            "usage:\n"
            "  %s build-tracks [--output DIR] [--power-threshold 0|1] [--output-raw-phylo 0|1] [--output-power 0|1]\n"
            "                  [--output-phylo 0|1] [--output-regions 0|1] [--genome-length N --coding-exons FILE | --hmm-params P]\n"
            "                  [--species A,B,...] [--mapping FILE] [--species-aliases FILE] [--device N | --devices N,M,...]\n"
            "                  [--threads N] [--window-bytes N] [--batch-sites N] [--contexts-per-device N] [--host-bls 0|1]\n"
            "                  [--ranks N [--gather-root-text 0|1]] [--table-leaves 2|3|4] [--table-bytes N]\n"
            "                  <model prefix> <alignment.maf>...\n"
            "  %s score-msa    [--strategy fixed|mle|omega|fixed_mean] [--comp-phylo 0|1] [--comp-anc 0|1] [--comp-bls 0|1]\n"
            "                  [--output DIR] [--species A,B,...] [--mapping FILE] [--species-aliases FILE] [--device N | --devices N,M,...]\n"
            "                  <model prefix> <alignment.maf>...\n"
            "  %s smooth-tracks [--output-phylo 0|1] [--output-regions 0|1] [--context-model MODEL [--device N]] (--genome-length N --coding-exons FILE | --hmm-params P) DIR\n"
            "  %s pack-maf     [--threads N] [--species A,B,...] [--mapping FILE] <model prefix> <alignment.maf> <out.pcsfbin>\n"
            "                  (build-tracks also reads the packed file: same tracks, no text parsing)\n"
            "--hmm-params: coding prior, mean coding length (codons), 3 non-coding weights, 3 non-coding mean lengths (codons)\n"
            "--species reduces the model's tree to the named species (tree labels or their other names)\n"
            "--table-leaves / --table-bytes: how far the scoring contexts' subtree tables go (default 3 leaves within 8 GiB; 4 leaves need ~9 GB per\n"
            "           node and model -- 94 GB for 58mammals -- and pay from about a gigabase of input); the scores do not depend on it\n"
            "--ranks N: build-tracks as N processes, one per GPU (--devices lists them; default 0..N-1): rank r scores the r-th of N byte ranges\n"
            "           of every input, the ranks' text is gathered onto rank 0's GPU over RCCL and rank 0 writes the files\n"
            "<model prefix> names <prefix>_coding.ECM, <prefix>_noncoding.ECM and <prefix>.nh\n"
            "environment: PCSF_FULL_TEARDOWN=1 keeps the orderly exit (scoring contexts and the HIP runtime torn down instead of _exit once every\n"
            "             output file is closed; automatic under LD_PRELOAD / rocprofv3, whose tools write their files at exit)\n",
            prog, prog, prog, prog);
    return 2;
}

using BackendFactory = std::function<std::unique_ptr<ScoringBackend>(const Model&, int device)>;
// the ranks' gather of a build (--ranks): over `backend`, for rank `rank` of `world`; `rendezvous` is a directory all ranks see
using GatherFactory = std::function<std::unique_ptr<RankGather>(ScoringBackend& backend, int rank, int world, const std::string& rendezvous, int threads)>;

// --ranks N, the launcher: N copies of this command line, told apart by PCSF_RANK / PCSF_WORLD, with a fresh directory to meet in
// (PCSF_RENDEZVOUS).  The launcher itself never touches a GPU.  The first rank that fails ends the others (they would wait for it in
// the gather for ever); the launcher's status is that rank's.
inline int launch_ranks(char** argv, int n_ranks)
{
    char rdv[] = "/tmp/pcsf_ranks_XXXXXX";
    if (!mkdtemp(rdv)) {
        perror("error: mkdtemp");
        return 1;
    }
    std::vector<pid_t> pids(static_cast<size_t>(n_ranks), -1);
    for (int r = 0; r < n_ranks; ++r) {
        const pid_t pid = fork();
        if (pid < 0) {
            perror("error: fork");
            for (pid_t p : pids) if (p > 0) kill(p, SIGTERM);
            return 1;
        }
        if (pid == 0) {
            setenv("PCSF_RANK", std::to_string(r).c_str(), 1);
            setenv("PCSF_WORLD", std::to_string(n_ranks).c_str(), 1);
            setenv("PCSF_RENDEZVOUS", rdv, 1);
            execv("/proc/self/exe", argv);
            perror("error: execv");
            _exit(127);
        }
        pids[static_cast<size_t>(r)] = pid;
    }
    int status_out = 0;
    for (int left = n_ranks; left > 0; --left) {
        int st = 0;
        const pid_t done = wait(&st);
        if (done < 0) break;
        const int rc = WIFEXITED(st) ? WEXITSTATUS(st) : 128 + (WIFSIGNALED(st) ? WTERMSIG(st) : 0);
        for (pid_t& p : pids) if (p == done) p = -1;
        if (rc != 0 && status_out == 0) {
            status_out = rc;
            for (pid_t p : pids) if (p > 0) kill(p, SIGTERM);
        }
    }
    if (DIR* d = opendir(rdv)) {   // what the ranks left in the meeting place
        while (struct dirent* e = readdir(d))
            if (strcmp(e->d_name, ".") != 0 && strcmp(e->d_name, "..") != 0) unlink((std::string(rdv) + "/" + e->d_name).c_str());
        closedir(d);
    }
    rmdir(rdv);
    return status_out;
}

inline int cli_main(int argc, char** argv, const BackendFactory& make_backend, bool fast_exit = false, const GatherFactory& make_gather = nullptr)
{
    const auto t_main = std::chrono::steady_clock::now();
    auto since_main = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_main).count(); };
    if (argc < 2) return usage(argv[0]);
    const std::string cmd = argv[1];
    if (cmd == "selftest-format") {  // tests: format_trimmed<3|4> against snprintf-based print_trimmed on n random floats
        const size_t n = argc > 2 ? static_cast<size_t>(atoll(argv[2])) : 1000000;
        uint64_t state = 0x9e3779b97f4a7c15ull;
        size_t bad = 0;
        auto check = [&](float v) {
            for (int d = 3; d <= 4; ++d) {
                char want[80], got[80];
                snprintf(want, sizeof want, d == 3 ? "%.3f" : "%.4f", static_cast<double>(v));
                for (int i = static_cast<int>(strlen(want)); i >= 0; --i) {
                    if (want[i] == '.') { want[i + 1] = '0'; break; }
                    if (want[i] >= '0' && want[i] <= '9') { if (want[i] != '0') break; want[i] = 0; }
                }
                if (d == 3) format_trimmed<3>(got, v); else format_trimmed<4>(got, v);
                if (strcmp(want, got) != 0 && bad++ < 10) fprintf(stderr, "mismatch %.9g (%%.%df): want %s got %s\n", static_cast<double>(v), d, want, got);
            }
        };
        const float edge[] = {0.f, -0.f, 0.0005f, -0.0005f, 0.0015f, 0.00049999f, 1e-10f, -1e-10f, 0.125f, 0.0625f, 2.5e-4f, 9.9995f, 99999.9995f, 1e14f, 1e15f, 3e38f, -3e38f,
                              std::numeric_limits<float>::infinity(), -std::numeric_limits<float>::infinity(), std::numeric_limits<float>::quiet_NaN(), 1.0f, -1.0f, 123456.789f};
        for (float v : edge) check(v);
        for (size_t i = 0; i < n; ++i) {
            state ^= state << 13; state ^= state >> 7; state ^= state << 17;
            float v;
            if (i % 3 == 0) { const uint32_t bits = static_cast<uint32_t>(state); memcpy(&v, &bits, 4); }        // any bit pattern
            else if (i % 3 == 1) v = static_cast<float>(static_cast<int64_t>(state % 2000001) - 1000000) / 8000.f;  // many exact ties (k/8000)
            else v = static_cast<float>((static_cast<double>(state >> 11) / 9007199254740992.0 - 0.5) * 200.0);   // typical score range
            check(v);
        }
        printf("%zu values checked, %zu mismatches\n", n + sizeof edge / sizeof edge[0], bad);
        const size_t bad_rows = BranchLengthScore::selftest_presence_bits(20000);
        printf("20000 character rows classified, %zu mismatches between the vector and the scalar path\n", bad_rows);
        return (bad || bad_rows) ? 1 : 0;
    }
    if (cmd == "selftest-printed-codes") {  // tests: the smoothing pass's printed-value bins -- thresholds by bisection vs the printing code itself
        const size_t n = argc > 2 ? static_cast<size_t>(atoll(argv[2])) : 1000000;
        const std::vector<double> thr = printed_code_thresholds(effective_threads());
        size_t bad = 0, checked = 0;
        bool ascending = true;
        for (size_t j = 1; j < thr.size(); ++j) ascending = ascending && thr[j - 1] <= thr[j];
        auto check = [&](double p) {
            if (p != p) return;
            const int bin = static_cast<int>(std::upper_bound(thr.begin(), thr.end(), p) - thr.begin());   // thresholds <= p
            char want[80], got[80];
            format_trimmed<3>(want, static_cast<float>(hmm_log_odds(p)));
            const int code = PRINTED_CODE_MIN + bin;
            format_scaled<3>(got, code < 0, static_cast<uint64_t>(code < 0 ? -1 - code : code));
            ++checked;
            if ((bin != printed_code(p) - PRINTED_CODE_MIN || strcmp(want, got) != 0) && bad++ < 10)
                fprintf(stderr, "mismatch p = %.17g: bin %d, code %d, text %s vs %s\n", p, bin, printed_code(p), want, got);
        };
        for (double t : thr) {   // both sides of every step
            if (!(t <= 1.0)) continue;
            check(t);
            check(std::nextafter(t, -1.0));
            check(std::nextafter(t, 2.0));
        }
        const double edge[] = {0.0, -0.0, 1.0, 0.5, std::nextafter(0.5, 0.0), std::nextafter(0.5, 1.0), 1e-15, 9.99e-16, 1.0 - 1e-15, 1e-300, 5e-324, -1e-3, 1.0000001, 2.0};
        for (double p : edge) check(p);
        uint64_t state = 0x9e3779b97f4a7c15ull;
        for (size_t i = 0; i < n; ++i) {
            state ^= state << 13; state ^= state >> 7; state ^= state << 17;
            const double u = static_cast<double>(state >> 11) / 9007199254740992.0;
            if (i % 3 == 0) check(u);                                       // anywhere
            else if (i % 3 == 1) check(std::pow(10.0, -17.0 * u));          // towards 0
            else check(1.0 - std::pow(10.0, -17.0 * u));                    // towards 1
        }
        printf("%zu thresholds (ascending: %s), %zu posteriors checked, %zu mismatches\n", thr.size(), ascending ? "yes" : "NO", checked, bad);
        return (bad || !ascending) ? 1 : 0;
    }
    if (cmd == "dump-alignments") {  // diagnostics/tests: what the MAF reader yields (build-tracks semantics unless --blocks 1)
        bool blocks = false, matrix = false;
        size_t jobs = 1;
        int j = 2;
        for (; j + 1 < argc && strncmp(argv[j], "--", 2) == 0; j += 2) {
            if (std::string(argv[j]) == "--blocks") blocks = parse_bool(argv[j + 1]);
            else if (std::string(argv[j]) == "--jobs") jobs = static_cast<size_t>(atoll(argv[j + 1]));  // byte-range jobs, concatenated in order
            else if (std::string(argv[j]) == "--matrix") matrix = parse_bool(argv[j + 1]);   // measure() + fill() into a matrix instead of next()
            else return usage(argv[0]);
        }
        if (argc - j < 2) return usage(argv[0]);
        try {
            const std::string prefix = argv[j];
            std::string al = dirname_of(prefix) + "/species_aliases.tsv";
            if (FILE* f = fopen(al.c_str(), "r")) fclose(f);
            else al.clear();
            const Model model = load_model(prefix, al);
            MafFile file(argv[j + 1]);
            for (size_t job = 0; job < std::max<size_t>(jobs, 1); ++job) {
                const size_t nj = std::max<size_t>(jobs, 1);
                MafReader rd(file, file.size() * job / nj, file.size() * (job + 1) / nj, model.name_to_leaf, model.n_leaves, !blocks, false);
                Alignment aln;
                if (matrix) {   // the job's alignments measured first, then parsed side by side into one matrix (what build-tracks does)
                    std::vector<AlignmentSpan> spans;
                    AlignmentSpan sp;
                    while (rd.measure(sp)) spans.push_back(sp);
                    size_t stride = 7;   // columns that belong to no alignment, left and right: nothing may be written there
                    std::vector<size_t> col0;
                    for (const AlignmentSpan& a : spans) {
                        col0.push_back(stride);
                        stride += a.len + 3;
                    }
                    std::vector<char> m(static_cast<size_t>(model.n_leaves) * stride, '#');
                    for (size_t k = spans.size(); k-- > 0;) rd.fill(spans[k], m.data() + col0[k], stride);   // any order
                    size_t untouched = 0;
                    for (char c : m) untouched += c == '#';
                    size_t expect = static_cast<size_t>(model.n_leaves) * stride;
                    for (const AlignmentSpan& a : spans) expect -= static_cast<size_t>(model.n_leaves) * a.len;
                    if (untouched != expect) {
                        fprintf(stderr, "error: fill() wrote outside its alignments' columns (%zu padding bytes left of %zu)\n", untouched, expect);
                        return 1;
                    }
                    for (size_t k = 0; k < spans.size(); ++k) {
                        const AlignmentSpan& a = spans[k];
                        printf("%s\t%" PRIu64 "\t%zu\t%c\t%" PRIu64, a.chrom.c_str(), a.start_pos, static_cast<size_t>(a.len), a.strand, a.chrom_len);
                        for (int l = 0; l < model.n_leaves; ++l) printf("\t%.*s", static_cast<int>(a.len), m.data() + static_cast<size_t>(l) * stride + col0[k]);
                        printf("\n");
                    }
                    continue;
                }
                while (rd.next(aln)) {
                    printf("%s\t%" PRIu64 "\t%zu\t%c\t%" PRIu64, aln.chrom.c_str(), aln.start_pos, aln.length(), aln.strand, aln.chrom_len);
                    for (const auto& r : aln.seqs) printf("\t%s", r.c_str());
                    printf("\n");
                }
            }
        } catch (const std::exception& e) {
            fprintf(stderr, "error: %s\n", e.what());
            return 1;
        }
        return 0;
    }
    if (cmd == "dump-model") {  // diagnostics/tests: the flattened tree the scoring core receives
        std::string species, mapping;
        int j = 2;
        for (; j + 1 < argc && strncmp(argv[j], "--", 2) == 0; j += 2) {
            if (std::string(argv[j]) == "--species") species = argv[j + 1];
            else if (std::string(argv[j]) == "--mapping") mapping = argv[j + 1];
            else return usage(argv[0]);
        }
        if (argc - j < 1) return usage(argv[0]);
        try {
            const std::string prefix = argv[j];
            std::string al = dirname_of(prefix) + "/species_aliases.tsv";
            if (FILE* f = fopen(al.c_str(), "r")) fclose(f);
            else al.clear();
            const Model model = load_model(prefix, al, species, mapping);
            for (int v = 0; v < model.n_nodes; ++v)
                printf("%d\t%d\t%d\t%d\t%.17g\t%.9g\t%s\n", v, model.child1[v], model.child2[v], model.parent[v], model.branch_len_d[v],
                       static_cast<double>(model.branch_len[v]), model.label[v].c_str());
            std::vector<std::pair<std::string, int>> names(model.name_to_leaf.begin(), model.name_to_leaf.end());
            std::sort(names.begin(), names.end());
            for (const auto& kv : names) printf("name\t%s\t%d\n", kv.first.c_str(), kv.second);  // row names a MAF may use -> leaf
        } catch (const std::exception& e) {
            fprintf(stderr, "error: %s\n", e.what());
            return 1;
        }
        return 0;
    }
    if (cmd == "pack-maf") {
        std::string species, mapping;
        int threads = 0, j = 2;
        for (; j + 1 < argc && strncmp(argv[j], "--", 2) == 0; j += 2) {
            if (std::string(argv[j]) == "--species") species = argv[j + 1];
            else if (std::string(argv[j]) == "--mapping") mapping = argv[j + 1];
            else if (std::string(argv[j]) == "--threads") threads = atoi(argv[j + 1]);
            else return usage(argv[0]);
        }
        if (argc - j < 3) return usage(argv[0]);
        try {
            const std::string prefix = argv[j];
            std::string al = dirname_of(prefix) + "/species_aliases.tsv";
            if (FILE* f = fopen(al.c_str(), "r")) fclose(f);
            else al.clear();
            const Model model = load_model(prefix, al, species, mapping);
            const PackStats st = pack_maf(model, argv[j + 1], argv[j + 2], threads);
            fprintf(stderr, "%s -> %s: %zu alignments, %zu reference bases, %" PRIu64 " -> %" PRIu64 " bytes, %.3f s (%.2f GB/s of text)\n", argv[j + 1], argv[j + 2],
                    st.alignments, st.ref_bases, st.bytes_in, st.bytes_out, st.seconds, static_cast<double>(st.bytes_in) / st.seconds / 1e9);
        } catch (const std::exception& e) {
            fprintf(stderr, "error: %s\n", e.what());
            return 1;
        }
        return 0;
    }
    if (cmd != "build-tracks" && cmd != "score-msa" && cmd != "smooth-tracks") return usage(argv[0]);
    BuildTracksParams bt;
    std::string coding_exons, hmm_params, hmm_matrix;
    uint64_t genome_length = 0;
    ScoreMsaParams sm;
    std::string aliases, species, mapping, context_model;
    std::vector<int> devices{0};
    bool devices_given = false;
    int contexts_per_device = 0;
    int n_ranks = 0;   // --ranks
    int i = 2;
    for (; i < argc && strncmp(argv[i], "--", 2) == 0; i += 2) {
        if (i + 1 >= argc) return usage(argv[0]);
        const std::string k = argv[i], v = argv[i + 1];
        if (k == "--output") bt.output_dir = sm.output_dir = v;
        // the reference reads this option with get_bool: anything but 1/true/one means 0.0
        else if (k == "--power-threshold") bt.power_threshold = parse_bool(v) ? 1.0f : 0.0f;
        else if (k == "--output-raw-phylo") bt.write_raw = parse_bool(v);
        else if (k == "--output-power") bt.write_power = parse_bool(v);
        else if (k == "--output-phylo") bt.write_smooth = parse_bool(v);
        else if (k == "--output-regions") bt.write_regions = parse_bool(v);
        else if (k == "--genome-length") genome_length = strtoull(v.c_str(), nullptr, 10);
        else if (k == "--coding-exons") coding_exons = v;
        else if (k == "--hmm-params") hmm_params = v;
        else if (k == "--hmm-matrix") hmm_matrix = v;
        else if (k == "--threads") bt.threads = atoi(v.c_str());
        else if (k == "--batch-sites") bt.batch_sites = static_cast<size_t>(strtoull(v.c_str(), nullptr, 10));    // codon-frames per backend call
        else if (k == "--window-bytes") bt.window_bytes = static_cast<size_t>(strtoull(v.c_str(), nullptr, 10));  // MAF bytes per pipeline step
        else if (k == "--contexts-per-device") contexts_per_device = atoi(v.c_str());  // build-tracks; 0 = what the backend suggests
        else if (k == "--host-bls") bt.host_bls = parse_bool(v);  // branch-length scores on the host instead of the scoring backend
        else if (k == "--species-aliases") aliases = v;
        else if (k == "--species") species = v;
        else if (k == "--mapping") mapping = v;
        else if (k == "--context-model") context_model = v;   // smooth-tracks: a model to make a scoring context with, for the device recurrences
        else if (k == "--table-leaves") setenv("PCSF_TABLE_LEAVES", v.c_str(), 1);   // how far the scoring contexts' subtree tables go (2, 3 or 4 leaves; include/phylocsf_amd.h)
        else if (k == "--table-bytes") setenv("PCSF_TABLE_BYTES", v.c_str(), 1);     // ... and the device memory their rows may take
        else if (k == "--ranks") n_ranks = atoi(v.c_str());
        else if (k == "--gather-root-text") bt.gather_root_text = parse_bool(v);   // rank 0's own text through the gather too (a rehearsal on one GPU)
        else if (k == "--device") { devices = {atoi(v.c_str())}; devices_given = true; }
        else if (k == "--devices") {
            devices_given = true;  // one scoring context per listed GPU; batches / block ranges are dealt to them
            devices.clear();
            const char* q = v.c_str();
            char* end = nullptr;
            for (long x = strtol(q, &end, 10); q != end; x = strtol(q, &end, 10)) {
                devices.push_back(static_cast<int>(x));
                q = *end == ',' ? end + 1 : end;
            }
            if (devices.empty()) return usage(argv[0]);
        }
        else if (k == "--verbose") bt.verbose = sm.verbose = parse_bool(v);
        else if (k == "--strategy") {
            const std::string s = lower(v);
            if (s == "fixed") sm.strategy = ALGO_FIXED;
            else if (s == "mle") sm.strategy = ALGO_MLE;
            else if (s == "omega") sm.strategy = ALGO_OMEGA;
            else if (s == "fixed_mean") sm.strategy = ALGO_FIXED_MEAN;
            else return usage(argv[0]);
        } else if (k == "--comp-phylo") sm.comp_phylo = parse_bool(v);
        else if (k == "--comp-anc") sm.comp_anc = parse_bool(v);
        else if (k == "--comp-bls") sm.comp_bls = parse_bool(v);
        else {
            fprintf(stderr, "unknown option %s\n", k.c_str());
            return usage(argv[0]);
        }
    }
    if (cmd == "smooth-tracks" && !bt.write_smooth && !bt.write_regions) bt.write_smooth = true;
    if (argc - i < (cmd == "smooth-tracks" ? 1 : 2)) return usage(argv[0]);
    // one process per GPU: the command line as given is the launcher; its copies find their rank in the environment
    int rank = 0, world = 0;
    std::string rendezvous;
    if (n_ranks != 0) {
        if (cmd != "build-tracks" || n_ranks < 1) {
            fprintf(stderr, "--ranks N (N >= 1) is a build-tracks option\n");
            return usage(argv[0]);
        }
        const char* er = std::getenv("PCSF_RANK");
        const char* ew = std::getenv("PCSF_WORLD");
        const char* ed = std::getenv("PCSF_RENDEZVOUS");
        if (!er || !ew || !ed) return launch_ranks(argv, n_ranks);
        rank = atoi(er);
        world = atoi(ew);
        rendezvous = ed;
        if (world != n_ranks || rank < 0 || rank >= world) {
            fprintf(stderr, "error: PCSF_RANK / PCSF_WORLD do not fit --ranks %d\n", n_ranks);
            return 1;
        }
        // this rank's GPU: the rank-th of --devices (taken round the list), or GPU `rank`
        devices = {devices_given ? devices[static_cast<size_t>(rank) % devices.size()] : rank};
        contexts_per_device = 1;
        // the ranks share the host: each takes its part of the threads the process group may use
        if (bt.threads <= 0) bt.threads = std::max(2, effective_threads() / world);
    }
    try {
        {
            if (cmd == "score-msa" ? sm.strategy == ALGO_FIXED_MEAN : (bt.write_smooth || bt.write_regions)) {
                HmmParams hp;
                bool have_matrix = false;
                if (!hmm_matrix.empty()) {  // 4 initial + 16 row-major transition probabilities of a hand-made HMM
                    std::vector<double> v;
                    const char* q = hmm_matrix.c_str();
                    char* end = nullptr;
                    for (double x = strtod(q, &end); q != end; x = strtod(q, &end)) {
                        v.push_back(x);
                        q = *end == ',' ? end + 1 : end;
                    }
                    if (v.size() != 20) throw std::runtime_error("--hmm-matrix expects 20 comma-separated numbers");
                    for (int a = 0; a < 4; ++a) bt.hmm.init[a] = v[a];
                    for (int a = 0; a < 16; ++a) bt.hmm.trans[a / 4][a % 4] = v[4 + a];
                    have_matrix = true;
                } else if (!hmm_params.empty()) {
                    double v[8];
                    if (sscanf(hmm_params.c_str(), "%lf,%lf,%lf,%lf,%lf,%lf,%lf,%lf", v, v + 1, v + 2, v + 3, v + 4, v + 5, v + 6, v + 7) != 8)
                        throw std::runtime_error("--hmm-params expects eight comma-separated numbers");
                    hp.coding_prior = v[0];
                    hp.coding_length_in_codons = v[1];
                    for (int q = 0; q < 3; ++q) {
                        hp.non_coding_weights[q] = v[2 + q];
                        hp.non_coding_lengths_in_codons[q] = v[5 + q];
                    }
                } else if (genome_length != 0 && !coding_exons.empty()) {
                    hp = estimate_hmm_params(coding_exons, static_cast<uint32_t>(genome_length));  // uint32_t like the reference
                    fprintf(stderr, "HMM parameters: %.17g,%.17g,%.17g,%.17g,%.17g,%.17g,%.17g,%.17g\n", hp.coding_prior, hp.coding_length_in_codons,
                            hp.non_coding_weights[0], hp.non_coding_weights[1], hp.non_coding_weights[2], hp.non_coding_lengths_in_codons[0],
                            hp.non_coding_lengths_in_codons[1], hp.non_coding_lengths_in_codons[2]);
                } else {
                    throw std::runtime_error("for smoothened tracks / regions (--output-phylo, --output-regions) and --strategy fixed_mean provide --genome-length and --coding-exons (or --hmm-params)");
                }
                if (!have_matrix) bt.hmm = make_coding_hmm(hp);
                sm.hmm = bt.hmm;
            }
            if (cmd == "smooth-tracks") {
                const int threads = bt.threads > 0 ? bt.threads : effective_threads();
                // The pass needs no model; the device recurrences need a scoring context, and a context is made for a model: with
                // --context-model <name or prefix> (any model will do) the pass runs as build-tracks runs it, otherwise as host code.
                std::unique_ptr<ScoringBackend> ctx_backend;
                if (!context_model.empty()) ctx_backend = make_backend(load_model(context_model, aliases, "", ""), devices[0]);
                if (ctx_backend && ctx_backend->device_hmm()) smooth_tracks_device(argv[i], bt.hmm, bt.write_smooth, bt.write_regions, threads, *ctx_backend);
                else smooth_tracks(argv[i], bt.hmm, bt.write_smooth, bt.write_regions, /*keep_raw=*/true, threads);
                return 0;
            }
        }
        const std::string prefix = argv[i++];
        if (aliases.empty()) {  // convention: species_aliases.tsv next to the model files
            const std::string cand = dirname_of(prefix) + "/species_aliases.tsv";
            if (FILE* f = fopen(cand.c_str(), "r")) {
                fclose(f);
                aliases = cand;
            }
        }
        const Model model = load_model(prefix, aliases, species, mapping);
        const double t_model = since_main();
        if (!species.empty()) fprintf(stderr, "model reduced to %d species\n", model.n_leaves);
        // OMEGA drives four HIP streams per device at once (two groups of blocks, each with a stream for the evaluations that wait for
        // new eigen-systems); the runtime maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4, some taken by the context's
        // own streams), and streams that share a queue serialise behind each other's kernels.  Eight queues: configs[2] OMEGA 1.63 ->
        // 1.47 s; MLE, one stream per group, is 8 % FASTER on four, so only OMEGA asks.  Must happen before the first HIP call.
        if (cmd == "score-msa" && sm.strategy == ALGO_OMEGA) setenv("GPU_MAX_HW_QUEUES", "8", /*overwrite=*/0);
        std::vector<std::unique_ptr<ScoringBackend>> owned;
        std::vector<ScoringBackend*> backends;
        // build-tracks: a second context on the same GPU puts the transfers of one call under the kernels of the other (2-4 % on long
        // runs).  Its set-up costs 0.1-0.2 s, so it is only made for inputs of 768 MB or more (0.45 GB of MAF: 0.19 s with one context,
        // 0.20 s with two; 0.9 GB: 0.36 and 0.34 s), and on a thread of its own: the first context starts on the first window meanwhile
        // (profiles/r02/startup_profile.txt).
        uint64_t input_total = 0;
        for (int j = i; j < argc; ++j) {
            struct stat sb;
            if (stat(argv[j], &sb) == 0) input_total += static_cast<uint64_t>(sb.st_size);
        }
        for (size_t k = 0; k < devices.size(); ++k) {
            const int d = devices[k];
            if (k == 0) owned.push_back(make_backend(model, d));   // owns the staging memory: needed at once
            else owned.push_back(std::unique_ptr<ScoringBackend>(new DeferredBackend([&model, &make_backend, d] { return make_backend(model, d); },
                                                                                     owned[0]->device_bls(), owned[0]->contexts_per_device(), owned[0]->device_wig())));
            backends.push_back(owned.back().get());
            const int wanted = contexts_per_device > 0 ? contexts_per_device : (input_total >= (uint64_t(3) << 28) ? owned[0]->contexts_per_device() : 1);
            const int per = cmd == "build-tracks" ? wanted : 1;
            for (int c = 1; c < per; ++c) {
                owned.push_back(std::unique_ptr<ScoringBackend>(new DeferredBackend([&model, &make_backend, d] { return make_backend(model, d); },
                                                                                     owned[0]->device_bls(), owned[0]->contexts_per_device(), owned[0]->device_wig())));
                backends.push_back(owned.back().get());
            }
        }
        // start-up, before any input is read: the first context pays for the HIP runtime and the code object
        fprintf(stderr, "start-up seconds: options + model files %.3f, first scoring context %.3f (%zu more being set up beside the run)\n", t_model,
                since_main() - t_model, backends.size() - 1);
        std::unique_ptr<RankGather> gather;
        if (world > 0) {
            if (!make_gather) throw std::runtime_error("--ranks: this command line has no gather");
            gather = make_gather(*backends[0], rank, world, rendezvous, bt.threads > 0 ? bt.threads : effective_threads());
            bt.ranks = gather.get();
        }
        bool first = true;
        for (; i < argc; ++i) {
            if (cmd == "build-tracks") {
                const BuildTracksStats st = build_tracks(model, argv[i], bt, backends, !first);
                if (world > 0) fprintf(stderr, "[rank %d of %d, device %d] gather %.3f s (communicator set-up %.3f s) | ", rank, world, devices[0], st.gather_seconds, gather->setup_seconds());
                fprintf(stderr, "%s: %zu alignments, %zu reference bases, %zu codon-frames, %zu backend calls\n", argv[i], st.alignments,
                        st.ref_bases, st.codon_frames, st.backend_calls);
                fprintf(stderr, "  seconds: total %.3f; busy per stage (the stages overlap): MAF read %.3f, BLS %.3f, batch packing %.3f | scoring backend %.3f | wig output %.3f | smoothing %.3f\n",
                        st.total_seconds, st.read_seconds, st.bls_seconds, st.extract_seconds, st.backend_seconds, st.write_seconds, st.smooth_seconds);
            } else {
                ScoreMsaStats st;
                const std::string out = score_msa(model, argv[i], sm, backends, &st);
                fprintf(stderr, "%s -> %s\n", argv[i], out.c_str());
                fprintf(stderr, "  %zu blocks, %zu codon columns; seconds: total %.3f (MAF read + translation + branch-length scores %.3f, scoring backend %.3f, output %.3f)\n",
                        st.blocks, st.codon_columns, st.total_seconds, st.read_seconds, st.scoring_seconds, st.write_seconds);
            }
            first = false;
        }
        // Every output file is closed by now.  What would follow is the tear-down of the scoring contexts and of the HIP runtime --
        // ~0.07 s of a process whose whole run on a 1 Mb alignment takes 0.07 s (profiles/r02/startup_profile.txt) -- which the
        // operating system does as well when the process ends.  PCSF_FULL_TEARDOWN=1 keeps the orderly exit (leak checkers, profilers
        // that write their files at exit).
        const char* preload = std::getenv("LD_PRELOAD");
        // Any preloaded library may be a tool that writes its results at exit (rocprofv3, rocprof-sys, coverage): take the orderly exit then.
        const bool profiled = std::getenv("ROCP_TOOL_LIBRARIES") != nullptr || (preload && preload[0] != '\0');
        if (fast_exit && !profiled && !std::getenv("PCSF_FULL_TEARDOWN")) {
            if (fflush(nullptr) != 0) {   // a full disk or a closed pipe behind stdout / stderr must not end in status 0
                perror("error: flushing the output streams");
                _exit(1);
            }
            _exit(0);
        }
    } catch (const std::exception& e) {
        fprintf(stderr, "error: %s\n", e.what());
        return 1;
    }
    return 0;
}

}  // namespace pcsf_host
