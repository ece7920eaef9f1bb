// drivers.hpp -- build-tracks and score-msa on top of a scoring backend.
//
// The reference calls run_tracks() once per alignment x frame and run() once per block from
// inside an OpenMP loop (src/phylocsf++build_tracks.hpp:88-232, src/phylocsf++score_msa.hpp:94-220).
// Here the host gathers many alignment-frames / blocks into one leaf-major codon matrix and hands
// it to the backend in one call (the C-ABI of include/phylocsf_amd.h); output text and file
// layout follow the reference (PhyloCSFpower.wig, PhyloCSFRaw{+,-}{1,2,3}.wig, <maf>.scores).
#pragma once

#include <sys/uio.h>
#include <sys/stat.h>
#include <unistd.h>

#include <array>
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <exception>
#include <functional>
#include <future>
#include <limits>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "hmm.hpp"
#include "parallel.hpp"
#include "maf_reader.hpp"
#include "packed_reader.hpp"
#include "model_io.hpp"
#include "tracks.hpp"

namespace pcsf_host {

enum Algorithm { ALGO_MLE = 0, ALGO_FIXED = 1, ALGO_OMEGA = 2 };  // algorithm_t, src/run.hpp:16-21
constexpr int ALGO_FIXED_MEAN = 3;  // score-msa --strategy FIXED_MEAN: mean of the HMM-smoothed per-codon FIXED scores

// What the drivers need from a scoring implementation; HipBackend (backend_hip.hpp) forwards to
// the C-ABI, the test-only OracleBackend (oracle/oracle_cli.cpp) to the CPU oracle.
struct ScoringBackend {
    virtual ~ScoringBackend() = default;
    // memory for the batch matrices handed to score_alignments_6frames (the HIP backend returns page-locked memory)
    virtual char* staging_alloc(size_t bytes) { return new char[bytes]; }
    virtual void staging_free(char* p) { delete[] p; }
    // run_tracks: leaf-major codons -> per-codon decibans
    virtual void score_sites_fixed(const uint8_t* codons, size_t n_sites, size_t leaf_stride, double* out) = 0;
    // run: CSR batch of blocks -> (phylo, anc) floats, NaN on model failure
    virtual void score_blocks(int algo, const uint8_t* codons, const uint64_t* offsets, size_t n_blocks, size_t leaf_stride,
                              bool compute_anc, uint32_t seed, float* phylo, float* anc) = 0;
    // score_blocks(algo, ...) is coming: what can be set up meanwhile (the HIP backend's device queues, pcsf_prepare_block_scoring)
    virtual void prepare_blocks(int /*algo*/) {}
    // the smoothing pass is over: device memory kept for it between calls may go
    virtual void hmm_release() {}
    // The MLE / OMEGA engine of the HIP backend has two pruning kernels that round differently, and by default picks one per LIST it is
    // handed (pcsf_block_kernel).  score_msa cuts a file into one list per backend, so it asks ONCE for the whole file and pins the answer
    // on every backend: the scores of a file do not depend on how many devices share it (src/phylocsf++score_msa.hpp:115 "make sure that
    // parallelization does not affect results").  0 = nothing to choose (the CPU oracle backend).
    virtual int choose_block_kernel(const uint8_t* /*codons*/, const uint64_t* /*offsets*/, size_t /*n_blocks*/, size_t /*leaf_stride*/) { return 0; }
    virtual void set_block_kernel(int /*kernel*/) {}
    // the six frames of a batch of alignments: nucleotide rows [n_leaves][row_stride], alignment k in columns
    // [col0[k], col0[k]+len[k]); frame f of alignment k gets n_codons[6k+f] scores at out[offsets[6k+f]...] in FORWARD
    // genome order.  Default: translate on the host and call score_sites_fixed (what the CPU oracle backend uses);
    // the HIP backend overrides it with the device front end.
    // `out` has room for out_capacity values (2 per column is always enough) and, like `nucl`, preferably comes from
    // staging_alloc; returns the number of values written.
    // true when score_alignments_6frames can also fill bls_out[row_stride] with the per-base branch-length scores of the
    // batch's columns (and then checks the characters itself); otherwise the driver computes them on the host
    virtual bool device_bls() const { return false; }
    // how many contexts per device pay off in build-tracks (the HIP backend: two, their kernels take turns on the GPU
    // while the transfers of one call run under the kernels of the other)
    virtual int contexts_per_device() const { return 1; }
    // build-tracks is about to send batches of up to max_columns alignment columns: a backend may set up whatever its
    // first call would otherwise pay for (the HIP backend: device scratch, kernel code loading).  Called on a thread of
    // its own while the first window is parsed; the backend's first batch waits for it.
    virtual void prepare_tracks(size_t /*max_columns*/) {}
    // false while a backend is still being set up (DeferredBackend): build-tracks deals its batches to the others meanwhile
    virtual bool ready() const { return true; }
    // The smoothing pass's recurrences for a batch of runs of raw scores (hmm.hpp: scaled forward-backward, Viterbi): coding-state
    // emissions[n], run r = positions [run_offsets[r], run_offsets[r + 1]) -> posterior[n]; with back / last also the Viterbi back
    // pointers (2 bits per state and position) and every run's last rescaled vector.  The HIP backend runs them on the device
    // (pcsf_hmm_smooth_runs); a backend without it leaves the pass to the host code.
    virtual bool device_hmm() const { return false; }
    virtual void hmm_smooth_runs(const Hmm& /*h*/, const double* /*emissions*/, const uint64_t* /*run_offsets*/, size_t /*n_runs*/, double* /*posterior*/,
                                 uint8_t* /*back*/, double* /*last*/)
    {
        throw std::runtime_error("hmm_smooth_runs: this backend does not smooth tracks");
    }
    // ... with, instead of the posterior, bins[n] = how many of the ascending thresholds are <= it (NaN: listed in exc_index / exc_value,
    // returns how many there were)
    virtual size_t hmm_smooth_runs_binned(const Hmm& /*h*/, const double* /*emissions*/, const uint64_t* /*run_offsets*/, size_t /*n_runs*/, const double* /*thresholds*/,
                                          size_t /*n_thresholds*/, uint16_t* /*bins*/, uint64_t* /*exc_index*/, double* /*exc_value*/, size_t /*exc_capacity*/,
                                          uint8_t* /*back*/, double* /*last*/)
    {
        throw std::runtime_error("hmm_smooth_runs_binned: this backend does not smooth tracks");
    }
    // true when the backend can hand back the TEXT of a batch's seven wig files instead of its scores and per-base branch-length scores
    // (formatted where the scores are: build-tracks then only copies the text down and appends it to the files)
    virtual bool device_wig() const { return false; }
    // The batch of score_alignments_6frames, scored and formatted: bytes[7] = the text per file (0: PhyloCSFpower.wig, 1..6: the frames);
    // chroms[k]: alignment k's chromosome name.  text[text_capacity] (staging memory of wig_text_estimate() bytes) receives the seven texts
    // back to back if they fit; otherwise the text stays with the backend until fetch_wig_text copies it (or the next call replaces it).
    // false: a value came up that the backend does not print -- score and format this batch the usual way.
    virtual bool score_alignments_6frames_wig(const char*, size_t, const std::vector<uint64_t>&, const std::vector<uint32_t>&, const std::vector<uint64_t>&,
                                              const std::vector<uint64_t>&, const std::vector<const char*>&, float, bool, bool, char*, size_t, uint64_t*, size_t*)
    {
        throw std::runtime_error("score_alignments_6frames_wig: this backend does not format tracks");
    }
    virtual size_t wig_text_estimate(size_t, size_t) { throw std::runtime_error("wig_text_estimate: this backend does not format tracks"); }
    virtual void fetch_wig_text(char* const*) { throw std::runtime_error("fetch_wig_text: this backend does not format tracks"); }
    // build-tracks as one process per GPU (RankGather below): the text of the last score_alignments_6frames_wig call stays where the
    // backend formatted it, under the caller's key (the HIP backend: in its GPU's memory, pcsf_wig_retain_last)
    virtual void retain_wig_text(uint64_t /*order*/) { throw std::runtime_error("retain_wig_text: this backend does not keep text"); }
    virtual size_t score_alignments_6frames(int n_leaves, const char* nucl, size_t row_stride, const std::vector<uint64_t>& col0,
                                            const std::vector<uint32_t>& len, const std::vector<uint64_t>& start_pos,
                                            const std::vector<uint64_t>& chrom_len, double* out, size_t out_capacity, std::vector<uint64_t>& offsets,
                                            std::vector<uint32_t>& n_codons, double* bls_out)
    {
        if (bls_out) throw std::runtime_error("score_alignments_6frames: this backend does not compute branch-length scores");
        const size_t n = col0.size();
        offsets.assign(n * 6, 0);
        n_codons.assign(n * 6, 0);
        uint64_t total = 0;
        for (size_t k = 0; k < n; ++k)
            for (int f = 0; f < 6; ++f) {
                const FrameView v = frame_view(start_pos[k], chrom_len[k], len[k], f < 3 ? '+' : '-', static_cast<unsigned>(f % 3 + 1));
                offsets[k * 6 + f] = total;
                n_codons[k * 6 + f] = static_cast<uint32_t>(v.n_codons);
                total += v.n_codons;
            }
        if (total > out_capacity) throw std::runtime_error("score_alignments_6frames: output buffer too small");
        if (total == 0) return 0;
        std::vector<uint8_t> codons(static_cast<size_t>(n_leaves) * total);
        std::vector<std::string> rows(n_leaves);
        for (size_t k = 0; k < n; ++k) {
            for (int l = 0; l < n_leaves; ++l) rows[l].assign(nucl + static_cast<size_t>(l) * row_stride + col0[k], len[k]);
            for (int f = 0; f < 6; ++f) {
                if (f == 3) reverse_complement(rows);
                const FrameView v = frame_view(start_pos[k], chrom_len[k], len[k], f < 3 ? '+' : '-', static_cast<unsigned>(f % 3 + 1));
                extract_codons(rows, v.skip, v.n_codons, codons.data(), total, offsets[k * 6 + f]);
            }
        }
        score_sites_fixed(codons.data(), total, total, out);
        for (size_t k = 0; k < n; ++k)  // '-' strand codons were scored right-to-left
            for (int f = 3; f < 6; ++f) std::reverse(out + offsets[k * 6 + f], out + offsets[k * 6 + f] + n_codons[k * 6 + f]);
        return total;
    }
};

// A backend whose construction runs on a thread of its own (a second scoring context costs 0.1-0.2 s of HIP set-up that a
// short run should not wait for): calls block until it exists, ready() tells the dealer not to wait.
class DeferredBackend : public ScoringBackend {
public:
    DeferredBackend(std::function<std::unique_ptr<ScoringBackend>()> make, bool device_bls, int contexts_per_device, bool device_wig = false)
        : made_(std::async(std::launch::async, std::move(make)).share()), device_bls_(device_bls), device_wig_(device_wig), contexts_(contexts_per_device) {}
    ~DeferredBackend() override
    {
        try {
            made_.wait();
        } catch (...) {
        }
    }
    bool ready() const override { return made_.wait_for(std::chrono::seconds(0)) == std::future_status::ready; }
    bool device_bls() const override { return device_bls_; }  // the same model on the same kind of device as the first backend
    int contexts_per_device() const override { return contexts_; }
    char* staging_alloc(size_t bytes) override { return get().staging_alloc(bytes); }
    void staging_free(char* p) override { get().staging_free(p); }
    void prepare_tracks(size_t max_columns) override { get().prepare_tracks(max_columns); }
    void score_sites_fixed(const uint8_t* codons, size_t n_sites, size_t leaf_stride, double* out) override
    {
        get().score_sites_fixed(codons, n_sites, leaf_stride, out);
    }
    void score_blocks(int algo, const uint8_t* codons, const uint64_t* offsets, size_t n_blocks, size_t leaf_stride, bool compute_anc, uint32_t seed,
                      float* phylo, float* anc) override
    {
        get().score_blocks(algo, codons, offsets, n_blocks, leaf_stride, compute_anc, seed, phylo, anc);
    }
    void prepare_blocks(int algo) override
    {
        if (ready()) get().prepare_blocks(algo);   // a backend still being made is not waited for
    }
    int choose_block_kernel(const uint8_t* codons, const uint64_t* offsets, size_t n_blocks, size_t leaf_stride) override
    {
        return get().choose_block_kernel(codons, offsets, n_blocks, leaf_stride);
    }
    void set_block_kernel(int kernel) override { get().set_block_kernel(kernel); }
    void hmm_release() override { get().hmm_release(); }
    bool device_hmm() const override { return get().device_hmm(); }
    void hmm_smooth_runs(const Hmm& h, const double* emissions, const uint64_t* run_offsets, size_t n_runs, double* posterior, uint8_t* back, double* last) override
    {
        get().hmm_smooth_runs(h, emissions, run_offsets, n_runs, posterior, back, last);
    }
    size_t hmm_smooth_runs_binned(const Hmm& h, const double* emissions, const uint64_t* run_offsets, size_t n_runs, const double* thresholds, size_t n_thresholds,
                                  uint16_t* bins, uint64_t* exc_index, double* exc_value, size_t exc_capacity, uint8_t* back, double* last) override
    {
        return get().hmm_smooth_runs_binned(h, emissions, run_offsets, n_runs, thresholds, n_thresholds, bins, exc_index, exc_value, exc_capacity, back, last);
    }
    size_t score_alignments_6frames(int n_leaves, const char* nucl, size_t row_stride, const std::vector<uint64_t>& col0, const std::vector<uint32_t>& len,
                                    const std::vector<uint64_t>& start_pos, const std::vector<uint64_t>& chrom_len, double* out, size_t out_capacity,
                                    std::vector<uint64_t>& offsets, std::vector<uint32_t>& n_codons, double* bls_out) override
    {
        return get().score_alignments_6frames(n_leaves, nucl, row_stride, col0, len, start_pos, chrom_len, out, out_capacity, offsets, n_codons, bls_out);
    }

    bool device_wig() const override { return device_wig_; }   // like device_bls: what the first backend answered
    bool score_alignments_6frames_wig(const char* nucl, size_t row_stride, const std::vector<uint64_t>& col0, const std::vector<uint32_t>& len,
                                      const std::vector<uint64_t>& start_pos, const std::vector<uint64_t>& chrom_len, const std::vector<const char*>& chroms,
                                      float power_threshold, bool want_power, bool want_raw, char* text, size_t text_capacity, uint64_t* bytes,
                                      size_t* n_codon_frames) override
    {
        return get().score_alignments_6frames_wig(nucl, row_stride, col0, len, start_pos, chrom_len, chroms, power_threshold, want_power, want_raw, text, text_capacity,
                                                  bytes, n_codon_frames);
    }
    size_t wig_text_estimate(size_t n_columns, size_t n_alignments) override { return get().wig_text_estimate(n_columns, n_alignments); }
    void fetch_wig_text(char* const* dst) override { get().fetch_wig_text(dst); }

private:
    ScoringBackend& get() const { return *made_.get(); }   // rethrows what the construction threw
    std::shared_future<std::unique_ptr<ScoringBackend>> made_;
    bool device_bls_;
    bool device_wig_;
    int contexts_;
};

// build-tracks as one process ("rank") per GPU: rank r scores the r-th of `world` byte ranges of the input -- the ranks stand in for
// the reference's byte-range jobs (src/phylocsf++build_tracks.hpp:88) -- and the text of ranks 1.. reaches rank 0, which owns the
// files, through ONE gather in rank order: the reference's merge_job_output_files (:27-53, :246-259).  The HIP backend's gather is
// RCCL between the GPUs (backend_hip.hpp RcclGather: the text never visits a sender's host); the test-only oracle command line has
// a file-based one (oracle/oracle_cli.cpp).
struct RankGather {
    virtual ~RankGather() = default;
    virtual int rank() const = 0;
    virtual int world() const = 0;
    virtual double setup_seconds() const { return 0.0; }   // what bringing the ranks together took (RCCL: the communicator), once known
    // a new input file begins: nothing is kept yet
    virtual void begin() = 0;
    // a batch of this rank's text (order = the batch's index in the rank's stream): where the backend's last wig call left it ...
    virtual void keep_backend_text(ScoringBackend& backend, uint64_t order) = 0;
    // ... or formatted on the host: bytes[7], text[f] = file f's text of the batch
    virtual void keep_host_text(uint64_t order, const char* const* text, const uint64_t* bytes) = 0;
    // Collective, once per input file.  On rank 0: sink(file, text, n) receives the senders' text, for every file in rank order (files
    // may be served from different threads at once, one file never from two).  include_root: rank 0 kept its text too (a rehearsal).
    virtual void gather(bool include_root, const std::function<void(int file, const char* text, size_t n)>& sink) = 0;
};

struct BuildTracksParams {
    std::string output_dir;          // empty: next to the alignment file
    float power_threshold = 0.1f;    // TrackCLIParams::phylo_threshold
    bool write_raw = true, write_power = true;
    bool write_smooth = false, write_regions = false;  // --output-phylo, --output-regions (both need `hmm`)
    Hmm hmm;
    size_t batch_sites = 4u << 20;   // codon-frames gathered per backend call
    bool verbose = false;
    bool host_bls = false;           // branch-length scores on the host even when the backend offers them (--host-bls)
    int threads = 0;                 // host threads for ingest / BLS / text output (0 = all the process may use)
    size_t window_bytes = 0;                  // MAF bytes parsed (in parallel) per pipeline step; 0: from the input's size (32 MiB for inputs of 256 MiB and more)
    RankGather* ranks = nullptr;              // --ranks: this process is one of several, each on a GPU of its own
    bool gather_root_text = false;            // ... and rank 0's own text takes the gather's path too instead of going straight to the files
};

// CPUs this process may actually use: hardware threads, capped by the cgroup CPU quota (a container on a
// 256-thread host may own 16) and by PCSF_THREADS
inline int effective_threads()
{
    int n = static_cast<int>(std::thread::hardware_concurrency());
    if (n <= 0) n = 1;
    if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char quota[32] = {0};
        long period = 0;
        if (fscanf(f, "%31s %ld", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0) {
            const long q = (atol(quota) + period - 1) / period;
            if (q >= 1 && q < n) n = static_cast<int>(q);
        }
        fclose(f);
    }
    if (const char* e = getenv("PCSF_THREADS")) {
        const int v = atoi(e);
        if (v >= 1) n = v;
    }
    return n;
}

inline std::string dirname_of(const std::string& path)
{
    const size_t p = path.find_last_of('/');
    return p == std::string::npos ? "./" : path.substr(0, p);
}

inline FILE* open_out(const std::string& path, const char* mode)
{
    FILE* f = fopen(path.c_str(), mode);
    if (!f) throw std::runtime_error("cannot create file '" + path + "'");
    return f;
}

// An output file that is closed on every path out of its scope.  Unless commit() was called, a file this run CREATED
// (mode "w") is removed again: a failed build-tracks must not leave truncated wig files behind for a later --append or
// smooth-tracks run to read.
class OutFile {
public:
    OutFile() = default;
    OutFile(const std::string& path, const char* mode) { open(path, mode); }
    OutFile(const OutFile&) = delete;
    OutFile& operator=(const OutFile&) = delete;
    ~OutFile()
    {
        if (!f_) return;
        fclose(f_);
        if (!committed_ && created_) unlink(path_.c_str());
    }
    void open(const std::string& path, const char* mode)
    {
        f_ = open_out(path, mode);
        path_ = path;
        created_ = mode[0] == 'w';
    }
    void commit()  // success: flush, close, keep
    {
        if (!f_) return;
        committed_ = true;
        const int rc = fclose(f_);
        f_ = nullptr;
        if (rc != 0) throw std::runtime_error("write failed: '" + path_ + "'");
    }
    FILE* get() const { return f_; }
    explicit operator bool() const { return f_ != nullptr; }

private:
    FILE* f_ = nullptr;
    std::string path_;
    bool created_ = false, committed_ = false;
};

struct BuildTracksStats {
    size_t alignments = 0, ref_bases = 0, codon_frames = 0, backend_calls = 0;
    double backend_seconds = 0.0;                       // inside ScoringBackend::score_sites_fixed (H2D + kernel + D2H)
    double read_seconds = 0.0, bls_seconds = 0.0, extract_seconds = 0.0, write_seconds = 0.0, smooth_seconds = 0.0, total_seconds = 0.0;
    double gather_seconds = 0.0;                        // one process per GPU: the collective at the end of a file (waiting for the slowest rank included)
};

struct StopWatch {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    double lap()
    {
        const auto t1 = std::chrono::steady_clock::now();
        const double d = std::chrono::duration<double>(t1 - t0).count();
        t0 = t1;
        return d;
    }
};

// appends text to a file through a large buffer
struct WigSink {
    FILE* f;
    std::vector<char> buf;
    explicit WigSink(FILE* file) : f(file) { buf.reserve(1 << 20); }
    char* room(size_t n)
    {
        if (buf.size() + n > buf.capacity()) flush();
        const size_t at = buf.size();
        buf.resize(at + n);
        return buf.data() + at;
    }
    void shrink(size_t unused) { buf.resize(buf.size() - unused); }
    void flush()
    {
        if (!buf.empty() && fwrite(buf.data(), 1, buf.size(), f) != buf.size()) throw std::runtime_error("write failed");
        buf.clear();
    }
    void header(const std::string& chrom, int64_t pos)
    {
        char* p = room(chrom.size() + 64);
        const int n = snprintf(p, chrom.size() + 64, "fixedStep chrom=%s start=%" PRId64 " step=3 span=3\n", chrom.c_str(), pos);
        shrink(chrom.size() + 64 - static_cast<size_t>(n));
    }
    template <int D>
    void value(float v)
    {
        char* p = room(72);
        const size_t n = format_trimmed<D>(p, v);
        p[n] = '\n';
        shrink(72 - (n + 1));
    }
};


// a piece of wig text formatted in memory (build_tracks stage 3); the buffer is kept between batches (fresh memory costs page faults)
struct TextPiece {
    std::vector<char> buf;
    size_t len = 0;
    void start(size_t expected)
    {
        len = 0;
        if (buf.size() < expected) buf.resize(expected);
    }
    char* room(size_t n)
    {
        if (len + n > buf.size()) buf.resize(std::max(buf.size() * 2, len + n));
        return buf.data() + len;
    }
    void header(const std::string& chrom, int64_t pos)
    {
        char* p = room(chrom.size() + 64);
        len += static_cast<size_t>(snprintf(p, chrom.size() + 64, "fixedStep chrom=%s start=%" PRId64 " step=3 span=3\n", chrom.c_str(), pos));
    }
    template <int D>
    void value(float v)
    {
        char* p = room(72);
        const size_t n = format_trimmed<D>(p, v);
        p[n] = '\n';
        len += n + 1;
    }
};

inline void remove_raw_tracks(const std::string& out_dir)
{
    for (int fi = 0; fi < 6; ++fi)
        unlink((out_dir + "/PhyloCSFRaw" + std::string(1, fi < 3 ? '+' : '-') + std::to_string(fi % 3 + 1) + ".wig").c_str());
}

// The smoothing pass (reference src/phylocsf++build_tracks.hpp:262-366): every PhyloCSFRaw<strand><frame>.wig of
// out_dir is read back as text, cut into maximal contiguous runs, and each run goes through the HMM:
// PhyloCSF<strand><frame>.wig gets the log-odds of the coding posterior per codon ("%.3f", trimmed),
// PhyloCSF<strand><frame>Regions.bed the coding stretches of the Viterbi path.  Six frames, six threads.
// `done` (optional): frames whose smoothed files are already committed (by a device pass that failed later on) are left alone.  The raw
// tracks are only removed once ALL six frames are finished: a failure half-way leaves every raw track in place.
inline void smooth_tracks(const std::string& out_dir, const Hmm& hmm, bool write_smooth, bool write_regions, bool keep_raw, int threads,
                          const std::array<bool, 6>* done = nullptr)
{
    parallel_for(6, threads, [&](size_t fi) {
        if (done && (*done)[fi]) return;
        const char strand = fi < 3 ? '+' : '-';
        const std::string tag = std::string(1, strand) + std::to_string(fi % 3 + 1);
        const std::string raw = out_dir + "/PhyloCSFRaw" + tag + ".wig";
        OutFile wig_file, bed_file;  // closed (and, on failure, removed) on every path out of this job
        if (write_smooth) wig_file.open(out_dir + "/PhyloCSF" + tag + ".wig", "w");
        if (write_regions) bed_file.open(out_dir + "/PhyloCSF" + tag + "Regions.bed", "w");
        FILE* f_wig = wig_file.get();
        FILE* f_bed = bed_file.get();
        std::vector<double> post, work;
        std::vector<BedRegion> regions;
        {
            std::unique_ptr<WigSink> out(f_wig ? new WigSink(f_wig) : nullptr);
            for_each_wig_run(raw, [&](const WigRun& run) {
                hmm_coding_posterior(hmm, run.scores, post, work);
                if (out) {
                    out->header(run.chrom, static_cast<int64_t>(run.start));
                    for (double p : post) out->value<3>(static_cast<float>(hmm_log_odds(p)));
                }
                if (f_bed) {
                    regions.clear();
                    hmm_regions(hmm_viterbi(hmm, run.scores), post, static_cast<uint32_t>(run.start), regions);
                    for (const BedRegion& r : regions)
                        fprintf(f_bed, "%s\t%" PRIu32 "\t%" PRIu32 "\t%s:%" PRIu32 "-%" PRIu32 "\t0\t%c\t%" PRIu32 "\t%" PRIu32 "\t%" PRIu32 ",%" PRIu32 ",%" PRIu32 "\n",
                                run.chrom.c_str(), r.start, r.end, run.chrom.c_str(), r.start + 1, r.end, strand, r.start, r.end, r.color, r.color, r.color);
                }
            });
            if (out) out->flush();
        }
        wig_file.commit();
        bed_file.commit();
    });
    if (!keep_raw) remove_raw_tracks(out_dir);
}

// What the smoothed track prints for a coding posterior p is a monotone step function of p: log-odds (hmm_log_odds), narrowed to
// float, rounded to three decimals (format_trimmed<3>), with "-0.0" and "0.0" apart.  printed_code numbers the steps: value x 1000
// for values >= +0, -1 - |value| x 1000 for negative ones ("-0.0" = -1, "-15.0" = -15001) -- increasing with p.
inline int printed_code(double p)
{
    const double x = static_cast<double>(static_cast<float>(hmm_log_odds(p)));
#if defined(__x86_64__)
    const long long scaled = _mm_cvtsd_si64(_mm_set_sd(std::fabs(x) * 1000.0));   // as format_trimmed rounds
#else
    const long long scaled = static_cast<long long>(std::nearbyint(std::fabs(x) * 1000.0));
#endif
    return std::signbit(x) ? -1 - static_cast<int>(scaled) : static_cast<int>(scaled);
}
constexpr int PRINTED_CODE_MIN = -15001, PRINTED_CODE_MAX = 15000;
// thresholds[j], j = 0 .. MAX - MIN - 1: the smallest double p in [0, 1] with printed_code(p) >= MIN + 1 + j, by bisection over the
// doubles with this host's own libm (so the steps are where THIS host's log10 puts them); +infinity where no p reaches the code.
// The number of thresholds <= p is then printed_code(p) - MIN for every p that is not NaN.
inline std::vector<double> printed_code_thresholds(int threads)
{
    const size_t n = static_cast<size_t>(PRINTED_CODE_MAX - PRINTED_CODE_MIN);
    std::vector<double> thr(n);
    constexpr size_t PARTS = 256;   // a task per threshold would spend its time handing out tasks
    parallel_for(PARTS, threads, [&](size_t part) {
        auto as_double = [](uint64_t b) { double d; memcpy(&d, &b, 8); return d; };
        const double one = 1.0;
        uint64_t one_bits;
        memcpy(&one_bits, &one, 8);
        const int code0 = printed_code(0.0), code1 = printed_code(1.0);
        uint64_t floor_bits = 0;   // thresholds ascend: the one before bounds the search from below
        for (size_t j = n * part / PARTS; j < n * (part + 1) / PARTS; ++j) {
            const int want = PRINTED_CODE_MIN + 1 + static_cast<int>(j);
            if (code0 >= want) { thr[j] = 0.0; continue; }
            if (code1 < want) { thr[j] = std::numeric_limits<double>::infinity(); continue; }
            uint64_t lo = floor_bits, hi = one_bits;   // bit patterns of non-negative doubles order like the doubles; code(lo) < want <= code(hi)
            if (printed_code(as_double(lo)) >= want) lo = 0;   // (cannot happen for ascending codes; keeps the invariant whatever libm does)
            while (hi - lo > 1) {
                const uint64_t mid = lo + (hi - lo) / 2;
                if (printed_code(as_double(mid)) >= want) hi = mid;
                else lo = mid;
            }
            thr[j] = as_double(hi);
            floor_bits = lo;   // code(lo) < want <= the next code wanted
        }
    });
    return thr;
}

// The same pass with the recurrences on the device (backend.hmm_smooth_runs).  Tracks are taken in groups of up to ~64 M values (all six
// of a 10 Mb chromosome, one at a time for a large one): (1) the text of the group's raw tracks is parsed on all host threads into runs
// merged like the reference's wig reader (for_each_wig_run); (2) the coding emissions pow(10, score / 10) come from the host's libm --
// scores are multiples of 0.001, so a table over [-200, 200] is computed once and every value is a look-up; anything that is not such
// a multiple goes through pow() itself; (3) ONE backend call returns the coding posterior and the Viterbi back pointers of all runs
// of the group; (4) log-odds, text and regions are made on all host threads, the files appended side by side.  Output files are
// byte-identical to smooth_tracks' (tests/test_gpu_hmm.py).
//
// The pass NEVER removes a raw track: it marks in `done` the frames whose smoothed files it has committed, so that a caller whose later
// group failed (device memory on a large genome) can hand the REMAINING frames to the host pass -- which must neither truncate the
// finished files nor look for raw tracks that are gone -- and removes the raw tracks itself once everything is written.
inline void smooth_tracks_device(const std::string& out_dir, const Hmm& hmm, bool write_smooth, bool write_regions, int threads,
                                 ScoringBackend& backend, std::array<bool, 6>* done = nullptr)
{
    constexpr int TABLE_HALF = 200000;   // scores in [-200, 200] decibans by look-up
    // tests: PCSF_HMM_GROUP_VALUES cuts the six tracks into several groups, PCSF_HMM_FAIL_GROUP=k makes group k throw before it writes
    const char* gv = std::getenv("PCSF_HMM_GROUP_VALUES");
    const size_t GROUP_VALUES = gv ? std::max<size_t>(1, strtoull(gv, nullptr, 10)) : size_t(64) << 20;
    const char* fg = std::getenv("PCSF_HMM_FAIL_GROUP");
    const int fail_group = fg ? atoi(fg) : -1;
    int group_index = 0;
    const bool verbose = std::getenv("PCSF_HMM_STATS") != nullptr;
    const std::vector<double> thresholds = printed_code_thresholds(threads);
    std::vector<double> table(2 * TABLE_HALF + 1);
    parallel_for(64, threads, [&](size_t part) {
        for (int k = static_cast<int>(part) - TABLE_HALF; k <= TABLE_HALF; k += 64) table[k + TABLE_HALF] = hmm_emit(0, static_cast<double>(k) / 1000.0);
    });
    struct Piece {   // a stretch of a file, cut at line starts: its headers and values in order
        std::vector<std::pair<size_t, std::pair<std::string, uint64_t>>> headers;   // (values before the header within the piece, (chrom, start))
        std::vector<double> values;
        size_t at = 0;   // position of the piece's first value in the group's arrays
    };
    struct Run {
        std::string chrom;
        uint64_t start;
    };
    struct Frame {
        int fi = 0;
        std::string raw;
        std::unique_ptr<MafFile> text;
        std::vector<Piece> pieces;
        std::vector<Run> runs;
        size_t run0 = 0;   // index of the frame's first run in the group's run list
        OutFile wig_file, bed_file;   // closed (and, on failure, removed) on every path out
    };
    auto tag_of = [](int fi) { return std::string(1, fi < 3 ? '+' : '-') + std::to_string(fi % 3 + 1); };
    for (int f0 = 0; f0 < 6;) {
        StopWatch phase;
        std::vector<std::unique_ptr<Frame>> group;
        size_t estimate = 0;
        for (; f0 < 6; ++f0) {
            std::unique_ptr<Frame> fr(new Frame);
            fr->fi = f0;
            fr->raw = out_dir + "/PhyloCSFRaw" + tag_of(f0) + ".wig";
            fr->text.reset(new MafFile(fr->raw));
            const size_t est = fr->text->size() / 6;
            if (!group.empty() && estimate + est > GROUP_VALUES) break;
            estimate += est;
            group.push_back(std::move(fr));
        }
        if (group_index++ == fail_group) throw std::runtime_error("PCSF_HMM_FAIL_GROUP: injected failure");
        // (1) parse: pieces of every file of the group
        std::vector<std::pair<size_t, size_t>> piece_jobs;   // (frame, piece)
        for (size_t g = 0; g < group.size(); ++g) {
            const size_t n_pieces = std::max<size_t>(1, std::min<size_t>(static_cast<size_t>(threads) * 2, group[g]->text->size() >> 16));
            group[g]->pieces.resize(n_pieces);
            for (size_t pi = 0; pi < n_pieces; ++pi) piece_jobs.push_back({g, pi});
        }
        parallel_for(piece_jobs.size(), threads, [&](size_t j) {
            Frame& fr = *group[piece_jobs[j].first];
            const size_t pi = piece_jobs[j].second, n_pieces = fr.pieces.size();
            const char* base = fr.text->data();
            const size_t size = fr.text->size();
            auto line_start = [&](size_t at) {   // first line start at or after `at`
                if (at == 0) return size_t(0);
                const void* nl = memchr(base + at - 1, '\n', size - (at - 1));
                return nl ? static_cast<size_t>(static_cast<const char*>(nl) - base) + 1 : size;
            };
            size_t at = line_start(size * pi / n_pieces);
            const size_t end = line_start(size * (pi + 1) / n_pieces);
            Piece& pc = fr.pieces[pi];
            pc.values.reserve((end - at) / 5 + 16);
            while (at < end) {
                const char* line = base + at;
                const void* nl = memchr(line, '\n', size - at);
                const size_t len = nl ? static_cast<size_t>(static_cast<const char*>(nl) - line) : size - at;
                at += len + 1;
                if (len == 0) continue;
                if (line[0] == 'f') {
                    char chrom[256] = {0};
                    uint64_t start = 0;
                    const std::string l(line, len);
                    if (sscanf(l.c_str(), "fixedStep chrom=%255s start=%" SCNu64, chrom, &start) != 2) throw std::runtime_error("malformed fixedStep line in " + fr.raw);
                    pc.headers.push_back({pc.values.size(), {chrom, start}});
                    continue;
                }
                // a plain decimal with at most three decimals is k / 1000 exactly rounded -- what strtod returns; anything else: strtod
                size_t i = 0;
                const bool neg = line[0] == '-';
                if (neg) ++i;
                uint64_t ip = 0, frac = 0;
                size_t nd = 0, nf = 0;
                while (i < len && line[i] >= '0' && line[i] <= '9' && nd < 9) { ip = ip * 10 + static_cast<uint64_t>(line[i] - '0'); ++i; ++nd; }
                bool plain = nd > 0;
                if (plain && i < len && line[i] == '.') {
                    ++i;
                    while (i < len && line[i] >= '0' && line[i] <= '9' && nf < 3) { frac = frac * 10 + static_cast<uint64_t>(line[i] - '0'); ++i; ++nf; }
                    for (size_t pad = nf; pad < 3; ++pad) frac *= 10;
                }
                plain = plain && i == len;
                if (plain) {
                    const double v = static_cast<double>(ip * 1000 + frac) / 1000.0;
                    pc.values.push_back(neg ? -v : v);
                } else {
                    pc.values.push_back(strtod(std::string(line, len).c_str(), nullptr));
                }
            }
        });
        // runs: a header that continues the stretch before it (same chromosome, start == start of the run + 3 x values so far) starts nothing
        std::vector<uint64_t> run_offsets;
        std::vector<std::pair<size_t, size_t>> run_owner;   // (frame, run within the frame)
        size_t n = 0;
        for (size_t g = 0; g < group.size(); ++g) {
            Frame& fr = *group[g];
            fr.run0 = run_offsets.size();
            const size_t frame_at = n;
            auto run_begin = [&]() { return run_offsets.back(); };
            for (Piece& pc : fr.pieces) {
                pc.at = n;
                for (const auto& hd : pc.headers) {
                    const uint64_t at = pc.at + hd.first;
                    const bool open = !fr.runs.empty();
                    if (open && at > run_begin() && fr.runs.back().chrom == hd.second.first && fr.runs.back().start + 3 * (at - run_begin()) == hd.second.second) continue;
                    if (open && at == run_begin()) {   // a header right behind a header: the earlier one had no values
                        fr.runs.back() = Run{hd.second.first, hd.second.second};
                        continue;
                    }
                    if (!open && at > frame_at) throw std::runtime_error("values before the first fixedStep line in " + fr.raw);
                    fr.runs.push_back(Run{hd.second.first, hd.second.second});
                    run_offsets.push_back(at);
                    run_owner.push_back({g, fr.runs.size() - 1});
                }
                n += pc.values.size();
            }
            if (fr.runs.empty() && n > frame_at) throw std::runtime_error("values before the first fixedStep line in " + fr.raw);
            if (!fr.runs.empty() && run_begin() == n) {   // a last header without values
                fr.runs.pop_back();
                run_offsets.pop_back();
                run_owner.pop_back();
            }
            if (write_smooth) fr.wig_file.open(out_dir + "/PhyloCSF" + tag_of(fr.fi) + ".wig", "w");
            if (write_regions) fr.bed_file.open(out_dir + "/PhyloCSF" + tag_of(fr.fi) + "Regions.bed", "w");
        }
        run_offsets.push_back(n);
        const size_t n_runs = run_owner.size();
        const double t_parse = phase.lap();
        double t_emit = 0, t_dev = 0, t_text = 0, t_regions = 0;
        if (n_runs > 0) {
            // (2) emissions (the arrays are not value-initialised: 17 bytes per value that are written before they are read)
            std::unique_ptr<double[]> emit(new double[n]), last(write_regions ? new double[n_runs * 4] : nullptr);
            std::unique_ptr<uint16_t[]> bins(new uint16_t[n]);
            std::unique_ptr<uint8_t[]> back(write_regions ? new uint8_t[n] : nullptr);
            parallel_for(piece_jobs.size(), threads, [&](size_t j) {
                Piece& pc = group[piece_jobs[j].first]->pieces[piece_jobs[j].second];
                const std::vector<double>& v = pc.values;
                double* dst = emit.get() + pc.at;
                for (size_t i = 0; i < v.size(); ++i) {
                    const double scaled = v[i] * 1000.0;
                    const bool in_table = std::fabs(scaled) <= TABLE_HALF;
                    const long k = in_table ? std::lrint(scaled) : 0;
                    dst[i] = (in_table && static_cast<double>(k) / 1000.0 == v[i]) ? table[k + TABLE_HALF] : hmm_emit(0, v[i]);
                }
                std::vector<double>().swap(pc.values);
            });
            t_emit = phase.lap();
            // (3) the recurrences
            // Runs of a million values and more (a contiguous chromosome without the reader's break points) stay on the host, one thread
            // each, beside the device call: a recurrence is sequential along a run, a device thread takes ~10x a host core's time per step,
            // and for this model the device's chunks of such a run seldom meet their neighbours' states (the classes' memory is tens of
            // thousands of positions long) -- hmm_abi.cpp would evaluate them again one after the other.
            size_t host_run_min = size_t(1) << 20;
            if (const char* v = std::getenv("PCSF_HMM_HOST_RUN")) host_run_min = std::max<size_t>(1, strtoull(v, nullptr, 10));
            std::vector<size_t> host_runs, dev_runs;
            for (size_t r = 0; r < n_runs; ++r) (run_offsets[r + 1] - run_offsets[r] >= host_run_min ? host_runs : dev_runs).push_back(r);
            std::vector<std::vector<uint32_t>> host_path(n_runs);   // Viterbi paths of the runs done on the host
            std::vector<uint64_t> exc_index(16);
            std::vector<double> exc_value(16);
            size_t n_exc = 0;
            // One backend call holds ~60 bytes of device memory per value (emissions, four forward values, posterior, back pointers, bins):
            // a frame's runs go to the backend in batches of at most GROUP_VALUES values (runs are independent; a single run is never
            // longer than that here: runs of a million values and more stay on the host, below).
            auto binned_in_batches = [&](const double* em, const uint64_t* offs, size_t nr, uint16_t* bn, uint8_t* bk, double* ls) -> size_t {
                size_t total_exc = 0;
                for (size_t r0 = 0; r0 < nr;) {
                    size_t r1 = r0 + 1;
                    while (r1 < nr && offs[r1 + 1] - offs[r0] <= GROUP_VALUES) ++r1;
                    const uint64_t base = offs[r0];
                    std::vector<uint64_t> sub(offs + r0, offs + r1 + 1);
                    for (uint64_t& v : sub) v -= base;
                    total_exc += backend.hmm_smooth_runs_binned(hmm, em + base, sub.data(), r1 - r0, thresholds.data(), thresholds.size(), bn + base, exc_index.data(),
                                                                exc_value.data(), exc_index.size(), bk ? bk + base : nullptr, ls ? ls + 4 * r0 : nullptr);
                    r0 = r1;
                }
                return total_exc;
            };
            if (host_runs.empty()) {
                n_exc = binned_in_batches(emit.get(), run_offsets.data(), n_runs, bins.get(), back.get(), last.get());
            } else {
                std::vector<uint64_t> d_off{0};
                for (size_t r : dev_runs) d_off.push_back(d_off.back() + (run_offsets[r + 1] - run_offsets[r]));
                const size_t nd = d_off.back();
                std::unique_ptr<double[]> d_emit(new double[std::max<size_t>(nd, 1)]), d_last(write_regions ? new double[std::max<size_t>(dev_runs.size(), 1) * 4] : nullptr);
                std::unique_ptr<uint16_t[]> d_bins(new uint16_t[std::max<size_t>(nd, 1)]);
                std::unique_ptr<uint8_t[]> d_back(write_regions ? new uint8_t[std::max<size_t>(nd, 1)] : nullptr);
                parallel_for(dev_runs.size(), threads, [&](size_t k) {
                    const size_t r = dev_runs[k];
                    memcpy(d_emit.get() + d_off[k], emit.get() + run_offsets[r], (run_offsets[r + 1] - run_offsets[r]) * sizeof(double));
                });
                auto device_part = std::async(std::launch::async, [&]() -> size_t {
                    if (dev_runs.empty()) return 0;
                    return binned_in_batches(d_emit.get(), d_off.data(), dev_runs.size(), d_bins.get(), d_back.get(), d_last.get());
                });
                try {
                    parallel_for(host_runs.size(), threads, [&](size_t k) {
                        const size_t r = host_runs[k], b = run_offsets[r], len = run_offsets[r + 1] - b;
                        std::vector<double> run_post(len), work;
                        hmm_coding_posterior_emissions(hmm, emit.get() + b, len, run_post.data(), work);
                        for (size_t i = 0; i < len; ++i) bins[b + i] = run_post[i] != run_post[i] ? static_cast<uint16_t>(std::signbit(run_post[i]) ? 0xfffeu : 0xffffu) : static_cast<uint16_t>(printed_code(run_post[i]) - PRINTED_CODE_MIN);
                        if (write_regions) host_path[r] = hmm_viterbi_emissions(hmm, emit.get() + b, len);
                    });
                } catch (...) {
                    device_part.wait();   // it works on this frame's arrays
                    throw;
                }
                n_exc = device_part.get();
                parallel_for(dev_runs.size(), threads, [&](size_t k) {
                    const size_t r = dev_runs[k], b = run_offsets[r], len = run_offsets[r + 1] - b;
                    memcpy(bins.get() + b, d_bins.get() + d_off[k], len * sizeof(uint16_t));
                    if (write_regions) {
                        memcpy(back.get() + b, d_back.get() + d_off[k], len);
                        memcpy(last.get() + 4 * r, d_last.get() + 4 * k, 4 * sizeof(double));
                    }
                });
            }
            (void)n_exc;   // positions whose posterior is NaN carry bin 0xffff / 0xfffe (sign) and are printed as the host pass prints such a NaN: the list is not needed
            emit.reset();
            t_dev = phase.lap();
            // (4) text: pieces of runs on all threads, every file's pieces appended in order
            if (write_smooth) {
                struct Job { size_t run, a, b; };
                std::vector<Job> jobs;
                size_t PIECE = 32768;
                if (const char* v = std::getenv("PCSF_WIG_PIECE")) PIECE = std::max<size_t>(1, strtoull(v, nullptr, 10));   // tests: seams inside a run
                for (size_t r = 0; r < n_runs; ++r)
                    for (size_t a = run_offsets[r]; a < run_offsets[r + 1]; a += PIECE) jobs.push_back(Job{r, a, std::min<size_t>(a + PIECE, run_offsets[r + 1])});
                constexpr size_t WAVE = 8192;   // jobs formatted before their text is written (bounds the memory held as text)
                std::vector<TextPiece> out(std::min(jobs.size(), WAVE));
                for (size_t j0 = 0; j0 < jobs.size(); j0 += WAVE) {
                    const size_t j1 = std::min(jobs.size(), j0 + WAVE);
                    parallel_for(j1 - j0, threads, [&](size_t j) {
                        const Job& jb = jobs[j0 + j];
                        const Run& run = group[run_owner[jb.run].first]->runs[run_owner[jb.run].second];
                        TextPiece& o = out[j];
                        o.start((jb.b - jb.a) * 9 + 256);
                        if (jb.a == run_offsets[jb.run]) o.header(run.chrom, static_cast<int64_t>(run.start));
                        for (size_t i = jb.a; i < jb.b; ++i) {
                            const uint16_t bin = bins[i];
                            if (bin >= 0xfffeu) {   // a NaN posterior (a raw score that is "nan"): printed as the host pass prints a NaN of that sign
                                o.value<3>(static_cast<float>(hmm_log_odds(std::copysign(std::numeric_limits<double>::quiet_NaN(), bin == 0xfffeu ? -1.0 : 1.0))));
                                continue;
                            }
                            const int code = PRINTED_CODE_MIN + static_cast<int>(bin);
                            char* p = o.room(72);
                            const size_t len = format_scaled<3>(p, code < 0, static_cast<uint64_t>(code < 0 ? -1 - code : code));
                            p[len] = '\n';
                            o.len += len + 1;
                        }
                    });
                    parallel_for(group.size(), threads, [&](size_t g) {
                        FILE* f_wig = group[g]->wig_file.get();
                        for (size_t j = 0; j < j1 - j0; ++j)
                            if (run_owner[jobs[j0 + j].run].first == g && out[j].len && fwrite(out[j].buf.data(), 1, out[j].len, f_wig) != out[j].len)
                                throw std::runtime_error("write failed");
                    });
                }
            }
            t_text = phase.lap();
            if (write_regions) {
                std::vector<std::vector<BedRegion>> regions(n_runs);
                parallel_for(n_runs, threads, [&](size_t r) {
                    const size_t b = run_offsets[r], len = run_offsets[r + 1] - b;
                    std::vector<uint32_t> path;
                    if (!host_path[r].empty()) {
                        path.swap(host_path[r]);
                    } else {
                        path.assign(len, 0);
                        const double* prev = last.get() + 4 * r;
                        uint32_t best = 0;  // hmm_viterbi's rule: the first maximum; all zero: state 0
                        for (uint32_t i = 1; i < 4; ++i)
                            if (prev[i] > prev[best]) best = i;
                        if (!(prev[best] > 0.0)) best = 0;
                        path[len - 1] = best;
                        for (size_t pos = len - 1; pos-- > 0;) path[pos] = (back[b + pos + 1] >> (2 * path[pos + 1])) & 3u;
                    }
                    const std::vector<double> run_post(len, 0.0);   // a region's `prob` is computed by the reference and never printed
                    hmm_regions(path, run_post, static_cast<uint32_t>(group[run_owner[r].first]->runs[run_owner[r].second].start), regions[r]);
                });
                parallel_for(group.size(), threads, [&](size_t g) {
                    FILE* f_bed = group[g]->bed_file.get();
                    const char strand = group[g]->fi < 3 ? '+' : '-';
                    for (size_t r = 0; r < n_runs; ++r) {
                        if (run_owner[r].first != g) continue;
                        const Run& run = group[g]->runs[run_owner[r].second];
                        for (const BedRegion& rg : regions[r])
                            fprintf(f_bed, "%s\t%" PRIu32 "\t%" PRIu32 "\t%s:%" PRIu32 "-%" PRIu32 "\t0\t%c\t%" PRIu32 "\t%" PRIu32 "\t%" PRIu32 ",%" PRIu32 ",%" PRIu32 "\n",
                                    run.chrom.c_str(), rg.start, rg.end, run.chrom.c_str(), rg.start + 1, rg.end, strand, rg.start, rg.end, rg.color, rg.color, rg.color);
                    }
                });
            }
            t_regions = phase.lap();
        }
        for (auto& fr : group) {
            fr->wig_file.commit();
            fr->bed_file.commit();
            fr->text.reset();
            if (done) (*done)[static_cast<size_t>(fr->fi)] = true;
        }
        if (verbose)
            fprintf(stderr, "smooth_tracks_device, tracks %s..%s: %zu values in %zu runs; parse %.3f s, emissions %.3f, recurrences (backend) %.3f, log-odds + text %.3f, paths + regions %.3f\n",
                    tag_of(group.front()->fi).c_str(), tag_of(group.back()->fi).c_str(), n, n_runs, t_parse, t_emit, t_dev, t_text, t_regions);
    }
}

// build-tracks for one MAF file.  append: later files of a multi-file run append to the wigs.
// Stages per window of the file: (1) byte-range jobs parse the MAF and compute the per-base branch-length
// scores on all host threads; (2) the alignments, in file order, are packed into nucleotide batches and
// scored by the backend (six frames per alignment, one call per batch); (3) seven writers -- the power
// track and the six frames -- format their text in parallel and append it to their files.
inline BuildTracksStats build_tracks(const Model& model, const std::string& maf_path, const BuildTracksParams& prm,
                                     const std::vector<ScoringBackend*>& backends, bool append = false)
{
    if (backends.empty()) throw std::runtime_error("build_tracks: no scoring backend");
    ScoringBackend& backend = *backends[0];  // owner of the staging memory
    const std::string out_dir = prm.output_dir.empty() ? dirname_of(maf_path) : prm.output_dir;
    mkdir(out_dir.c_str(), 0764);
    const char* mode = append ? "a" : "w";
    // one process per GPU: this one scores the rank-th byte range; only rank 0 has files, the others keep their text for the gather
    RankGather* const ranks = prm.ranks;
    const int rank = ranks ? ranks->rank() : 0, world = ranks ? ranks->world() : 1;
    const bool owns_files = rank == 0;
    const bool keeps_text = ranks && (rank != 0 || prm.gather_root_text);   // this rank's text goes through the gather
    if (ranks && backends.size() != 1) throw std::runtime_error("build_tracks: a rank scores on one backend");
    if (ranks) ranks->begin();
    OutFile power_file, raw_file[6];  // closed on every path; files created by a failed run are removed
    if (prm.write_power && owns_files) power_file.open(out_dir + "/PhyloCSFpower.wig", mode);
    FILE* f_power = power_file.get();
    FILE* f_raw[6] = {nullptr};
    const bool want_power = prm.write_power;   // (a rank without files still makes the text)
    // the smoothing pass reads the raw tracks back (as text, like the reference), so they are written even when
    // they were not asked for, and removed afterwards
    const bool need_raw = prm.write_raw || prm.write_smooth || prm.write_regions;
    if (need_raw && owns_files)
        for (int i = 0; i < 6; ++i) {
            raw_file[i].open(out_dir + "/PhyloCSFRaw" + std::string(1, i < 3 ? '+' : '-') + std::to_string(i % 3 + 1) + ".wig", mode);
            f_raw[i] = raw_file[i].get();
        }

    struct Item {  // one alignment
        std::string chrom;
        uint64_t start_pos = 0, chrom_len = 0, col0 = 0;
        uint32_t len = 0;
        std::vector<std::string> seqs;
        const char* rows = nullptr;        // packed input: the species rows where they lie in the mapped file ([n_leaves][len]) instead of seqs
        AlignmentSpan span;                // text input, rows not needed on the host: where the alignment lies in the file (measured, not read)
        bool measured = false;
        char* dst = nullptr;               // ... and where its first column goes in its batch's matrix (row stride dst_stride)
        size_t dst_stride = 0;
        std::vector<double> bls_host;      // per-base branch-length scores when computed on the host
        const double* bls = nullptr;       // [len]: bls_host, or the alignment's columns of its batch's device result
    };
    const size_t nl = static_cast<size_t>(model.n_leaves);
    const int threads = prm.threads > 0 ? prm.threads : effective_threads();
    size_t cap = std::max<size_t>(prm.batch_sites / 2, 1);  // columns per batch: one column yields two codon-frames
    BuildTracksStats st;
    const float threshold3 = prm.power_threshold * 3;
    // the branch-length scores come from the backend (same uploaded rows as the six frames) when every backend offers that
    bool dev_bls = need_raw && !prm.host_bls;
    for (const ScoringBackend* b : backends) dev_bls = dev_bls && b->device_bls();
    // ... and so does the text of the seven files: stage 3 then only appends what stage 2 copied down
    bool dev_wig = dev_bls;
    for (const ScoringBackend* b : backends) dev_wig = dev_wig && b->device_wig();
    StopWatch total, sw;

    // Stage 3 formats a batch's text in PIECES on all host threads -- (file, alignment, range of codons) -- and then appends each file's
    // pieces in order, the seven files side by side.  (Seven tasks, one per file, left nine of sixteen threads idle and made the writers the
    // next ceiling after the parser: 0.87 s per 100 Mb.)  A piece that starts inside an alignment needs to know one thing about what came
    // before it: whether the codon just before was written (the run continues) or not (a header line comes first).
    struct PieceJob {
        int file = 0;          // 0: power track, 1..6: frames
        const Item* it = nullptr;
        const double* sc = nullptr;   // the frame's scores of this alignment
        size_t a = 0, b = 0;   // power: positions [a, b) of the alignment (a = skip mod 3); frames: codons [a, b)
    };
    size_t PIECE_CODONS = 16384;
    if (const char* v = std::getenv("PCSF_WIG_PIECE")) PIECE_CODONS = std::max<size_t>(1, strtoull(v, nullptr, 10));   // tests: seams inside an alignment
    std::vector<TextPiece> text;   // stage 3's buffers, reused from batch to batch
    auto format_piece = [&](const PieceJob& j, TextPiece& out) {
        const Item& p = *j.it;
        out.start((j.b - j.a) * (j.file == 0 ? 4 : 10) + 256);
        if (j.file == 0) {  // frame 3 of the + strand, mean of the three per-base scores
            int64_t skip = static_cast<int64_t>(3 - static_cast<int64_t>(p.start_pos)) % 3;
            if (skip < 0) skip += 3;
            if (j.a == static_cast<size_t>(skip) && static_cast<uint64_t>(skip) + 2 < p.len) out.header(p.chrom, static_cast<int64_t>(p.start_pos) + skip);
            for (size_t pos = j.a; pos < j.b && pos + 2 < p.len; pos += 3)
                out.value<4>(static_cast<float>((p.bls[pos] + p.bls[pos + 1] + p.bls[pos + 2]) / 3.0));
            return;
        }
        const int fi = j.file - 1;
        const FrameView v = frame_view(p.start_pos, p.chrom_len, p.len, fi < 3 ? '+' : '-', static_cast<unsigned>(fi % 3 + 1));
        auto kept = [&](size_t xx) {
            const size_t bpos = v.bls_offset + 3 * xx;
            return !(static_cast<float>(p.bls[bpos] + p.bls[bpos + 1] + p.bls[bpos + 2]) < threshold3);
        };
        int64_t prev = (j.a > 0 && kept(j.a - 1)) ? static_cast<int64_t>(v.start_pos + (j.a - 1) * 3) : -4;
        for (size_t xx = j.a; xx < j.b; ++xx) {
            if (!kept(xx)) continue;
            const int64_t pos = static_cast<int64_t>(v.start_pos + xx * 3);
            if (prev + 3 != pos) out.header(p.chrom, pos);
            prev = pos;
            out.value<3>(static_cast<float>(j.sc[xx]));
        }
    };
    // the pieces of one batch, file by file in the order their text goes to the file
    auto batch_pieces = [&](const std::vector<Item*>& items, const double* scores, const std::vector<uint64_t>& offsets) {
        std::vector<PieceJob> jobs;
        if (want_power)
            for (const Item* it : items) {
                int64_t skip = static_cast<int64_t>(3 - static_cast<int64_t>(it->start_pos)) % 3;
                if (skip < 0) skip += 3;
                for (size_t a = static_cast<size_t>(skip); a + 2 < it->len; a += 3 * PIECE_CODONS) {
                    PieceJob j;
                    j.file = 0; j.it = it; j.a = a; j.b = std::min<size_t>(a + 3 * PIECE_CODONS, it->len);
                    jobs.push_back(j);
                }
            }
        if (need_raw)
            for (int fi = 0; fi < 6; ++fi)
                for (size_t k = 0; k < items.size(); ++k) {
                    const Item& p = *items[k];
                    const FrameView v = frame_view(p.start_pos, p.chrom_len, p.len, fi < 3 ? '+' : '-', static_cast<unsigned>(fi % 3 + 1));
                    for (size_t a = 0; a < v.n_codons; a += PIECE_CODONS) {
                        PieceJob j;
                        j.file = fi + 1; j.it = &p; j.sc = scores + offsets[k * 6 + fi]; j.a = a; j.b = std::min<size_t>(a + PIECE_CODONS, v.n_codons);
                        jobs.push_back(j);
                    }
                }
        return jobs;
    };

    // PCSF_DIRECT_PARSE=0: text input through strings and a packing copy (rounds 1-3; A/B and tests -- same files)
    const bool direct_parse_off = [] { const char* v = std::getenv("PCSF_DIRECT_PARSE"); return v && v[0] == '0'; }();
    MafFile file(maf_path);
    // a packed alignment file (pack-maf, packed_reader.hpp) holds the reader's alignments ready-made: stage 1 only copies rows
    const bool packed = is_packed_file(file);
    const size_t input_bytes = packed ? static_cast<size_t>(PackedReader::data_bytes(file)) : file.size();
    // (a rank's byte range is cut like the reference cuts its jobs': an alignment belongs to the range its first block's "a" line starts in)
    const size_t range_begin = static_cast<size_t>(static_cast<unsigned __int128>(input_bytes) * static_cast<unsigned>(rank) / static_cast<unsigned>(world));
    const size_t range_end = static_cast<size_t>(static_cast<unsigned __int128>(input_bytes) * static_cast<unsigned>(rank + 1) / static_cast<unsigned>(world));
    const size_t range_bytes = range_end - range_begin;
    // 32 MiB steps; a short input gets about eight (and the two smaller ones in front), so that the stages overlap inside it too: the last
    // step's scoring runs with nothing beside it, and with four steps that was a fifth of a 90 MB run (0.060 -> 0.050 s)
    const size_t window = prm.window_bytes ? std::max<size_t>(prm.window_bytes, 1 << 16)
                                           : std::min<size_t>(size_t(32) << 20, std::max<size_t>(size_t(4) << 20, (range_bytes / 8 + 0xfffff) & ~size_t(0xfffff)));
    // the first windows are smaller (1/4, 1/2 of the size) so that the scoring stage starts after a short parse
    std::vector<size_t> window_begin;
    for (size_t at = range_begin, k = 0; at < range_end; ++k) {
        window_begin.push_back(at);
        at += k == 0 ? std::max<size_t>(window / 4, 1 << 16) : (k == 1 ? std::max<size_t>(window / 2, 1 << 16) : window);
    }
    window_begin.push_back(range_end);
    // batch matrices come from the backend (page-locked for the GPU) and are recycled: allocating them is slow
    struct StagingPool {
        ScoringBackend& backend;
        std::mutex lock;
        std::vector<std::pair<char*, size_t>> free_list;
        explicit StagingPool(ScoringBackend& b) : backend(b) {}
        ~StagingPool()
        {
            for (auto& f : free_list) backend.staging_free(f.first);
        }
        std::pair<char*, size_t> get(size_t bytes)
        {
            {
                std::lock_guard<std::mutex> g(lock);
                for (size_t i = 0; i < free_list.size(); ++i)
                    if (free_list[i].second >= bytes) {
                        auto f = free_list[i];
                        free_list.erase(free_list.begin() + static_cast<long>(i));
                        return f;
                    }
            }
            const size_t cap_bytes = bytes + bytes / 4;  // head room so that the next, slightly larger batch fits too
            return {backend.staging_alloc(cap_bytes), cap_bytes};
        }
        void put(std::pair<char*, size_t> f)
        {
            std::lock_guard<std::mutex> g(lock);
            free_list.push_back(f);
        }
    } pool(backend), score_pool(backend), text_pool(backend);  // batch matrices; score vectors; wig text (different sizes: kept apart)
    struct Batch {  // alignments scored by one backend call, their nucleotides packed leaf-major
        std::vector<Item*> items;
        size_t stride = 0;
        std::pair<char*, size_t> nucl{nullptr, 0};  // [nl][stride], uninitialised outside the alignments' columns
        std::vector<uint64_t> col0, start_pos, chrom_len;
        std::vector<uint32_t> len;
        std::pair<char*, size_t> score_mem{nullptr, 0};  // page-locked, from the pool
        const double* scores = nullptr;    // filled by the backend
        size_t n_scores = 0;
        std::vector<uint64_t> offsets;
        std::vector<uint32_t> n_codons;
        std::pair<char*, size_t> text_mem{nullptr, 0};  // the backend formatted the batch: the seven files' text back to back, page-locked
        uint64_t text_bytes[7] = {0, 0, 0, 0, 0, 0, 0};
        bool has_text = false;
        uint64_t order = 0;      // the batch's index in this process's stream of batches
        bool kept = false;       // one process per GPU: the backend keeps the text for the gather, nothing comes down here
    };
    struct PreparedWindow {
        std::vector<std::vector<Item>> jobs;
        std::vector<Batch> batches;
        size_t submitted = 0;  // batches handed to stage 2
        double read = 0.0, bls = 0.0, pack = 0.0;  // thread-seconds in the reader / the branch-length score; wall seconds packing
        double parse_wall = 0.0;                   // wall seconds of the parallel parse
    };
    // Stage 1 (runs one window ahead of the scoring): byte-range jobs parse the MAF and compute the branch-length
    // scores; then the alignments, in file order, are cut into batches and their rows copied into the batch matrix.
    // (Parsing window k+1 while window k is packed was measured and is slower -- 1 Mb: 0.059 -> 0.086 s, 10 Mb: 0.316 ->
    // 0.339 s: both halves work on the process's page tables and on all host threads.)
    auto prepare_window = [&](size_t wi) {
        std::unique_ptr<PreparedWindow> pw(new PreparedWindow);
        StopWatch parse_clock;
        const size_t w0 = window_begin[wi], w1 = window_begin[wi + 1];
        const size_t n_jobs = std::max<size_t>(1, std::min<size_t>(static_cast<size_t>(threads) * 4, (w1 - w0 + (1 << 16) - 1) >> 16));
        pw->jobs.resize(n_jobs);
        std::vector<double> job_read(n_jobs, 0.0), job_bls(n_jobs, 0.0);
        // text input whose rows nothing on the host needs: the jobs only MEASURE their alignments (reference rows alone), and once every
        // alignment has its columns the same readers parse the rows straight into the batch matrices -- no strings, no packing copy
        const bool direct = !packed && dev_bls && need_raw && !direct_parse_off;
        std::vector<std::unique_ptr<MafReader>> direct_readers(direct ? n_jobs : 0);
        parallel_for(n_jobs, threads, [&](size_t j) {
            const size_t b = w0 + (w1 - w0) * j / n_jobs, e = w0 + (w1 - w0) * (j + 1) / n_jobs;
            std::unique_ptr<MafReader> text_reader;
            std::unique_ptr<PackedReader> packed_reader;
            if (packed) packed_reader.reset(new PackedReader(file, b, e, model.n_leaves, row_map_hash(model)));
            else text_reader.reset(new MafReader(file, b, e, model.name_to_leaf, model.n_leaves, /*concatenate=*/true, prm.verbose && j == 0 && w0 == 0));
            BranchLengthScore bls(model);
            Alignment aln;
            StopWatch jw;
            if (direct) {
                Item it;
                while (text_reader->measure(it.span)) {
                    it.chrom = it.span.chrom;
                    it.start_pos = it.span.start_pos;
                    it.chrom_len = it.span.chrom_len;
                    it.len = it.span.len;
                    it.measured = true;
                    pw->jobs[j].push_back(it);
                }
                direct_readers[j] = std::move(text_reader);
                job_read[j] += jw.lap();
                return;
            }
            if (packed && dev_bls && need_raw) {
                // nothing of a record is needed on the host: its rows go from the mapped file straight into the batch matrix
                PackedReader::Record r;
                while (packed_reader->next_record(r)) {
                    Item it;
                    it.chrom.assign(r.chrom, r.chrom_bytes);
                    it.start_pos = r.start_pos;
                    it.chrom_len = r.chrom_len;
                    it.len = r.len;
                    it.rows = r.rows;
                    pw->jobs[j].push_back(std::move(it));
                }
                job_read[j] += jw.lap();
                return;
            }
            while (packed ? packed_reader->next(aln) : text_reader->next(aln)) {
                job_read[j] += jw.lap();
                Item it;
                it.chrom = aln.chrom;
                it.start_pos = aln.start_pos;
                it.chrom_len = aln.chrom_len;
                it.len = static_cast<uint32_t>(aln.length());
                // also validates every character (the reference exits on anything but ACGTNacgtn-.); the backend does both otherwise
                if (!dev_bls) bls.per_base(aln, &it.bls_host);
                it.seqs.swap(aln.seqs);
                pw->jobs[j].push_back(std::move(it));
                job_bls[j] += jw.lap();
            }
        });
        for (size_t j = 0; j < n_jobs; ++j) {
            pw->read += job_read[j];
            pw->bls += job_bls[j];
        }
        pw->parse_wall = parse_clock.lap();
        StopWatch pack;
        pw->batches.emplace_back();
        for (auto& job : pw->jobs)
            for (Item& it : job) {
                if (!pw->batches.back().items.empty() && pw->batches.back().stride + it.len > cap) pw->batches.emplace_back();
                Batch& bt = pw->batches.back();
                it.col0 = bt.stride;
                it.bls = it.bls_host.data();
                bt.stride += it.len;
                bt.items.push_back(&it);
            }
        if (pw->batches.back().items.empty()) pw->batches.pop_back();
        for (Batch& bt : pw->batches) {
            bt.stride = std::max(bt.stride, size_t(1));
            for (const Item* p : bt.items) {
                bt.col0.push_back(p->col0);
                bt.len.push_back(p->len);
                bt.start_pos.push_back(p->start_pos);
                bt.chrom_len.push_back(p->chrom_len);
            }
            if (!need_raw) continue;
            bt.nucl = pool.get(nl * bt.stride);
            if (direct) {   // filled below, by the jobs that measured the alignments
                for (Item* p : bt.items) {
                    p->dst = bt.nucl.first + p->col0;
                    p->dst_stride = bt.stride;
                }
                continue;
            }
            if (!bt.items.empty() && bt.items[0]->rows) {
                // packed input: one positional scatter read per record, page cache -> batch matrix.  (Copying out of the mapping instead
                // faults its pages in from sixteen threads at once, neighbours in the same page-table pages: 0.39 -> 1.1 s per 100 Mb.)
                parallel_for(bt.items.size(), threads, [&](size_t k) {
                    const Item& p = *bt.items[k];
                    if (p.len == 0) return;
                    size_t done_rows = 0;
                    std::vector<struct iovec> iov(std::min<size_t>(nl, 512));
                    while (done_rows < nl) {
                        const size_t n = std::min(iov.size(), nl - done_rows);
                        for (size_t i = 0; i < n; ++i) iov[i] = {bt.nucl.first + (done_rows + i) * bt.stride + p.col0, p.len};
                        const off_t at = static_cast<off_t>(p.rows - file.data()) + static_cast<off_t>(done_rows * p.len);
                        const ssize_t got = preadv(file.fd(), iov.data(), static_cast<int>(n), at);
                        if (got != static_cast<ssize_t>(n * p.len))   // short read or error: the mapping has the same bytes
                            for (size_t i = 0; i < n; ++i) memcpy(iov[i].iov_base, p.rows + (done_rows + i) * p.len, p.len);
                        done_rows += n;
                    }
                });
            } else {
                parallel_for(bt.items.size() * nl, threads, [&](size_t j) {
                    const Item& p = *bt.items[j / nl];
                    const size_t l = j % nl;
                    memcpy(bt.nucl.first + l * bt.stride + p.col0, p.seqs[l].data(), p.len);
                });
            }
            for (Item* p : bt.items) std::vector<std::string>().swap(p->seqs);  // the rows now live in the batch matrix
        }
        if (direct && need_raw)
            parallel_for(n_jobs, threads, [&](size_t j) {
                for (const Item& it : pw->jobs[j]) direct_readers[j]->fill(it.span, it.dst, it.dst_stride);
            });
        pw->pack = pack.lap();
        return pw;
    };
    // Stage 2: one backend call per batch, batches dealt round-robin to the backends (one per GPU; a backend takes
    // its batches in order, different backends run concurrently).  Stage 3 (behind stage 2, in batch order): seven
    // writers -- the power track and the six frames -- format the batch's text and append it to their files.
    struct WindowSlot {
        std::unique_ptr<PreparedWindow> pw;
        size_t unwritten = 0;  // batches whose text is not on file yet: the window (items, batches) must stay alive
    };
    struct InFlight {
        Batch* bt = nullptr;
        WindowSlot* ws = nullptr;
        std::shared_future<double> scored;  // seconds inside the backend call
    };
    std::deque<WindowSlot> alive;  // references to deque elements survive push_back / pop_front of other elements
    std::deque<InFlight> inflight;
    InFlight writing;
    std::future<double> writer;  // seconds the writers of `writing` took
    const size_t n_dev = backends.size();
    std::vector<std::shared_future<double>> dev_last(n_dev), dev_prepared(n_dev);
    size_t next_dev = 0;
    auto finish_writer = [&]() {
        if (!writer.valid()) return;
        st.write_seconds += writer.get();
        --writing.ws->unwritten;
        while (!alive.empty() && alive.front().unwritten == 0 && alive.front().pw->batches.size() == alive.front().pw->submitted) alive.pop_front();
    };
    auto retire_front = [&]() {  // oldest scored batch -> writers
        InFlight f = inflight.front();
        inflight.pop_front();
        st.backend_seconds += f.scored.get();
        if (need_raw) {
            ++st.backend_calls;
            st.codon_frames += f.bt->n_scores;
        }
        finish_writer();  // files are appended in batch order
        writing = f;
        writer = std::async(std::launch::async, [&, batch = f.bt]() {
            StopWatch ww;
            if (batch->kept) return ww.lap();   // the text waits for the gather where the backend formatted it
            if (batch->has_text) {   // formatted by the backend: the files side by side
                uint64_t at[8] = {0};
                for (int t = 0; t < 7; ++t) at[t + 1] = at[t] + batch->text_bytes[t];
                parallel_for(7, threads, [&](size_t t) {
                    FILE* f = t == 0 ? f_power : f_raw[t - 1];
                    const size_t n = static_cast<size_t>(batch->text_bytes[t]);
                    if (f && n && fwrite(batch->text_mem.first + at[t], 1, n, f) != n) throw std::runtime_error("write failed");
                });
                text_pool.put(batch->text_mem);
                batch->text_mem = {nullptr, 0};
                return ww.lap();
            }
            const std::vector<PieceJob> jobs = batch_pieces(batch->items, batch->scores, batch->offsets);
            if (text.size() < jobs.size()) text.resize(jobs.size());   // one writer runs at a time; the buffers keep their memory
            parallel_for(jobs.size(), threads, [&](size_t i) { format_piece(jobs[i], text[i]); });
            if (keeps_text) {   // one process per GPU: every file's pieces joined, handed to the gather under the batch's index
                std::vector<char> joined[7];
                parallel_for(7, threads, [&](size_t t) {
                    size_t n = 0;
                    for (size_t i = 0; i < jobs.size(); ++i)
                        if (jobs[i].file == static_cast<int>(t)) n += text[i].len;
                    joined[t].reserve(n);
                    for (size_t i = 0; i < jobs.size(); ++i)
                        if (jobs[i].file == static_cast<int>(t)) joined[t].insert(joined[t].end(), text[i].buf.data(), text[i].buf.data() + text[i].len);
                });
                const char* ptr[7];
                uint64_t bytes[7];
                for (int t = 0; t < 7; ++t) {
                    ptr[t] = joined[t].data();
                    bytes[t] = joined[t].size();
                }
                ranks->keep_host_text(batch->order, ptr, bytes);
            } else
            parallel_for(7, threads, [&](size_t t) {   // every file's pieces in order; the files side by side
                FILE* f = t == 0 ? f_power : f_raw[t - 1];
                if (!f) return;
                for (size_t i = 0; i < jobs.size(); ++i)
                    if (jobs[i].file == static_cast<int>(t) && text[i].len && fwrite(text[i].buf.data(), 1, text[i].len, f) != text[i].len)
                        throw std::runtime_error("write failed");
            });
            if (batch->score_mem.first) score_pool.put(batch->score_mem);
            batch->score_mem = {nullptr, 0};
            batch->scores = nullptr;
            return ww.lap();
        });
    };
    uint64_t next_order = 0;
    auto submit = [&](Batch& bt, WindowSlot* ws) {
        bt.order = next_order++;
        // round robin over the backends that exist (one still being set up -- DeferredBackend -- and its prepare_tracks are skipped)
        size_t d = next_dev % n_dev;
        for (size_t k = 0; k < n_dev; ++k) {
            const size_t cand = (next_dev + k) % n_dev;
            const bool set_up = backends[cand]->ready() && (!dev_prepared[cand].valid() || dev_prepared[cand].wait_for(std::chrono::seconds(0)) == std::future_status::ready);
            if (set_up || cand == 0) {
                d = cand;
                break;
            }
        }
        next_dev = d + 1;
        const std::shared_future<double> prev = dev_last[d];
        std::shared_future<double> task = std::async(std::launch::async, [&, b = &bt, d, prev]() -> double {
            if (prev.valid()) prev.wait();
            StopWatch bw;
            if (need_raw && dev_wig) {
                std::vector<const char*> chroms(b->items.size());
                for (size_t k = 0; k < chroms.size(); ++k) chroms[k] = b->items[k]->chrom.c_str();
                // the text is copied down behind the kernels, before its size is known: a buffer from the backend's estimate
                // (one process per GPU: no buffer -- the text stays with the backend until the gather)
                if (!keeps_text) b->text_mem = text_pool.get(backends[d]->wig_text_estimate(b->stride, b->items.size()));
                b->has_text = backends[d]->score_alignments_6frames_wig(b->nucl.first, b->stride, b->col0, b->len, b->start_pos, b->chrom_len, chroms,
                                                                        prm.power_threshold, want_power, true, b->text_mem.first, b->text_mem.second,
                                                                        b->text_bytes, &b->n_scores);
                if (b->has_text && keeps_text) {
                    ranks->keep_backend_text(*backends[d], b->order);
                    b->kept = true;
                    pool.put(b->nucl);
                    b->nucl = {nullptr, 0};
                    return bw.lap();
                }
                if (b->has_text) {
                    uint64_t total = 0;
                    for (uint64_t v : b->text_bytes) total += v;
                    if (total > b->text_mem.second) {   // unusually long lines: the text waits on the device
                        text_pool.put(b->text_mem);
                        b->text_mem = text_pool.get(static_cast<size_t>(total));
                        char* dst[7];
                        dst[0] = b->text_mem.first;
                        for (int t = 1; t < 7; ++t) dst[t] = dst[t - 1] + b->text_bytes[t - 1];
                        backends[d]->fetch_wig_text(dst);
                    }
                    pool.put(b->nucl);
                    b->nucl = {nullptr, 0};
                    return bw.lap();
                }
                if (b->text_mem.first) text_pool.put(b->text_mem);
                b->text_mem = {nullptr, 0};
            }
            if (need_raw) {
                const size_t cap_scores = 2 * b->stride;  // six frames of len/3 codons each
                // one page-locked buffer: the scores, then (device branch-length scores) one value per column
                b->score_mem = score_pool.get((cap_scores + (dev_bls ? b->stride : 0)) * sizeof(double));
                double* out = reinterpret_cast<double*>(b->score_mem.first);
                double* bls_out = dev_bls ? out + cap_scores : nullptr;
                b->n_scores = backends[d]->score_alignments_6frames(model.n_leaves, b->nucl.first, b->stride, b->col0, b->len, b->start_pos, b->chrom_len,
                                                                    out, cap_scores, b->offsets, b->n_codons, bls_out);
                b->scores = out;
                if (dev_bls)
                    for (Item* p : b->items) p->bls = bls_out + p->col0;
                pool.put(b->nucl);
                b->nucl = {nullptr, 0};
            }
            return bw.lap();
        }).share();
        dev_last[d] = task;
        ++ws->pw->submitted;
        inflight.push_back({&bt, ws, task});
        while (inflight.size() > n_dev) retire_front();
    };
    std::future<std::unique_ptr<PreparedWindow>> next;
    if (range_bytes > 0) next = std::async(std::launch::async, prepare_window, size_t(0));
    if (need_raw)
        for (size_t d = 0; d < n_dev; ++d)
            dev_prepared[d] = dev_last[d] = std::async(std::launch::async, [&, d]() -> double {
                                  backends[d]->prepare_tracks(std::min<size_t>(cap, std::max<size_t>(window / 64, 4096)));
                                  return 0.0;
                              }).share();
    try {
    for (size_t wi = 0; wi + 1 < window_begin.size(); ++wi) {
        std::unique_ptr<PreparedWindow> pw = next.get();
        if (wi + 2 < window_begin.size()) next = std::async(std::launch::async, prepare_window, wi + 1);
        {  // busy seconds of stage 1 (it runs concurrently with the other stages): the parallel parse split by thread time
            const double cpu = pw->read + pw->bls, tot = (cpu > 0 ? cpu : 1.0);
            st.read_seconds += pw->parse_wall * pw->read / tot;
            st.bls_seconds += pw->parse_wall * pw->bls / tot;
            st.extract_seconds += pw->pack;
        }
        for (auto& job : pw->jobs)
            for (const Item& it : job) {
                ++st.alignments;
                st.ref_bases += it.len;
            }
        if (pw->batches.empty()) continue;
        alive.emplace_back();
        WindowSlot* ws = &alive.back();
        ws->pw = std::move(pw);
        ws->unwritten = ws->pw->batches.size();
        for (Batch& bt : ws->pw->batches) submit(bt, ws);
    }
    while (!inflight.empty()) retire_front();
    finish_writer();
    if (ranks) {
        // ONE gather per input file: every other rank's text, in rank order = file order, behind what rank 0 wrote itself (the
        // reference appends its jobs' files in job order, src/phylocsf++build_tracks.hpp:27-53)
        StopWatch gw;
        ranks->gather(prm.gather_root_text, [&](int file, const char* p, size_t n) {
            FILE* f = file == 0 ? f_power : f_raw[file - 1];
            if (f && n && fwrite(p, 1, n, f) != n) throw std::runtime_error("write failed");
        });
        st.gather_seconds += gw.lap();
    }
    } catch (...) {  // the background stages reference this frame: let them finish before it unwinds
        for (auto& f : inflight)
            if (f.scored.valid()) f.scored.wait();
        for (auto& f : dev_last)
            if (f.valid()) f.wait();
        if (writer.valid()) writer.wait();
        if (next.valid()) next.wait();
        throw;
    }
    alive.clear();
    power_file.commit();
    for (OutFile& f : raw_file) f.commit();
    if (owns_files && (prm.write_smooth || prm.write_regions)) {
        sw.lap();
        bool smoothed = false;
        std::array<bool, 6> done{};   // frames whose smoothed files the device pass committed before it failed
        if (backend.device_hmm()) {
            // every score is written by now: a failure of the device pass (its memory, say) must not lose the run -- the host pass reads the
            // same raw tracks and writes the same bytes
            try {
                smooth_tracks_device(out_dir, prm.hmm, prm.write_smooth, prm.write_regions, threads, backend, &done);
                smoothed = true;
                if (!prm.write_raw) remove_raw_tracks(out_dir);
            } catch (const std::exception& ex) {
                fprintf(stderr, "the smoothing pass on the device failed (%s); running the remaining tracks on the host\n", ex.what());
            }
            backend.hmm_release();   // the pass's device buffers (~60 bytes per value of its largest batch)
        }
        if (!smoothed) smooth_tracks(out_dir, prm.hmm, prm.write_smooth, prm.write_regions, /*keep_raw=*/prm.write_raw, threads, &done);
        st.smooth_seconds += sw.lap();
    }
    st.total_seconds = total.lap();
    return st;
}

inline BuildTracksStats build_tracks(const Model& model, const std::string& maf_path, const BuildTracksParams& prm, ScoringBackend& backend,
                                     bool append = false)
{
    return build_tracks(model, maf_path, prm, std::vector<ScoringBackend*>{&backend}, append);
}

// pack-maf: the build-tracks view of a MAF (what MafReader yields with concatenate = true) as a packed alignment file.
// Byte-range jobs parse in parallel, window by window; records are written in file order.
struct PackStats {
    size_t alignments = 0, ref_bases = 0;
    uint64_t bytes_in = 0, bytes_out = 0;
    double seconds = 0.0;
};
inline PackStats pack_maf(const Model& model, const std::string& maf_path, const std::string& out_path, int threads, size_t window_bytes = size_t(256) << 20)
{
    StopWatch total;
    MafFile file(maf_path);
    if (is_packed_file(file)) throw std::runtime_error("'" + maf_path + "' is already a packed alignment file");
    PackedWriter out(out_path, model.n_leaves, row_map_hash(model));
    PackStats st;
    st.bytes_in = file.size();
    if (threads <= 0) threads = effective_threads();
    for (size_t w0 = 0; w0 < file.size(); w0 += window_bytes) {
        const size_t w1 = std::min(file.size(), w0 + window_bytes);
        const size_t n_jobs = std::max<size_t>(1, std::min<size_t>(static_cast<size_t>(threads) * 4, (w1 - w0 + (1 << 16) - 1) >> 16));
        std::vector<std::vector<Alignment>> jobs(n_jobs);
        parallel_for(n_jobs, threads, [&](size_t j) {
            const size_t b = w0 + (w1 - w0) * j / n_jobs, e = w0 + (w1 - w0) * (j + 1) / n_jobs;
            MafReader reader(file, b, e, model.name_to_leaf, model.n_leaves, /*concatenate=*/true, false);
            Alignment aln;
            while (reader.next(aln)) {
                jobs[j].emplace_back();
                std::swap(jobs[j].back(), aln);
            }
        });
        for (const auto& job : jobs)
            for (const Alignment& a : job) {
                out.add(a);
                ++st.alignments;
                st.ref_bases += a.length();
            }
    }
    out.finish();
    st.bytes_out = out.bytes();
    st.seconds = total.lap();
    return st;
}

struct ScoreMsaParams {
    std::string output_dir;  // empty: <maf>.scores next to the alignment
    int strategy = ALGO_MLE;
    bool comp_phylo = true, comp_anc = false, comp_bls = true;
    bool verbose = false;
    Hmm hmm;  // FIXED_MEAN only
    std::string version_line = "# PhyloCSF scores computed with phylocsf-amd (MI355X)";
};

struct ScoreMsaStats {
    size_t blocks = 0, codon_columns = 0;
    double read_seconds = 0.0, scoring_seconds = 0.0, write_seconds = 0.0, total_seconds = 0.0;
};

inline std::string score_msa(const Model& model, const std::string& maf_path, const ScoreMsaParams& prm, const std::vector<ScoringBackend*>& backends,
                             ScoreMsaStats* stats = nullptr)
{
    if (backends.empty()) throw std::runtime_error("score_msa: no scoring backend");
    const auto t_enter = std::chrono::steady_clock::now();
    auto since = [](std::chrono::steady_clock::time_point t) { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t).count(); };
    ScoringBackend& backend = *backends[0];
    if (prm.strategy != ALGO_FIXED_MEAN)
        for (ScoringBackend* b : backends) b->prepare_blocks(prm.strategy);   // while the file is read
    std::string out_path = prm.output_dir.empty() ? maf_path + ".scores" : prm.output_dir + "/" + maf_path.substr(maf_path.find_last_of('/') == std::string::npos ? 0 : maf_path.find_last_of('/') + 1) + ".scores";
    FILE* out = open_out(out_path, "w");
    fprintf(out, "%s\n", prm.version_line.c_str());
    fprintf(out, "seq\tstart\tend\tstrand");
    if (prm.comp_phylo) fprintf(out, "\tphylocsf-score");
    if (prm.comp_anc) fprintf(out, "\tanc-score");
    if (prm.comp_bls) fprintf(out, "\tbls-score");
    fprintf(out, "\n");

    struct Row { std::string chrom; uint64_t start, end; char strand; float bls; };
    std::vector<Row> rows;
    const size_t nl = static_cast<size_t>(model.n_leaves);
    // Byte-range jobs on all host threads (the reader's split semantics: a block belongs to the job its "a" line starts in), each
    // translating its blocks and computing their branch-length scores; the jobs' rows are then laid end to end in file order.  (One
    // thread over the 90 MB of BASELINE.json configs[2]'s MAF took 0.23 s -- as long as the MLE scoring of its 5,088 blocks.)
    struct JobRows {
        std::vector<Row> rows;
        std::vector<uint64_t> n_codons;
        std::vector<std::vector<uint8_t>> per_leaf;
    };
    MafFile file(maf_path);
    const int threads = effective_threads();
    const size_t n_jobs = std::max<size_t>(1, std::min<size_t>(static_cast<size_t>(threads) * 4, (file.size() + (1 << 16) - 1) >> 16));
    std::vector<JobRows> jobs(n_jobs);
    parallel_for(n_jobs, threads, [&](size_t j) {
        const size_t b = file.size() * j / n_jobs, e = file.size() * (j + 1) / n_jobs;
        MafReader reader(file, b, e, model.name_to_leaf, model.n_leaves, /*concatenate=*/false, prm.verbose && j == 0);
        BranchLengthScore bls(model);
        JobRows& jr = jobs[j];
        jr.per_leaf.resize(nl);
        Alignment aln;
        while (reader.next(aln)) {
            const size_t len = aln.length();
            Row r{aln.chrom, aln.start_pos, aln.start_pos + len - 1, aln.strand, NAN};
            if (prm.comp_bls) r.bls = static_cast<float>(bls.per_base(aln, nullptr));
            const size_t nc = len / 3;  // translate(): frame +1 from column 0
            for (size_t l = 0; l < nl; ++l) {
                const size_t at = jr.per_leaf[l].size();
                jr.per_leaf[l].resize(at + nc);
                const char* sq = aln.seqs[l].data();
                for (size_t c = 0; c < nc; ++c) jr.per_leaf[l][at + c] = codon_id(sq[3 * c], sq[3 * c + 1], sq[3 * c + 2]);
            }
            jr.n_codons.push_back(nc);
            jr.rows.push_back(std::move(r));
        }
    });
    std::vector<uint64_t> offsets{0};
    std::vector<size_t> job_site0(n_jobs + 1, 0);
    for (size_t j = 0; j < n_jobs; ++j) {
        for (uint64_t nc : jobs[j].n_codons) offsets.push_back(offsets.back() + nc);
        job_site0[j + 1] = offsets.back();
        for (Row& r : jobs[j].rows) rows.push_back(std::move(r));
    }
    const size_t total = offsets.back(), stride = std::max<size_t>(total, 1);
    std::vector<uint8_t> codons(nl * stride);
    parallel_for(n_jobs, threads, [&](size_t j) {
        for (size_t l = 0; l < nl; ++l) std::copy(jobs[j].per_leaf[l].begin(), jobs[j].per_leaf[l].end(), codons.begin() + l * stride + job_site0[j]);
    });
    std::vector<float> phylo(rows.size(), NAN), anc(rows.size(), NAN);
    const double t_read = since(t_enter);
    const auto t_score = std::chrono::steady_clock::now();
    if (prm.strategy == ALGO_FIXED_MEAN) {
        // src/phylocsf++score_msa.hpp:133-212: per-codon FIXED scores of the block (run_tracks), smoothed by the HMM as
        // ONE run, and the mean log-odds -- accumulated in float like the reference -- reported as the score; no anc score
        std::vector<double> per_codon(total), post, work;
        if (total) backend.score_sites_fixed(codons.data(), total, stride, per_codon.data());
        for (size_t i = 0; i < rows.size(); ++i) {
            const std::vector<double> scores(per_codon.begin() + offsets[i], per_codon.begin() + offsets[i + 1]);
            hmm_coding_posterior(prm.hmm, scores, post, work);
            float sum = 0.0f;
            for (double q : post) sum += hmm_log_odds(q);
            phylo[i] = sum / static_cast<uint64_t>(post.size());
        }
    } else if ((prm.comp_phylo || prm.comp_anc) && !rows.empty()) {
        // blocks are independent: contiguous ranges with about equal site counts (+ a constant per block for its branch
        // matrices) go to the backends, one thread each; CSR offsets are absolute, so every backend sees the same matrix
        const size_t n_dev = std::min(backends.size(), rows.size());
        if (prm.strategy != ALGO_FIXED) {   // one pruning kernel for the whole FILE, whatever the cut below
            const int kernel = backend.choose_block_kernel(codons.data(), offsets.data(), rows.size(), stride);
            for (size_t d = 0; d < n_dev; ++d) backends[d]->set_block_kernel(kernel);
        }
        std::vector<size_t> cut(n_dev + 1, rows.size());
        cut[0] = 0;
        const double total_cost = static_cast<double>(total) + 16.0 * static_cast<double>(rows.size());
        size_t b = 0;
        double acc_cost = 0.0;
        for (size_t d = 1; d < n_dev; ++d) {
            while (b < rows.size() && acc_cost < total_cost * static_cast<double>(d) / static_cast<double>(n_dev)) {
                acc_cost += static_cast<double>(offsets[b + 1] - offsets[b]) + 16.0;
                ++b;
            }
            cut[d] = b;
        }
        parallel_for(n_dev, static_cast<int>(n_dev), [&](size_t d) {
            const size_t b0 = cut[d], b1 = cut[d + 1];
            if (b1 > b0)
                backends[d]->score_blocks(prm.strategy, codons.data(), offsets.data() + b0, b1 - b0, stride, prm.comp_anc, 42u, phylo.data() + b0, anc.data() + b0);
        });
    }
    const double t_scoring = since(t_score);
    const auto t_write = std::chrono::steady_clock::now();
    for (size_t i = 0; i < rows.size(); ++i) {
        const Row& r = rows[i];
        fprintf(out, "%s\t%" PRIu64 "\t%" PRIu64 "\t%c", r.chrom.c_str(), r.start, r.end, r.strand);
        if (prm.comp_phylo) fprintf(out, "\t%.6f", static_cast<double>(phylo[i]));
        if (prm.comp_anc) fprintf(out, "\t%.6f", static_cast<double>(anc[i]));
        if (prm.comp_bls) fprintf(out, "\t%.6f", static_cast<double>(r.bls));
        fprintf(out, "\n");
    }
    fclose(out);
    if (stats) {
        stats->blocks = rows.size();
        stats->codon_columns = total;
        stats->read_seconds = t_read;
        stats->scoring_seconds = t_scoring;
        stats->write_seconds = since(t_write);
        stats->total_seconds = since(t_enter);
    }
    return out_path;
}

inline std::string score_msa(const Model& model, const std::string& maf_path, const ScoreMsaParams& prm, ScoringBackend& backend)
{
    return score_msa(model, maf_path, prm, std::vector<ScoringBackend*>{&backend});
}

}  // namespace pcsf_host
