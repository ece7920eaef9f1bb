// hmm_abi.cpp -- pcsf_hmm_smooth_runs: the smoothing pass of build-tracks (--output-phylo / --output-regions) on the device.
// Replaces, for runs of raw per-codon scores, the reference's hmm::state_posterior_probabilities and the forward half of
// get_best_path_by_viterbi (src/create_tracks.hpp:25-159); the kernels and why their chunked evaluation yields the sequential
// algorithm's own bits are in hmm.hip.  This file cuts the runs into chunks, runs the three recurrences and repeats the chunks whose
// entry state did not meet their neighbour's exit state until every chunk is exact.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ctx.hpp"

struct BinRequest {   // pcsf_hmm_smooth_runs_binned: bins of the posterior instead of the posterior
    const double* thresholds = nullptr;
    size_t n_thresholds = 0;
    uint16_t* out_bins = nullptr;
    uint64_t* exc_index = nullptr;
    double* exc_value = nullptr;
    size_t exc_capacity = 0;
    size_t* n_exc = nullptr;
};

static int smooth_runs(pcsf_ctx* c, const double* init, const double* trans, const double* emissions, const uint64_t* run_offsets, size_t n_runs,
                       double* out_posterior, const BinRequest* bins, uint8_t* out_back, double* out_last, pcsf_hmm_stats* stats)
{
    if (!c) return PCSF_ERR_INVALID_ARG;
    if (!init || !trans || !run_offsets || (!out_posterior && !bins)) return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_hmm_smooth_runs: null argument");
    if (bins && (!bins->thresholds || !bins->out_bins || !bins->n_exc || bins->n_thresholds >= 0xfffeu || (bins->exc_capacity && (!bins->exc_index || !bins->exc_value))))
        return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_hmm_smooth_runs_binned: null argument or 65,534 thresholds or more");
    if (bins) *bins->n_exc = 0;
    if ((out_back == nullptr) != (out_last == nullptr)) return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_hmm_smooth_runs: out_back and out_last go together");
    if (stats) *stats = pcsf_hmm_stats{};
    const uint64_t n = n_runs ? run_offsets[n_runs] : 0;
    if (n == 0) return PCSF_OK;
    if (!emissions) return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_hmm_smooth_runs: null argument");
    for (size_t r = 0; r < n_runs; ++r)
        if (run_offsets[r + 1] <= run_offsets[r]) return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_hmm_smooth_runs: empty run or offsets not increasing");
    // Chunk length and burn-in.  How fast the filter forgets is set by the model more than by the scores: the three non-coding classes
    // emit the same, a class is only left towards coding, and the reference's parameters give them mean lengths of 105, 570 and
    // 21,970 codons -- what tells two trajectories apart survives until a stretch of high scores pushes both through the coding state.
    // Measured (tools/hmm_long_run.py, one run of 5 M random scores): with 32,768 / 32,768 half of the 152 later chunks arrive at
    // their neighbour's state and half are evaluated again (1.7 s against 5.3 s for the run as one chunk); with 4,096 / 4,096 nearly
    // all are.  Runs shorter than a chunk -- the rule: the reader's break points end a run every 333 k codons at the latest, gaps in
    // the alignment much sooner -- are one thread each and exact by construction; the command line keeps runs of a million values and
    // more on host threads (drivers.hpp).  PCSF_HMM_CHUNK / PCSF_HMM_BURN_IN for experiments and tests.
    uint64_t chunk_len = 32768, burn_in = 32768;
    if (const char* v = std::getenv("PCSF_HMM_CHUNK")) chunk_len = std::max<uint64_t>(1, std::strtoull(v, nullptr, 10));
    if (const char* v = std::getenv("PCSF_HMM_BURN_IN")) burn_in = std::max<uint64_t>(1, std::strtoull(v, nullptr, 10));
    std::vector<pcsf::HmmChunk> chunks;
    chunks.reserve(n / chunk_len + n_runs);
    std::vector<uint32_t> last_chunk(n_runs);
    for (size_t r = 0; r < n_runs; ++r) {
        const uint64_t b = run_offsets[r], e = run_offsets[r + 1];
        for (uint64_t at = b; at < e; at += chunk_len) chunks.push_back(pcsf::HmmChunk{b, e, at, std::min(e, at + chunk_len)});
        last_chunk[r] = static_cast<uint32_t>(chunks.size() - 1);
    }
    if (chunks.size() > 0x7fffffffull) return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_hmm_smooth_runs: too many chunks in one call");
    const uint32_t n_chunks = static_cast<uint32_t>(chunks.size());

    hipError_t e;
    if ((e = hipSetDevice(c->device)) != hipSuccess) return c->hip_fail(e, "hipSetDevice");
    hipStream_t st = c->stream;
    pcsf_ctx::HmmBuffers& B = c->hmm;
    // Every return below -- also the error returns in the middle of the passes -- first waits for what is in flight on the stream: the
    // asynchronous copies read this function's vectors (chunks, same, todo) and the caller's arrays, all of which may be gone a moment later.
    struct Drain {
        hipStream_t s;
        ~Drain() { (void)hipStreamSynchronize(s); }
    } drain{st};
#define HMM_TRY(x, what) \
    if ((e = (x)) != hipSuccess) return c->hip_fail(e, what)
    HMM_TRY(B.emit.reserve(n * sizeof(double)), "hipMalloc(hmm emissions)");
    HMM_TRY(B.fwd.reserve(n * 4 * sizeof(double)), "hipMalloc(hmm forward)");
    HMM_TRY(B.post.reserve(n * sizeof(double)), "hipMalloc(hmm posterior)");
    if (out_back) HMM_TRY(B.back.reserve(n), "hipMalloc(hmm back pointers)");
    HMM_TRY(B.chunks.reserve(static_cast<size_t>(n_chunks) * sizeof(pcsf::HmmChunk)), "hipMalloc(hmm chunks)");
    HMM_TRY(B.entry.reserve(static_cast<size_t>(n_chunks) * 4 * sizeof(double)), "hipMalloc(hmm states)");
    HMM_TRY(B.exit_state.reserve(static_cast<size_t>(n_chunks) * 4 * sizeof(double)), "hipMalloc(hmm states)");
    HMM_TRY(B.same.reserve(n_chunks), "hipMalloc(hmm flags)");
    HMM_TRY(B.todo.reserve(static_cast<size_t>(n_chunks) * sizeof(uint32_t)), "hipMalloc(hmm list)");
    HMM_TRY(hipMemcpyAsync(B.emit.ptr, emissions, n * sizeof(double), hipMemcpyHostToDevice, st), "hipMemcpyAsync(hmm emissions)");
    HMM_TRY(hipMemcpyAsync(B.chunks.ptr, chunks.data(), static_cast<size_t>(n_chunks) * sizeof(pcsf::HmmChunk), hipMemcpyHostToDevice, st), "hipMemcpyAsync(hmm chunks)");
    if (out_back) HMM_TRY(hipMemsetAsync(B.back.ptr, 0, n, st), "hipMemsetAsync(hmm back pointers)");

    pcsf::HmmModel h;
    std::memcpy(h.init, init, sizeof h.init);
    std::memcpy(h.trans, trans, sizeof h.trans);
    std::vector<uint8_t> same(n_chunks), exact(n_chunks);
    std::vector<uint32_t> todo;
    // one recurrence: all chunks with burn-in, then passes over the first inexact chunk of every stretch until none is left
    auto recurrence = [&](int mode, uint32_t* reruns, uint32_t* passes) -> int {
        const bool backward = mode == 2;
        HMM_TRY(pcsf::launch_hmm_chunks(mode, h, B.chunks.as<pcsf::HmmChunk>(), nullptr, n_chunks, false, static_cast<uint32_t>(std::min<uint64_t>(burn_in, 0xffffffffu)),
                                        B.emit.as<double>(), B.fwd.as<double>(), B.back.as<uint8_t>(), B.post.as<double>(), B.entry.as<double>(),
                                        B.exit_state.as<double>(), st),
                "hmm_chunks_kernel");
        for (;;) {
            HMM_TRY(pcsf::launch_hmm_compare(B.chunks.as<pcsf::HmmChunk>(), n_chunks, backward, B.entry.as<double>(), B.exit_state.as<double>(), B.same.as<uint8_t>(), st),
                    "hmm_compare_kernel");
            HMM_TRY(hipMemcpyAsync(same.data(), B.same.ptr, n_chunks, hipMemcpyDeviceToHost, st), "hipMemcpyAsync(hmm flags)");
            HMM_TRY(hipStreamSynchronize(st), "hipStreamSynchronize(hmm)");
            todo.clear();
            if (!backward) {
                for (uint32_t k = 0; k < n_chunks; ++k) {
                    const bool first = chunks[k].begin == chunks[k].run_begin;
                    exact[k] = first || (same[k] && exact[k - 1]);
                    if (!exact[k] && exact[k - 1]) todo.push_back(k);
                }
            } else {
                for (uint32_t k = n_chunks; k-- > 0;) {
                    const bool lastc = chunks[k].end == chunks[k].run_end;
                    exact[k] = lastc || (same[k] && exact[k + 1]);
                    if (!exact[k] && exact[k + 1]) todo.push_back(k);
                }
            }
            if (todo.empty()) return PCSF_OK;
            *reruns += static_cast<uint32_t>(todo.size());
            ++*passes;
            HMM_TRY(hipMemcpyAsync(B.todo.ptr, todo.data(), todo.size() * sizeof(uint32_t), hipMemcpyHostToDevice, st), "hipMemcpyAsync(hmm list)");
            HMM_TRY(pcsf::launch_hmm_chunks(mode, h, B.chunks.as<pcsf::HmmChunk>(), B.todo.as<uint32_t>(), static_cast<uint32_t>(todo.size()), true, 0, B.emit.as<double>(),
                                            B.fwd.as<double>(), B.back.as<uint8_t>(), B.post.as<double>(), B.entry.as<double>(), B.exit_state.as<double>(), st),
                    "hmm_chunks_kernel (rerun)");
        }
    };
    pcsf_hmm_stats local{};
    local.chunks = n_chunks;
    int rc;
    if (out_back) {
        if ((rc = recurrence(1, &local.viterbi_reruns, &local.viterbi_passes)) != PCSF_OK) return rc;
        std::vector<double> exits(static_cast<size_t>(n_chunks) * 4);
        HMM_TRY(hipMemcpyAsync(exits.data(), B.exit_state.ptr, exits.size() * sizeof(double), hipMemcpyDeviceToHost, st), "hipMemcpyAsync(hmm states)");
        HMM_TRY(hipMemcpyAsync(out_back, B.back.ptr, n, hipMemcpyDeviceToHost, st), "hipMemcpyAsync(hmm back pointers)");
        HMM_TRY(hipStreamSynchronize(st), "hipStreamSynchronize(hmm)");
        for (size_t r = 0; r < n_runs; ++r) std::memcpy(out_last + 4 * r, exits.data() + 4 * static_cast<size_t>(last_chunk[r]), 4 * sizeof(double));
    }
    if ((rc = recurrence(0, &local.forward_reruns, &local.forward_passes)) != PCSF_OK) return rc;
    if ((rc = recurrence(2, &local.backward_reruns, &local.backward_passes)) != PCSF_OK) return rc;
    if (out_posterior) HMM_TRY(hipMemcpyAsync(out_posterior, B.post.ptr, n * sizeof(double), hipMemcpyDeviceToHost, st), "hipMemcpyAsync(hmm posterior)");
    unsigned int n_exc = 0;
    if (bins) {
        pcsf::DeviceBuffer &d_thr = B.thresholds, &d_bins = B.bins, &d_exc_i = B.exc_index, &d_exc_v = B.exc_value, &d_count = B.exc_count;
        HMM_TRY(d_thr.reserve(std::max<size_t>(bins->n_thresholds, 1) * sizeof(double)), "hipMalloc(hmm thresholds)");
        HMM_TRY(d_bins.reserve(n * sizeof(uint16_t)), "hipMalloc(hmm bins)");
        HMM_TRY(d_exc_i.reserve(std::max<size_t>(bins->exc_capacity, 1) * sizeof(uint64_t)), "hipMalloc(hmm exceptions)");
        HMM_TRY(d_exc_v.reserve(std::max<size_t>(bins->exc_capacity, 1) * sizeof(double)), "hipMalloc(hmm exceptions)");
        HMM_TRY(d_count.reserve(sizeof(unsigned int)), "hipMalloc(hmm exceptions)");
        HMM_TRY(hipMemcpyAsync(d_thr.ptr, bins->thresholds, bins->n_thresholds * sizeof(double), hipMemcpyHostToDevice, st), "hipMemcpyAsync(hmm thresholds)");
        HMM_TRY(hipMemsetAsync(d_count.ptr, 0, sizeof(unsigned int), st), "hipMemsetAsync(hmm exceptions)");
        HMM_TRY(pcsf::launch_hmm_bins(B.post.as<double>(), n, d_thr.as<double>(), static_cast<uint32_t>(bins->n_thresholds), d_bins.as<uint16_t>(), d_exc_i.as<uint64_t>(),
                                      d_exc_v.as<double>(), static_cast<uint32_t>(std::min<size_t>(bins->exc_capacity, 0xffffffffu)), d_count.as<unsigned int>(), st),
                "hmm_bin_kernel");
        HMM_TRY(hipMemcpyAsync(bins->out_bins, d_bins.ptr, n * sizeof(uint16_t), hipMemcpyDeviceToHost, st), "hipMemcpyAsync(hmm bins)");
        HMM_TRY(hipMemcpyAsync(&n_exc, d_count.ptr, sizeof n_exc, hipMemcpyDeviceToHost, st), "hipMemcpyAsync(hmm exceptions)");
        HMM_TRY(hipStreamSynchronize(st), "hipStreamSynchronize(hmm)");
        const size_t got = std::min<size_t>(n_exc, bins->exc_capacity);
        if (got) {
            HMM_TRY(hipMemcpy(bins->exc_index, d_exc_i.ptr, got * sizeof(uint64_t), hipMemcpyDeviceToHost), "hipMemcpy(hmm exceptions)");
            HMM_TRY(hipMemcpy(bins->exc_value, d_exc_v.ptr, got * sizeof(double), hipMemcpyDeviceToHost), "hipMemcpy(hmm exceptions)");
        }
        *bins->n_exc = n_exc;
    }
    HMM_TRY(hipStreamSynchronize(st), "hipStreamSynchronize(hmm)");
#undef HMM_TRY
    if (stats) *stats = local;
    return PCSF_OK;
}

extern "C" int pcsf_hmm_smooth_runs(pcsf_ctx* c, const double* init, const double* trans, const double* emissions, const uint64_t* run_offsets, size_t n_runs,
                                    double* out_posterior, uint8_t* out_back, double* out_last, pcsf_hmm_stats* stats)
{
    if (c && !out_posterior) return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_hmm_smooth_runs: null argument");
    return smooth_runs(c, init, trans, emissions, run_offsets, n_runs, out_posterior, nullptr, out_back, out_last, stats);
}

extern "C" int pcsf_hmm_smooth_runs_binned(pcsf_ctx* c, const double* init, const double* trans, const double* emissions, const uint64_t* run_offsets, size_t n_runs,
                                           const double* thresholds, size_t n_thresholds, uint16_t* out_bins, uint64_t* exc_index, double* exc_value,
                                           size_t exc_capacity, size_t* n_exceptions, uint8_t* out_back, double* out_last, pcsf_hmm_stats* stats)
{
    BinRequest b;
    b.thresholds = thresholds;
    b.n_thresholds = n_thresholds;
    b.out_bins = out_bins;
    b.exc_index = exc_index;
    b.exc_value = exc_value;
    b.exc_capacity = exc_capacity;
    b.n_exc = n_exceptions;
    return smooth_runs(c, init, trans, emissions, run_offsets, n_runs, nullptr, &b, out_back, out_last, stats);
}

extern "C" int pcsf_hmm_release(pcsf_ctx* c)
{
    if (!c) return PCSF_ERR_INVALID_ARG;
    if (hipSetDevice(c->device) != hipSuccess) return c->fail(PCSF_ERR_HIP, "hipSetDevice");
    (void)hipStreamSynchronize(c->stream);
    pcsf_ctx::HmmBuffers& B = c->hmm;
    for (pcsf::DeviceBuffer* b : {&B.emit, &B.fwd, &B.back, &B.post, &B.chunks, &B.entry, &B.exit_state, &B.same, &B.todo, &B.thresholds, &B.bins, &B.exc_index, &B.exc_value, &B.exc_count})
        b->release();
    return PCSF_OK;
}
