// hmm.hip -- the smoothing pass of build-tracks on the device: coding posterior (scaled forward-backward) and Viterbi back pointers
// of the reference's 4-state HMM over runs of raw per-codon scores
//   hmm::state_posterior_probabilities / get_best_path_by_viterbi   src/create_tracks.hpp:25-159
// with the reference's arithmetic (sums over states in index order, division by the per-position maximum; the file is compiled
// with -ffp-contract=off and IEEE division), so every value equals the host restatement's (host/hmm.hpp) bit for bit.
//
// The recurrences are sequential along a run, and a run can be a whole chromosome (tens of millions of values in a contiguous
// alignment).  They are cut into CHUNKS, one thread per chunk, on this observation: the scaled state vector after a step is a
// deterministic function of the vector before it and of that position's emission, and the filter forgets where it started -- two
// trajectories that start apart at position p - B arrive, as a rule, at bit-identical vectors well before p, and identical they
// stay.  So a chunk that does not begin its run first BURNS IN over the B positions before it from a made-up state, keeps what it
// arrives at as its ENTRY state, and runs on from there; afterwards every entry state is compared, bit for bit, with the EXIT state
// of the chunk before it.  A chunk whose predecessor is exact and whose entry state equals that predecessor's exit state is exact
// itself (it computed exactly what the sequential scan computes).  A chunk that fails the comparison runs again from its
// predecessor's exit state (hmm_abi.cpp: passes until every chunk is exact; the worst case is the sequential order).  The outputs
// are therefore the sequential algorithm's own bits, not an approximation of them.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace pcsf {

namespace {

struct Vec {
    double s[4];
};

__device__ __forceinline__ double max_of(double a, double b) { return a < b ? b : a; }   // std::max(a, b)

// one forward step: fwd[pos] from fwd[pos - 1] and the coding emission at pos (host/hmm.hpp hmm_coding_posterior, first loop)
__device__ __forceinline__ Vec forward_step(const HmmModel& h, const Vec& prev, double e0)
{
    Vec v;
    double maxf = 0.0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        double sum = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) sum += prev.s[q] * h.trans[q][s];
        v.s[s] = sum * (s == 0 ? e0 : 1.0);
        maxf = max_of(maxf, v.s[s]);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) v.s[s] /= maxf;
    return v;
}

// one backward step: bwd[pos] from bwd[pos + 1] and the coding emission at pos + 1 (second loop)
__device__ __forceinline__ Vec backward_step(const HmmModel& h, const Vec& next, double e0_next)
{
    Vec v;
    double maxb = 0.0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        double sum = 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) sum += (h.trans[s][q] * (q == 0 ? e0_next : 1.0) * next.s[q]);
        v.s[s] = sum;
        maxb = max_of(maxb, sum);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) v.s[s] /= maxb;
    return v;
}

// one Viterbi step (hmm_viterbi): the best predecessor of every state (ties keep the lowest), rescaled by the position's maximum
__device__ __forceinline__ Vec viterbi_step(const HmmModel& h, const Vec& prev, double e0, uint8_t* back)
{
    Vec cur;
    double max_prob = 0.0;
    unsigned packed = 0;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        double best = 0.0;
        unsigned arg = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double v = prev.s[q] * h.trans[q][s];
            if (v > best) {
                arg = q;
                best = v;
            }
        }
        best *= (s == 0 ? e0 : 1.0);
        packed |= arg << (2 * s);
        cur.s[s] = best;
        max_prob = max_of(max_prob, best);
    }
    if (back) *back = static_cast<uint8_t>(packed);
#pragma unroll
    for (int s = 0; s < 4; ++s) cur.s[s] /= max_prob;
    return cur;
}

__device__ __forceinline__ Vec start_vector(const HmmModel& h, double e0)   // fwd[0] = init o emission: not rescaled
{
    Vec v;
#pragma unroll
    for (int s = 0; s < 4; ++s) v.s[s] = h.init[s] * (s == 0 ? e0 : 1.0);
    return v;
}

__device__ __forceinline__ Vec load_vec(const double* p)
{
    const double2* q = reinterpret_cast<const double2*>(p);
    const double2 a = q[0], b = q[1];
    return Vec{{a.x, a.y, b.x, b.y}};
}
__device__ __forceinline__ void store_vec(double* p, const Vec& v)
{
    double2* q = reinterpret_cast<double2*>(p);
    q[0] = double2{v.s[0], v.s[1]};
    q[1] = double2{v.s[2], v.s[3]};
}

// A thread's emissions, eight positions at a time and one block ahead: every lane walks its own stretch of memory, so a load per step
// is a cache line per lane and a full memory round trip in the recurrence's dependency chain (measured: 6-8 k cycles per step where
// the arithmetic needs ~1 k).  Eight independent loads per block, issued while the block before is computed.  UP: positions ascend.
template <bool UP>
struct EmissionStream {
    const double* __restrict__ base;
    uint64_t next;        // first position of the block after the current one (UP) / one past the last position of it (!UP)
    uint64_t limit;       // positions are in [limit_lo .. ) for !UP resp. [ .. limit) for UP: loads beyond are clamped and never used
    double cur[8], ahead[8];
    __device__ __forceinline__ void fetch(double (&dst)[8], uint64_t at)
    {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (UP) {
                const uint64_t p = at + k;
                dst[k] = base[p < limit ? p : limit - 1];
            } else {
                const uint64_t p = at >= static_cast<uint64_t>(k) + 1 + limit ? at - 1 - k : limit;   // at - 1 - k, not below limit
                dst[k] = base[p];
            }
        }
    }
    // positions first, first + 1, ... (UP) or first, first - 1, ... (!UP), none outside [lo, hi)
    __device__ __forceinline__ void start(const double* b, uint64_t first, uint64_t lo, uint64_t hi)
    {
        base = b;
        if (UP) {
            limit = hi;
            fetch(cur, first);
            next = first + 8;
            fetch(ahead, next);
        } else {
            limit = lo;
            fetch(cur, first + 1);
            next = first + 1 >= 8 + lo ? first + 1 - 8 : lo;
            fetch(ahead, next);
        }
    }
    __device__ __forceinline__ void advance()
    {
#pragma unroll
        for (int k = 0; k < 8; ++k) cur[k] = ahead[k];
        if (UP) {
            next += 8;
            fetch(ahead, next);
        } else {
            next = next >= 8 + limit ? next - 8 : limit;
            fetch(ahead, next);
        }
    }
};

}  // namespace

// MODE 0: forward (fwd[n][4] out), 1: Viterbi (back[n] out: four 2-bit predecessors per position; nothing at a run's first position),
// 2: backward + posterior (reads fwd, post[n] out).  `todo` lists the chunks this pass runs (null: all); rerun: the chunks start from
// their predecessor's exit state instead of burning in.
template <int MODE>
__global__ __launch_bounds__(64) void hmm_chunks_kernel(const HmmModel h, const HmmChunk* __restrict__ chunks, const uint32_t* __restrict__ todo, uint32_t n_todo,
                                                        bool rerun, uint32_t burn_in, const double* __restrict__ emit, double* __restrict__ fwd,
                                                        uint8_t* __restrict__ back, double* __restrict__ post, double* __restrict__ entry,
                                                        double* __restrict__ exit_state)
{
    const uint32_t t = blockIdx.x * 64u + threadIdx.x;
    if (t >= n_todo) return;
    const uint32_t c = todo ? todo[t] : t;
    const HmmChunk ch = chunks[c];
    if (MODE != 2) {
        Vec v;
        uint64_t pos = ch.begin;
        if (ch.begin == ch.run_begin) {   // the run starts here: exact by construction
            v = start_vector(h, emit[pos]);
            if (MODE == 0) store_vec(fwd + pos * 4, v);
            store_vec(entry + static_cast<size_t>(c) * 4, v);
            ++pos;
        } else {
            if (rerun) {
                v = load_vec(exit_state + static_cast<size_t>(c - 1) * 4);   // the chunk before this one in the run: exact by now
            } else {
                uint64_t p = ch.begin - ch.run_begin > burn_in ? ch.begin - burn_in : ch.run_begin;
                v = start_vector(h, emit[p]);   // a made-up state unless p is the run's first position
                ++p;
                if (p < ch.begin) {
                    EmissionStream<true> es;
                    es.start(emit, p, ch.run_begin, ch.begin);
                    while (p < ch.begin) {
#pragma unroll
                        for (int k = 0; k < 8; ++k)
                            if (p + k < ch.begin) v = MODE == 0 ? forward_step(h, v, es.cur[k]) : viterbi_step(h, v, es.cur[k], nullptr);
                        p += 8;
                        es.advance();
                    }
                }
            }
            store_vec(entry + static_cast<size_t>(c) * 4, v);   // the state at begin - 1
        }
        if (pos < ch.end) {
            EmissionStream<true> es;
            es.start(emit, pos, ch.run_begin, ch.end);
            while (pos < ch.end) {
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (pos + k < ch.end) {
                        if (MODE == 0) {
                            v = forward_step(h, v, es.cur[k]);
                            store_vec(fwd + (pos + k) * 4, v);
                        } else {
                            v = viterbi_step(h, v, es.cur[k], back + pos + k);
                        }
                    }
                pos += 8;
                es.advance();
            }
        }
        store_vec(exit_state + static_cast<size_t>(c) * 4, v);   // the state at end - 1
    } else {
        // backward: the chunk order is the same, the dependency runs the other way -- entry = the state at `end` (the position after the
        // chunk), exit = the state at `begin`; the chunk AFTER this one in the run is c + 1
        Vec v;
        uint64_t pos = ch.end;   // positions pos - 1 down to begin are this chunk's
        if (ch.end == ch.run_end) {
            v = Vec{{1.0, 1.0, 1.0, 1.0}};   // bwd[n - 1]
            store_vec(entry + static_cast<size_t>(c) * 4, v);
            --pos;
            const Vec f = load_vec(fwd + pos * 4);
            double total = 0.0;
#pragma unroll
            for (int s = 0; s < 4; ++s) total += (f.s[s] * v.s[s]);
            post[pos] = (f.s[0] * v.s[0]) / total;
        } else {
            if (rerun) {
                v = load_vec(exit_state + static_cast<size_t>(c + 1) * 4);
            } else {
                uint64_t p = ch.run_end - ch.end > burn_in ? ch.end + burn_in : ch.run_end;   // one past the made-up start
                v = Vec{{1.0, 1.0, 1.0, 1.0}};
                --p;
                if (p > ch.end) {   // state at p - 1 from the state at p, for p down to end + 1
                    EmissionStream<false> es;
                    es.start(emit, p, ch.end + 1, ch.run_end);
                    while (p > ch.end) {
#pragma unroll
                        for (int k = 0; k < 8; ++k)
                            if (p >= ch.end + 1 + static_cast<uint64_t>(k)) v = backward_step(h, v, es.cur[k]);
                        p = p >= ch.end + 8 ? p - 8 : ch.end;
                        es.advance();
                    }
                }
            }
            store_vec(entry + static_cast<size_t>(c) * 4, v);   // the state at `end`
        }
        if (pos > ch.begin) {   // state at pos - 1 from the state at pos (emission of pos), for pos down to begin + 1
            EmissionStream<false> es;
            es.start(emit, pos, ch.begin + 1, ch.run_end);
            while (pos > ch.begin) {
                Vec f[8];   // the forward vectors of the block's positions, requested together
#pragma unroll
                for (int k = 0; k < 8; ++k) f[k] = load_vec(fwd + (pos >= ch.begin + 1 + static_cast<uint64_t>(k) ? pos - 1 - k : ch.begin) * 4);
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (pos >= ch.begin + 1 + static_cast<uint64_t>(k)) {
                        v = backward_step(h, v, es.cur[k]);
                        double total = 0.0;
#pragma unroll
                        for (int s = 0; s < 4; ++s) total += (f[k].s[s] * v.s[s]);
                        post[pos - 1 - k] = (f[k].s[0] * v.s[0]) / total;
                    }
                pos = pos >= ch.begin + 8 ? pos - 8 : ch.begin;
                es.advance();
            }
        }
        store_vec(exit_state + static_cast<size_t>(c) * 4, v);   // the state at begin
    }
}

// same[c] = 1 when chunk c's entry state equals, bit for bit, the exit state of the chunk it depends on (forward / Viterbi: c - 1,
// backward: c + 1); chunks that begin (end) their run depend on nothing
__global__ __launch_bounds__(256) void hmm_compare_kernel(const HmmChunk* __restrict__ chunks, uint32_t n_chunks, bool backward, const double* __restrict__ entry,
                                                          const double* __restrict__ exit_state, uint8_t* __restrict__ same)
{
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    if (c >= n_chunks) return;
    const HmmChunk ch = chunks[c];
    const bool free_end = backward ? ch.end == ch.run_end : ch.begin == ch.run_begin;
    bool eq = true;
    if (!free_end) {
        const unsigned long long* a = reinterpret_cast<const unsigned long long*>(entry + static_cast<size_t>(c) * 4);
        const unsigned long long* b = reinterpret_cast<const unsigned long long*>(exit_state + static_cast<size_t>(backward ? c + 1 : c - 1) * 4);
#pragma unroll
        for (int s = 0; s < 4; ++s) eq = eq && a[s] == b[s];
    }
    same[c] = eq ? 1 : 0;
}

// bins[i] = the number of thresholds <= posterior[i] (thresholds ascending): for a caller that only PRINTS a monotone function of the
// posterior, with the thresholds at the values where the printed text changes, this is all it needs -- 2 bytes per position instead of
// 8, and no log10 per value on the host.  NaN has no bin (0xffff, with the sign bit set 0xfffe): its position and value go to the exception list.
__global__ __launch_bounds__(256) void hmm_bin_kernel(const double* __restrict__ post, uint64_t n, const double* __restrict__ thresholds, uint32_t n_thresholds,
                                                      uint16_t* __restrict__ bins, uint64_t* __restrict__ exc_index, double* __restrict__ exc_value,
                                                      uint32_t exc_capacity, unsigned int* __restrict__ n_exc)
{
    const uint64_t i = static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x;
    if (i >= n) return;
    const double p = post[i];
    if (p != p) {
        const unsigned slot = atomicAdd(n_exc, 1u);
        if (slot < exc_capacity) {
            exc_index[slot] = i;
            exc_value[slot] = p;
        }
        bins[i] = static_cast<uint16_t>(0xffffu - (__double_as_longlong(p) < 0 ? 1u : 0u));   // 0xffff: NaN, 0xfffe: NaN with the sign bit set
        return;
    }
    uint32_t lo = 0, hi = n_thresholds;
    while (lo < hi) {
        const uint32_t mid = (lo + hi) >> 1;
        if (thresholds[mid] <= p) lo = mid + 1;
        else hi = mid;
    }
    bins[i] = static_cast<uint16_t>(lo);
}

hipError_t launch_hmm_bins(const double* d_post, uint64_t n, const double* d_thresholds, uint32_t n_thresholds, uint16_t* d_bins, uint64_t* d_exc_index,
                           double* d_exc_value, uint32_t exc_capacity, unsigned int* d_n_exc, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    if (n_thresholds >= 0xfffeu || (n + 255u) / 256u > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(hmm_bin_kernel, dim3(static_cast<unsigned>((n + 255u) / 256u)), dim3(256), 0, stream, d_post, n, d_thresholds, n_thresholds, d_bins, d_exc_index,
                       d_exc_value, exc_capacity, d_n_exc);
    return hipGetLastError();
}

hipError_t launch_hmm_chunks(int mode, const HmmModel& h, const HmmChunk* d_chunks, const uint32_t* d_todo, uint32_t n_todo, bool rerun, uint32_t burn_in,
                             const double* d_emit, double* d_fwd, uint8_t* d_back, double* d_post, double* d_entry, double* d_exit, hipStream_t stream)
{
    if (n_todo == 0) return hipSuccess;
    const dim3 grid((n_todo + 63u) / 64u), block(64);
    switch (mode) {
        case 0: hipLaunchKernelGGL(hmm_chunks_kernel<0>, grid, block, 0, stream, h, d_chunks, d_todo, n_todo, rerun, burn_in, d_emit, d_fwd, d_back, d_post, d_entry, d_exit); break;
        case 1: hipLaunchKernelGGL(hmm_chunks_kernel<1>, grid, block, 0, stream, h, d_chunks, d_todo, n_todo, rerun, burn_in, d_emit, d_fwd, d_back, d_post, d_entry, d_exit); break;
        case 2: hipLaunchKernelGGL(hmm_chunks_kernel<2>, grid, block, 0, stream, h, d_chunks, d_todo, n_todo, rerun, burn_in, d_emit, d_fwd, d_back, d_post, d_entry, d_exit); break;
        default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_hmm_compare(const HmmChunk* d_chunks, uint32_t n_chunks, bool backward, const double* d_entry, const double* d_exit, uint8_t* d_same, hipStream_t stream)
{
    if (n_chunks == 0) return hipSuccess;
    hipLaunchKernelGGL(hmm_compare_kernel, dim3((n_chunks + 255u) / 256u), dim3(256), 0, stream, d_chunks, n_chunks, backward, d_entry, d_exit, d_same);
    return hipGetLastError();
}

}  // namespace pcsf
