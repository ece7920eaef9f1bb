// ctx.hpp -- the context object behind the C-ABI handle (shared by api.cpp and blocks.cpp)
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/phylocsf_amd.h"
#include "hostmath.hpp"
#include "kernels.h"
#include "schedule.hpp"

namespace pcsf {

struct DeviceBuffer {
    void* ptr = nullptr;
    size_t bytes = 0;
    DeviceBuffer() = default;
    DeviceBuffer(const DeviceBuffer&) = delete;
    DeviceBuffer& operator=(const DeviceBuffer&) = delete;
    ~DeviceBuffer() { release(); }
    void release()
    {
        if (ptr) (void)hipFree(ptr);
        ptr = nullptr;
        bytes = 0;
    }
    hipError_t reserve(size_t n)
    {
        if (n <= bytes) return hipSuccess;
        release();
        // head room: consecutive batches differ by a few per cent, and growing by exactly that much every time
        // would free and allocate (and synchronise the device) on most calls
        const size_t roomy = n + n / 4;
        hipError_t e = hipMalloc(&ptr, roomy);
        if (e == hipSuccess) {
            bytes = roomy;
            return e;
        }
        (void)hipGetLastError();
        e = hipMalloc(&ptr, n);
        if (e == hipSuccess) bytes = n;
        return e;
    }
    // exactly n bytes (tables of many gigabytes: no head room)
    hipError_t reserve_exact(size_t n)
    {
        if (n <= bytes) return hipSuccess;
        release();
        const hipError_t e = hipMalloc(&ptr, n);
        if (e == hipSuccess) bytes = n;
        else ptr = nullptr;
        return e;
    }
    // like reserve(), but the first `keep` bytes of the old contents survive a growth (device-to-device copy on
    // `stream`, which the caller's later work on that stream is ordered behind)
    hipError_t reserve_keep(size_t n, size_t keep, hipStream_t stream)
    {
        if (n <= bytes) return hipSuccess;
        if (!ptr || keep == 0) return reserve(n);
        void* old = ptr;
        const size_t old_bytes = bytes;
        ptr = nullptr;
        bytes = 0;
        hipError_t e = reserve(n);
        if (e == hipSuccess) e = hipMemcpyAsync(ptr, old, keep < old_bytes ? keep : old_bytes, hipMemcpyDeviceToDevice, stream);
        if (e == hipSuccess) e = hipStreamSynchronize(stream);  // the old allocation is freed right below
        (void)hipFree(old);
        return e;
    }
    template <class T> T* as() const { return static_cast<T*>(ptr); }
};

// Streams of the block engines (blocks.cpp: a main and a side stream per engine, one engine per group of blocks).  The first use of
// a new stream makes the runtime create a hardware queue, ~11 ms each on this machine -- 35 ms of a 55 ms MLE call on 516 blocks
// when two groups created theirs on the spot.  The pool keeps streams across calls, and warm_up() creates them on a thread of its own
// (pcsf_prepare_block_scoring: the command line starts it before it reads the alignment file).
struct StreamPool {
    std::mutex lock;
    std::vector<hipStream_t> idle;
    std::thread warming;
    explicit StreamPool(bool high = false) : high_priority(high) {}
    bool high_priority = false;   // the pool of control streams (blocks.cpp run_chain): created at the device's highest stream priority
    static hipError_t make_stream(hipStream_t* s, bool high)
    {
        if (high) {
            int least = 0, greatest = 0;
            if (hipDeviceGetStreamPriorityRange(&least, &greatest) == hipSuccess && greatest != least)
                return hipStreamCreateWithPriority(s, hipStreamNonBlocking, greatest);
            (void)hipGetLastError();
        }
        return hipStreamCreateWithFlags(s, hipStreamNonBlocking);
    }
    void warm_up(int device, int count)
    {
        std::lock_guard<std::mutex> g(lock);
        if (warming.joinable() || !idle.empty()) return;
        warming = std::thread([this, device, count]() {
            if (hipSetDevice(device) != hipSuccess) return;
            void* scratch = nullptr;
            if (hipMalloc(&scratch, 256) != hipSuccess) return;
            std::vector<hipStream_t> made;
            for (int i = 0; i < count; ++i) {
                hipStream_t s = nullptr;
                if (make_stream(&s, high_priority) != hipSuccess) break;
                (void)hipMemsetAsync(scratch, 0, 256, s);   // the queue behind a stream is created at its first operation
                made.push_back(s);
            }
            for (hipStream_t s : made) (void)hipStreamSynchronize(s);
            (void)hipFree(scratch);
            (void)hipGetLastError();
            std::lock_guard<std::mutex> g2(lock);
            idle.insert(idle.end(), made.begin(), made.end());
        });
    }
    void wait_warm()
    {
        std::thread t;
        {
            std::lock_guard<std::mutex> g(lock);
            t.swap(warming);
        }
        if (t.joinable()) t.join();
    }
    hipError_t acquire(hipStream_t* out)
    {
        wait_warm();
        {
            std::lock_guard<std::mutex> g(lock);
            if (!idle.empty()) {
                *out = idle.back();
                idle.pop_back();
                return hipSuccess;
            }
        }
        return make_stream(out, high_priority);
    }
    void release(hipStream_t s)   // the caller has synchronised it
    {
        std::lock_guard<std::mutex> g(lock);
        idle.push_back(s);
    }
    ~StreamPool()
    {
        wait_warm();
        for (hipStream_t s : idle) (void)hipStreamDestroy(s);
    }
};

}  // namespace pcsf

struct pcsf_ctx {
    int device = 0;
    std::string error;
    pcsf::Schedule schedule;
    int n_nodes = 0, n_leaves = 0;
    std::vector<int16_t> child1, child2;
    std::vector<float> branch_len;
    std::vector<int> branch_slot;        // branch (node id) -> fragment slot of the schedule, -1 for leaves
    pcsf::Mat S[2];
    pcsf::Vec pi[2];
    pcsf::EigenSystem eig[2];            // coding, non-coding ECM (tree scale independent)
    std::vector<double> host_P;          // [2][n-1][64*64] at tree scale 1
    // device-resident model (table sets 0 = coding, 1 = non-coding at tree scale 1)
    pcsf::DeviceBuffer d_program, d_frag, d_leaf_tab, d_prior, d_anc_weight;
    pcsf::DeviceBuffer d_op_leaves, d_op_meta;  // schedule.hpp ProgramFacts of the program (empty for trees of more than 128 leaves)
    pcsf::DeviceBuffer d_gap_tab;               // [2][n_edges + 1][64] messages of all-gap subtrees, recorded by the full kernel (kernels.hip SKIP)
    // Cherry tables (kernels.hip PAIRS, schedule.hpp reduce_cherries): what tracks mode runs when the tree has cherries -- the program of
    // the tree whose cherries are leaves, its fragment images, the leaves' rows followed by the cherries' pair rows, its all-gap messages.
    // The block kernels (per-evaluation branch lengths) keep the full program above.
    struct CherryTables {
        bool on = false;
        pcsf::Schedule schedule;
        std::vector<std::array<int, 3>> pairs, triples;
        int n_free_pairs = 0;               // pairs[0 .. n_free_pairs) are operands of the program, the others lie inside triples / quads
        int n_free_triples = 0, n_quads = 0;
        uint32_t leaf_rows = 0;             // per table set: n_leaves * 65 + pairs * 65^2 + triples * 65^3
        pcsf::DeviceBuffer d_program, d_op_leaves, d_op_meta, d_frag, d_leaf_tab, d_gap_tab, d_pair_leaves, d_trip_info, d_quad_info;
        void reset()
        {
            on = false;
            pairs.clear();
            triples.clear();
            n_free_pairs = n_free_triples = n_quads = 0;
            leaf_rows = 0;
            for (pcsf::DeviceBuffer* b : {&d_program, &d_op_leaves, &d_op_meta, &d_frag, &d_leaf_tab, &d_gap_tab, &d_pair_leaves, &d_trip_info, &d_quad_info}) b->release();
        }
    } cherry;
    // scratch for the host-pointer entry points
    pcsf::DeviceBuffer d_codons, d_out, d_lnz, d_anc, d_stamps, d_nucl, d_segments, d_wg_seg;
    bool dump_codons = false;            // diagnostics (PCSF_DUMP_CODONS=1): the fused front end also writes its codons to d_codons
    size_t last_total_sites = 0;
    bool fused_frontend = true;          // six-frame translation inside the pruning kernel (PCSF_FUSED_FRONTEND=0: separate kernel)
    bool skip_allgap = true;             // tracks mode skips all-gap subtrees per work-group (kernels.hip SKIP)
    bool skip_allgap_forced = false;     // PCSF_SKIP_ALLGAP given: `skip_allgap` for every entry point; else on for the six-frame ones only
    // branch-length score (pcsf_set_bls_branch_lengths)
    pcsf::DeviceBuffer d_bls_tree, d_bls_out, d_bls_spans, d_bls_flag;
    // pcsf_hmm_smooth_runs (hmm_abi.cpp): kept between calls -- a call on a 10 Mb chromosome holds ~1 GB, and allocating and freeing that
    // around every call cost as much as its kernels
    struct HmmBuffers {
        pcsf::DeviceBuffer emit, fwd, back, post, chunks, entry, exit_state, same, todo, thresholds, bins, exc_index, exc_value, exc_count;
    } hmm;
    // pcsf_score_alignments_6frames_wig: the text of the call's seven files stays on the device until pcsf_fetch_wig_text
    struct WigBuffers {
        pcsf::DeviceBuffer frames, chroms, lens, offsets, temp, text, flag;
        uint32_t* h_bounds = nullptr;      // page-locked: byte offsets of the eight file boundaries, then the needs-host flag
        uint64_t file_off[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        bool ready = false;
    } wig;
    // pcsf_wig_retain (comm.cpp): the text of six-frame wig calls kept on the device, in the caller's order, until the ranks' gather
    struct WigRetain {
        bool on = false;
        pcsf::DeviceBuffer arena;          // the kept calls' texts, each as the call left it (seven files back to back)
        uint64_t used = 0;
        struct Record {
            uint64_t order;                // the caller's key: records are gathered in ascending order
            uint64_t at;                   // offset of the record's text in the arena
            uint64_t file_off[8];          // its seven files' boundaries, relative to `at`
        };
        std::vector<Record> records;
    } retain;
    struct CommState;                      // comm.cpp: the RCCL communicator of a multi-process run (library loaded on demand)
    CommState* comm = nullptr;
    void (*comm_free)(CommState*) = nullptr;
    pcsf::BlsTree bls_tree{};
    bool bls_ready = false;
    unsigned long long* bls_first_bad = nullptr;  // page-locked host copy of the kernel's character check
    uint32_t stamp_block = 0;
    hipStream_t stream = nullptr;        // private stream for host-pointer entry points
    pcsf::StreamPool block_streams;      // streams of the MLE / OMEGA engines (blocks.cpp)
    pcsf::StreamPool control_streams{true};   // ... and their control streams (highest priority: the small kernels between two pruning launches)
    int block_kernel = 0;                // pcsf_block_kernel: AUTO = every pcsf_score_blocks call decides from its own list
    hipEvent_t ev_kernels = nullptr;     // behind the kernels of this context's last six-frame call
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    bool timing = false;
    double last_kernel_ms = 0.0;
    int variant = 0;
    size_t extra_lds = 0;                // experiment knob (PCSF_EXTRA_LDS): lowers occupancy
    int prio_mode = 0;                   // experiment knob (PCSF_PRIO_MODE), see kernels.h
    int stagger_sleep = 0;               // x1024 cycles late start of the odd wave slot (PCSF_STAGGER overrides)
    uint32_t stagger_first_blocks = 0;

    ~pcsf_ctx()
    {
        if (comm && comm_free) comm_free(comm);
        if (bls_first_bad) (void)hipHostFree(bls_first_bad);
        if (wig.h_bounds) (void)hipHostFree(wig.h_bounds);
        if (ev_kernels) (void)hipEventDestroy(ev_kernels);
        if (ev0) (void)hipEventDestroy(ev0);
        if (ev1) (void)hipEventDestroy(ev1);
        if (stream) (void)hipStreamDestroy(stream);
    }
    int fail(int code, const std::string& msg)
    {
        error = msg;
        return code;
    }
    int hip_fail(hipError_t e, const char* what)
    {
        error = std::string(what) + ": " + hipGetErrorString(e);
        return PCSF_ERR_HIP;
    }
};

namespace pcsf {
// rows of `width` bytes, `rows` of them, host pitch src_pitch -> device pitch `width`: one linear copy when the host
// matrix is packed (measured 0.6 ms faster than the 2-D copy for a 20 MB matrix), a 2-D copy otherwise
inline hipError_t upload_rows(void* dst, const void* src, size_t src_pitch, size_t width, size_t rows, hipStream_t stream)
{
    if (src_pitch == width || rows <= 1) return hipMemcpyAsync(dst, src, width * rows, hipMemcpyHostToDevice, stream);
    return hipMemcpy2DAsync(dst, width, src, src_pitch, width, rows, hipMemcpyHostToDevice, stream);
}
// fills the kernel parameters that describe the resident model (tables of sets 0 and 1): the full program (block kernels, per-block tables)
PruneParams base_prune_params(const pcsf_ctx* c);
// ... and what tracks mode runs: the cherry-table form when the context has one (*variant_flags |= 256), else the same
PruneParams tracks_prune_params(const pcsf_ctx* c, int* variant_flags);
// the six-frame front end fused into the pruning kernel: device pointers of one call
struct FusedFrontEnd {
    const unsigned char* nucl;
    size_t nucl_stride;
    const FrameSegment* segments;
    const uint32_t* wg_first_seg;
    unsigned long long* first_bad;
    unsigned char* dump_codons = nullptr;  // diagnostics
};
// launches the tracks-mode kernel over `n_sites` device-resident sites (both models); fused != null: the sites are
// the codons of fused->segments, formed in the kernel (d_codons is not read)
int launch_fixed(pcsf_ctx* c, const uint8_t* d_codons, size_t n_sites, size_t leaf_stride, double* d_out, double* d_lnz,
                 double* d_anc, hipStream_t stream, const FusedFrontEnd* fused = nullptr);
// the AUTO rule of pcsf_block_kernel on a list (blocks.cpp): true = prune_tiles
bool list_prefers_tiles(size_t n_leaves, const uint8_t* codons, size_t stride, const uint64_t* off, size_t n_blocks);
// run() for a batch of blocks (blocks.cpp)
int score_blocks_impl(pcsf_ctx* c, int algo, const uint8_t* codons, const uint64_t* site_offsets, size_t n_blocks,
                      size_t leaf_stride, int compute_anc, uint32_t seed, float* out_phylo, float* out_anc);
int block_loglik_impl(pcsf_ctx* c, int model, const double* q_settings, const double* tree_scale, const uint8_t* codons,
                      const uint64_t* site_offsets, size_t n_blocks, size_t leaf_stride, double* out_lpr, int32_t* out_failed,
                      const EigenSystem* custom = nullptr, size_t n_custom = 0);
int debug_eigen_omega_impl(pcsf_ctx* c, const double* q_settings, const double* warm_settings, size_t n, double* right, double* left,
                           double* lambda, double* prior, int32_t* failed);
}  // namespace pcsf
