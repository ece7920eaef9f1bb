// comm.cpp -- build-tracks as one process per GPU: the ranks' wig text kept on the device and gathered over RCCL.
//
// What it replaces: the reference's parallel loop over byte-range jobs, each writing "<track>.wig.<job_id>", and the merge of those
// files in job order (src/phylocsf++build_tracks.hpp:88, :93-124, merge_job_output_files :27-53, :246-259).  Here a rank's text never
// leaves its GPU until the gather: pcsf_score_alignments_6frames_wig formats it there (wig.hip), pcsf_wig_retain_last copies it into
// the context's arena, and pcsf_comm_gather_wig moves every rank's seven texts to the root's GPU with one grouped ncclSend / ncclRecv
// (xGMI between the GPUs of a node; the point-to-point form of a gather with per-rank sizes).
//
// RCCL is loaded with dlopen when a communicator is first asked for: the single-GPU command line must not pay for mapping it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstring>
#include <mutex>
#include <string>
#include <vector>

#include "ctx.hpp"

namespace pcsf {
void set_global_error(const std::string& msg);   // api.cpp: what pcsf_last_error(NULL) returns
}

namespace {

struct Rccl {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    std::string error;
};

// the library, loaded once per process; null with `why` set when it is not there
Rccl* rccl(std::string* why)
{
    static std::mutex lock;
    static Rccl r;
    static bool tried = false;
    std::lock_guard<std::mutex> g(lock);
    if (!tried) {
        tried = true;
        const char* names[] = {std::getenv("PCSF_RCCL_LIBRARY"), "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* n : names) {
            if (!n || !*n) continue;
            if ((r.lib = dlopen(n, RTLD_NOW | RTLD_LOCAL)) != nullptr) break;
            r.error = dlerror();
        }
        if (r.lib) {
            auto sym = [&](const char* name) {
                void* p = dlsym(r.lib, name);
                if (!p) r.error = std::string("librccl: no symbol ") + name;
                return p;
            };
            r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
            r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
            r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
            r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
            r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
            r.Send = reinterpret_cast<decltype(r.Send)>(sym("ncclSend"));
            r.Recv = reinterpret_cast<decltype(r.Recv)>(sym("ncclRecv"));
            r.GroupStart = reinterpret_cast<decltype(r.GroupStart)>(sym("ncclGroupStart"));
            r.GroupEnd = reinterpret_cast<decltype(r.GroupEnd)>(sym("ncclGroupEnd"));
            if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.GetErrorString || !r.AllGather || !r.Send || !r.Recv || !r.GroupStart || !r.GroupEnd) {
                dlclose(r.lib);
                r.lib = nullptr;
            }
        }
    }
    if (!r.lib) {
        if (why) *why = "RCCL is not available (" + r.error + ")";
        return nullptr;
    }
    return &r;
}

}  // namespace

struct pcsf_ctx::CommState {
    Rccl* api = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
    pcsf::DeviceBuffer sizes_in, sizes_all, send, recv;
    std::vector<uint64_t> bytes;      // [world][7] of the last gather
    std::vector<uint64_t> recv_at;    // [world]: where a sender's text starts in `recv` (root)
    std::mutex fetch_lock;            // pcsf_comm_fetch_gathered may be called from the seven writers at once
};

static void comm_state_free(pcsf_ctx::CommState* s)
{
    if (!s) return;
    if (s->comm && s->api) (void)s->api->CommDestroy(s->comm);
    delete s;
}

static int nccl_fail(pcsf_ctx* c, const Rccl* api, ncclResult_t r, const char* what)
{
    return c->fail(PCSF_ERR_HIP, std::string(what) + ": " + api->GetErrorString(r));
}

extern "C" {

int pcsf_comm_unique_id(void* id)
{
    if (!id) return PCSF_ERR_INVALID_ARG;
    std::string why;
    Rccl* api = rccl(&why);
    if (!api) {
        pcsf::set_global_error("pcsf_comm_unique_id: " + why);
        return PCSF_ERR_UNSUPPORTED;
    }
    ncclUniqueId u;
    static_assert(sizeof(u) == PCSF_COMM_ID_BYTES, "PCSF_COMM_ID_BYTES is ncclUniqueId's size");
    const ncclResult_t r = api->GetUniqueId(&u);
    if (r != ncclSuccess) {
        pcsf::set_global_error(std::string("ncclGetUniqueId: ") + api->GetErrorString(r));
        return PCSF_ERR_HIP;
    }
    memcpy(id, &u, sizeof u);
    return PCSF_OK;
}

int pcsf_comm_create(pcsf_ctx* c, int rank, int world, const void* id)
{
    if (!c) return PCSF_ERR_INVALID_ARG;
    if (!id || world < 1 || rank < 0 || rank >= world) return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_comm_create: bad argument");
    if (c->comm) return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_comm_create: this context already has a communicator");
    std::string why;
    Rccl* api = rccl(&why);
    if (!api) return c->fail(PCSF_ERR_UNSUPPORTED, "pcsf_comm_create: " + why);
    hipError_t e;
    if ((e = hipSetDevice(c->device)) != hipSuccess) return c->hip_fail(e, "hipSetDevice");
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclComm_t comm = nullptr;
    const ncclResult_t r = api->CommInitRank(&comm, world, u, rank);
    if (r != ncclSuccess) return nccl_fail(c, api, r, "ncclCommInitRank");
    pcsf_ctx::CommState* s = new pcsf_ctx::CommState;
    s->api = api;
    s->comm = comm;
    s->rank = rank;
    s->world = world;
    c->comm = s;
    c->comm_free = comm_state_free;
    return PCSF_OK;
}

int pcsf_comm_destroy(pcsf_ctx* c)
{
    if (!c) return PCSF_ERR_INVALID_ARG;
    if (c->comm) {
        (void)hipSetDevice(c->device);
        comm_state_free(c->comm);
    }
    c->comm = nullptr;
    return PCSF_OK;
}

// room for `more` bytes behind the arena's used part; what is there survives
static int arena_room(pcsf_ctx* c, uint64_t more)
{
    pcsf_ctx::WigRetain& R = c->retain;
    if (R.used + more <= R.arena.bytes) return PCSF_OK;
    const size_t want = std::max<size_t>(static_cast<size_t>(R.used + more) * 2, size_t(64) << 20);
    const hipError_t e = R.arena.reserve_keep(want, static_cast<size_t>(R.used), c->stream);
    return e == hipSuccess ? PCSF_OK : c->hip_fail(e, "hipMalloc(kept wig text)");
}

int pcsf_wig_retain_last(pcsf_ctx* c, uint64_t order)
{
    if (!c) return PCSF_ERR_INVALID_ARG;
    if (!c->wig.ready) return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_wig_retain_last: no text (call pcsf_score_alignments_6frames_wig first)");
    hipError_t e;
    if ((e = hipSetDevice(c->device)) != hipSuccess) return c->hip_fail(e, "hipSetDevice");
    const uint64_t first = c->wig.file_off[0], n = c->wig.file_off[7] - first;
    int rc;
    if ((rc = arena_room(c, n)) != PCSF_OK) return rc;
    pcsf_ctx::WigRetain& R = c->retain;
    // stream order: behind the kernels that wrote the text (a second write pass included), before the next call's
    if (n && (e = hipMemcpyAsync(R.arena.as<char>() + R.used, c->wig.text.as<char>() + first, n, hipMemcpyDeviceToDevice, c->stream)) != hipSuccess)
        return c->hip_fail(e, "hipMemcpyAsync(kept wig text)");
    pcsf_ctx::WigRetain::Record rec{};
    rec.order = order;
    rec.at = R.used;
    for (int f = 0; f < 8; ++f) rec.file_off[f] = c->wig.file_off[f] - first;
    R.records.push_back(rec);
    R.used += n;
    return PCSF_OK;   // (the call's text can still be fetched: keeping it does not consume it)
}

int pcsf_wig_retain_host(pcsf_ctx* c, uint64_t order, const char* const* text, const uint64_t* bytes)
{
    if (!c) return PCSF_ERR_INVALID_ARG;
    if (!text || !bytes) return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_wig_retain_host: null argument");
    hipError_t e;
    if ((e = hipSetDevice(c->device)) != hipSuccess) return c->hip_fail(e, "hipSetDevice");
    uint64_t n = 0;
    for (int f = 0; f < 7; ++f) {
        if (bytes[f] && !text[f]) return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_wig_retain_host: null text for a file with bytes");
        n += bytes[f];
    }
    int rc;
    if ((rc = arena_room(c, n)) != PCSF_OK) return rc;
    pcsf_ctx::WigRetain& R = c->retain;
    pcsf_ctx::WigRetain::Record rec{};
    rec.order = order;
    rec.at = R.used;
    uint64_t at = 0;
    for (int f = 0; f < 7; ++f) {
        rec.file_off[f] = at;
        if (bytes[f] && (e = hipMemcpyAsync(R.arena.as<char>() + R.used + at, text[f], bytes[f], hipMemcpyHostToDevice, c->stream)) != hipSuccess) {
            (void)hipStreamSynchronize(c->stream);
            return c->hip_fail(e, "hipMemcpyAsync(kept wig text)");
        }
        at += bytes[f];
    }
    rec.file_off[7] = at;
    if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return c->hip_fail(e, "hipStreamSynchronize");   // the caller's buffers are its own again
    R.records.push_back(rec);
    R.used += n;
    return PCSF_OK;
}

int pcsf_wig_retained_bytes(pcsf_ctx* c, uint64_t* bytes)
{
    if (!c) return PCSF_ERR_INVALID_ARG;
    if (!bytes) return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_wig_retained_bytes: null argument");
    for (int f = 0; f < 7; ++f) bytes[f] = 0;
    for (const auto& r : c->retain.records)
        for (int f = 0; f < 7; ++f) bytes[f] += r.file_off[f + 1] - r.file_off[f];
    return PCSF_OK;
}

int pcsf_wig_retain_reset(pcsf_ctx* c)
{
    if (!c) return PCSF_ERR_INVALID_ARG;
    c->retain.records.clear();
    c->retain.used = 0;
    return PCSF_OK;
}

int pcsf_comm_gather_wig(pcsf_ctx* c, int root, int include_root, uint64_t* bytes_out)
{
    if (!c) return PCSF_ERR_INVALID_ARG;
    pcsf_ctx::CommState* s = c->comm;
    if (!s) return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_comm_gather_wig: no communicator (pcsf_comm_create)");
    if (root < 0 || root >= s->world) return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_comm_gather_wig: bad root");
    const Rccl* api = s->api;
    hipError_t e;
    ncclResult_t r;
    if ((e = hipSetDevice(c->device)) != hipSuccess) return c->hip_fail(e, "hipSetDevice");
    const int W = s->world;
    const bool sender = s->rank != root || include_root;
    // ---- the sizes: seven per rank, all-gathered
    uint64_t mine[7];
    pcsf_wig_retained_bytes(c, mine);
    if (!sender) for (uint64_t& v : mine) v = 0;
    if ((e = s->sizes_in.reserve(7 * sizeof(uint64_t))) != hipSuccess || (e = s->sizes_all.reserve(static_cast<size_t>(W) * 7 * sizeof(uint64_t))) != hipSuccess)
        return c->hip_fail(e, "hipMalloc(gather sizes)");
    if ((e = hipMemcpyAsync(s->sizes_in.ptr, mine, sizeof mine, hipMemcpyHostToDevice, c->stream)) != hipSuccess) return c->hip_fail(e, "hipMemcpyAsync(gather sizes)");
    if ((r = api->AllGather(s->sizes_in.ptr, s->sizes_all.ptr, 7, ncclUint64, s->comm, c->stream)) != ncclSuccess) return nccl_fail(c, api, r, "ncclAllGather");
    s->bytes.assign(static_cast<size_t>(W) * 7, 0);
    if ((e = hipMemcpyAsync(s->bytes.data(), s->sizes_all.ptr, s->bytes.size() * sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream)) != hipSuccess)
        return c->hip_fail(e, "hipMemcpyAsync(gather sizes)");
    if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return c->hip_fail(e, "hipStreamSynchronize(gather sizes)");
    if (bytes_out) memcpy(bytes_out, s->bytes.data(), s->bytes.size() * sizeof(uint64_t));
    auto rank_total = [&](int k) {
        uint64_t t = 0;
        for (int f = 0; f < 7; ++f) t += s->bytes[static_cast<size_t>(k) * 7 + f];
        return t;
    };
    // ---- a sender's message: its seven files back to back, each file's kept pieces in the caller's order
    const uint64_t my_total = rank_total(s->rank);
    if (sender && my_total) {
        pcsf_ctx::WigRetain& R = c->retain;
        std::vector<const pcsf_ctx::WigRetain::Record*> recs;
        for (const auto& rec : R.records) recs.push_back(&rec);
        std::stable_sort(recs.begin(), recs.end(), [](const auto* a, const auto* b) { return a->order < b->order; });
        if ((e = s->send.reserve(my_total)) != hipSuccess) return c->hip_fail(e, "hipMalloc(gather send buffer)");
        uint64_t at = 0;
        for (int f = 0; f < 7; ++f)
            for (const auto* rec : recs) {
                const uint64_t n = rec->file_off[f + 1] - rec->file_off[f];
                if (n && (e = hipMemcpyAsync(s->send.as<char>() + at, R.arena.as<char>() + rec->at + rec->file_off[f], n, hipMemcpyDeviceToDevice, c->stream)) != hipSuccess)
                    return c->hip_fail(e, "hipMemcpyAsync(gather send buffer)");
                at += n;
            }
    }
    // ---- the root's receive buffer: the senders' messages in rank order
    s->recv_at.assign(static_cast<size_t>(W) + 1, 0);
    for (int k = 0; k < W; ++k) s->recv_at[k + 1] = s->recv_at[k] + rank_total(k);
    if (s->rank == root && s->recv_at[W] && (e = s->recv.reserve(s->recv_at[W])) != hipSuccess) return c->hip_fail(e, "hipMalloc(gather receive buffer)");
    // ---- one group: every sender's ncclSend, the root's ncclRecv per sender (a rank without text posts nothing, and the root knows)
    if ((r = api->GroupStart()) != ncclSuccess) return nccl_fail(c, api, r, "ncclGroupStart");
    ncclResult_t first_bad = ncclSuccess;
    if (sender && my_total && (r = api->Send(s->send.ptr, my_total, ncclChar, root, s->comm, c->stream)) != ncclSuccess) first_bad = r;
    if (s->rank == root)
        for (int k = 0; k < W; ++k) {
            const uint64_t n = rank_total(k);
            if (n && (r = api->Recv(s->recv.as<char>() + s->recv_at[k], n, ncclChar, k, s->comm, c->stream)) != ncclSuccess && first_bad == ncclSuccess) first_bad = r;
        }
    if ((r = api->GroupEnd()) != ncclSuccess && first_bad == ncclSuccess) first_bad = r;
    if (first_bad != ncclSuccess) {
        (void)hipStreamSynchronize(c->stream);
        return nccl_fail(c, api, first_bad, "ncclSend / ncclRecv (wig gather)");
    }
    // no rank leaves before the root has everything: a sender's buffers (and its process) must outlive the root's receive.  A second
    // tiny all-gather on the same stream completes on a rank only once every rank has reached it -- the root behind its receives
    if ((r = api->AllGather(s->sizes_in.ptr, s->sizes_all.ptr, 1, ncclUint64, s->comm, c->stream)) != ncclSuccess) return nccl_fail(c, api, r, "ncclAllGather(closing)");
    if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return c->hip_fail(e, "hipStreamSynchronize(wig gather)");
    pcsf_wig_retain_reset(c);
    return PCSF_OK;
}

int pcsf_comm_fetch_gathered(pcsf_ctx* c, int rank, int file, char* dst)
{
    if (!c) return PCSF_ERR_INVALID_ARG;
    pcsf_ctx::CommState* s = c->comm;
    if (!s || s->bytes.empty()) return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_comm_fetch_gathered: nothing gathered (pcsf_comm_gather_wig)");
    if (rank < 0 || rank >= s->world || file < 0 || file > 6) return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_comm_fetch_gathered: bad argument");
    const uint64_t* b = s->bytes.data() + static_cast<size_t>(rank) * 7;
    if (!b[file]) return PCSF_OK;
    if (!dst) return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_comm_fetch_gathered: null destination");
    if (!s->recv.ptr || s->recv_at[rank + 1] > s->recv.bytes) return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_comm_fetch_gathered: not the root of the last gather");
    uint64_t at = s->recv_at[rank];
    for (int f = 0; f < file; ++f) at += b[f];
    std::lock_guard<std::mutex> g(s->fetch_lock);
    hipError_t e;
    if ((e = hipSetDevice(c->device)) != hipSuccess) return c->hip_fail(e, "hipSetDevice");
    // a blocking copy on the null stream of this thread would do; the context's stream keeps every operation of a context in one order
    if ((e = hipMemcpyAsync(dst, s->recv.as<char>() + at, b[file], hipMemcpyDeviceToHost, c->stream)) != hipSuccess || (e = hipStreamSynchronize(c->stream)) != hipSuccess)
        return c->hip_fail(e, "hipMemcpy(gathered wig text)");
    return PCSF_OK;
}

}  // extern "C"
