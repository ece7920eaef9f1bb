// api.cpp -- the C-ABI of include/phylocsf_amd.h on top of the gfx950 kernels.
// Host side of the drop-in boundary: what the reference does in PhyloCSFModel_make /
// lpr_leaves / run_tracks / run (src/instance.hpp:687-712, src/fixed_lik.hpp:362-449,
// src/run.hpp:35-211) minus the per-site arithmetic, which runs on the GPU.
#include <chrono>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <new>
#include <mutex>
#include <random>
#include <string>
#include <vector>

#include "ctx.hpp"
#include "frames.hpp"
#include "optimizer.hpp"

namespace {
thread_local std::string g_create_error;

// Contexts that share a GPU run their six-frame calls' KERNELS one call after the other, in the order the calls
// arrive here, while the transfers of one call overlap the kernels of the other: two pruning launches resident at
// the same time only slow each other down (measured), a transfer under a kernel is free.  Per device: the event
// recorded behind the kernels of the call that was enqueued last.
struct KernelTurn {
    std::mutex lock;
    hipEvent_t last = nullptr;
    const pcsf_ctx* owner = nullptr;
};
KernelTurn g_kernel_turn[64];
}  // namespace

using pcsf::DeviceBuffer;

namespace pcsf {
void set_global_error(const std::string& msg) { g_create_error = msg; }
}  // namespace pcsf

namespace {

// uploads fragment images / leaf tables / priors for a set of branch matrices P[2][n-1][4096]
int upload_model_tables(pcsf_ctx* c, const double* P, const pcsf::Vec prior[2])
{
    const int n = c->n_nodes, nl = c->n_leaves;
    const int ne = static_cast<int>(c->schedule.slot_node.size());
    std::vector<double> frag(static_cast<size_t>(2) * ne * pcsf::FRAG_DOUBLES);
    std::vector<double> leaf(static_cast<size_t>(2) * nl * pcsf::LEAF_TABLE_DOUBLES);
    std::vector<double> pr(2 * 64), aw(2 * 64);
    for (int m = 0; m < 2; ++m) {
        const double* Pm = P + static_cast<size_t>(m) * (n - 1) * 4096;
        for (int s = 0; s < ne; ++s)
            pcsf::to_fragment_order(Pm + static_cast<size_t>(c->schedule.slot_node[s]) * 4096,
                                    frag.data() + (static_cast<size_t>(m) * ne + s) * pcsf::FRAG_DOUBLES);
        for (int l = 0; l < nl; ++l)
            pcsf::to_leaf_table(Pm + static_cast<size_t>(l) * 4096,
                                leaf.data() + (static_cast<size_t>(m) * nl + l) * pcsf::LEAF_TABLE_DOUBLES);
        pcsf::to_lane_order(prior[m].data(), pr.data() + m * 64);
        pcsf::Vec w;
        for (int x = 0; x < 64; ++x) w[x] = std::log(prior[m][x]) * prior[m][x];
        pcsf::to_lane_order(w.data(), aw.data() + m * 64);
    }
    hipError_t e;
    if ((e = c->d_frag.reserve(std::max<size_t>(frag.size(), 1) * sizeof(double))) != hipSuccess) return c->hip_fail(e, "hipMalloc(frag)");
    if ((e = c->d_leaf_tab.reserve(leaf.size() * sizeof(double))) != hipSuccess) return c->hip_fail(e, "hipMalloc(leaf_tab)");
    if ((e = c->d_prior.reserve(pr.size() * sizeof(double))) != hipSuccess) return c->hip_fail(e, "hipMalloc(prior)");
    if ((e = c->d_anc_weight.reserve(aw.size() * sizeof(double))) != hipSuccess) return c->hip_fail(e, "hipMalloc(anc_weight)");
    if ((e = hipMemcpy(c->d_anc_weight.ptr, aw.data(), aw.size() * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess)
        return c->hip_fail(e, "hipMemcpy(anc_weight)");
    if (!frag.empty() && (e = hipMemcpy(c->d_frag.ptr, frag.data(), frag.size() * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess)
        return c->hip_fail(e, "hipMemcpy(frag)");
    if ((e = hipMemcpy(c->d_leaf_tab.ptr, leaf.data(), leaf.size() * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess)
        return c->hip_fail(e, "hipMemcpy(leaf_tab)");
    if ((e = hipMemcpy(c->d_prior.ptr, pr.data(), pr.size() * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess)
        return c->hip_fail(e, "hipMemcpy(prior)");
    return PCSF_OK;
}

int build_cherry_tables(pcsf_ctx* c, const pcsf::Vec prior[2]);

}  // namespace

namespace pcsf {

PruneParams base_prune_params(const pcsf_ctx* c)
{
    PruneParams p{};
    p.program = c->d_program.as<uint32_t>();
    p.op_leaves = c->d_op_meta.ptr ? c->d_op_leaves.as<uint64_t>() : nullptr;
    p.op_meta = c->d_op_meta.ptr ? c->d_op_meta.as<uint32_t>() : nullptr;
    p.gap_tab = c->d_gap_tab.as<double>();
    p.n_ops = static_cast<int>(c->schedule.program.size());
    p.n_edges = static_cast<int>(c->schedule.slot_node.size());
    p.n_leaves = c->n_leaves;
    p.n_passes = 2;
    p.frag = c->d_frag.as<double>();
    p.leaf_tab = c->d_leaf_tab.as<double>();
    p.prior = c->d_prior.as<double>();
    p.anc_weight = c->d_anc_weight.as<double>();
    p.stamps = c->d_stamps.as<unsigned long long>();
    p.stamp_block = c->stamp_block;
    p.stagger_sleep = c->stagger_sleep;
    p.prio_mode = c->prio_mode;
    p.stagger_first_blocks = c->stagger_first_blocks;
    return p;
}

PruneParams tracks_prune_params(const pcsf_ctx* c, int* variant_flags)
{
    PruneParams p = base_prune_params(c);
    if (!c->cherry.on) return p;
    const pcsf_ctx::CherryTables& t = c->cherry;
    p.program = t.d_program.as<uint32_t>();
    p.op_leaves = t.d_op_meta.ptr ? t.d_op_leaves.as<uint64_t>() : nullptr;
    p.op_meta = t.d_op_meta.ptr ? t.d_op_meta.as<uint32_t>() : nullptr;
    p.gap_tab = t.d_gap_tab.as<double>();
    p.n_ops = static_cast<int>(t.schedule.program.size());
    p.n_edges = static_cast<int>(t.schedule.slot_node.size());
    p.frag = t.d_frag.as<double>();
    p.leaf_tab = t.d_leaf_tab.as<double>();
    p.pair_leaves = t.d_pair_leaves.as<uint16_t>();
    p.trip_info = t.d_trip_info.as<uint32_t>();
    p.n_pairs = static_cast<int>(t.pairs.size());
    p.n_free_pairs = t.n_free_pairs;
    p.n_triples = static_cast<int>(t.triples.size());
    p.n_free_triples = t.n_free_triples;
    p.n_quads = t.n_quads;
    p.quad_info = t.d_quad_info.as<uint32_t>();
    p.leaf_rows = t.leaf_rows;
    if (variant_flags) *variant_flags |= 256;
    return p;
}

int launch_fixed(pcsf_ctx* c, const uint8_t* d_codons, size_t n_sites, size_t leaf_stride, double* d_out, double* d_lnz,
                 double* d_anc, hipStream_t stream, const FusedFrontEnd* fused)
{
    int cherry_flag = 0;
    PruneParams p = tracks_prune_params(c, &cherry_flag);
    const int stack_depth = cherry_flag ? c->cherry.schedule.stack_depth : c->schedule.stack_depth;
    p.codons = d_codons;
    p.n_sites = n_sites;
    p.leaf_stride = leaf_stride;
    p.out_score = d_out;
    p.out_lnz = d_lnz;
    p.out_anc = d_anc;
    int variant = c->variant;
    if (fused) {
        p.nucl = fused->nucl;
        p.nucl_stride = fused->nucl_stride;
        p.segments = fused->segments;
        p.wg_first_seg = fused->wg_first_seg;
        p.first_bad = fused->first_bad;
        p.dump_codons = fused->dump_codons;
        variant = 32;
    }
    if (cherry_flag && variant != 0 && variant != 2 && variant != 32) cherry_flag = 0, p = base_prune_params(c);   // (the 8-wave form: the full program)
    if (stack_depth > pcsf::MAX_STACK_DEPTH) variant = (variant == 32 ? 32 : 0) | 64;  // parked tiles in scratch memory
    // All-gap subtrees of a work-group's 64 columns cost nothing (kernels.hip SKIP).  Default: on where the columns come from
    // alignments (the six-frame entry points: species missing from a block are missing for hundreds of columns; 1.7x on a MAF with
    // clade-level drop-out, neutral on the bench's), off for the raw codon-matrix entry points, whose columns have no such structure
    // and would only pay the compaction (1.3 %).  PCSF_SKIP_ALLGAP=1 / 0 forces it on / off everywhere; scores are bit-identical.
    const bool skip = c->skip_allgap_forced ? c->skip_allgap : (fused != nullptr);
    if (skip && p.op_meta && p.gap_tab && p.n_ops <= 128 && p.n_leaves <= 128 && (variant & ~(32 | 64)) == 0) variant |= 128;
    variant |= cherry_flag;
    hipError_t e;
    if (c->timing && (e = hipEventRecord(c->ev0, stream)) != hipSuccess) return c->hip_fail(e, "hipEventRecord");
    if ((e = launch_prune_fixed(p, variant, stream, c->extra_lds)) != hipSuccess) return c->hip_fail(e, "launch prune_fixed");
    if (c->timing) {
        if ((e = hipEventRecord(c->ev1, stream)) != hipSuccess) return c->hip_fail(e, "hipEventRecord");
        if ((e = hipEventSynchronize(c->ev1)) != hipSuccess) return c->hip_fail(e, "hipEventSynchronize");
        float ms = 0.f;
        if ((e = hipEventElapsedTime(&ms, c->ev0, c->ev1)) != hipSuccess) return c->hip_fail(e, "hipEventElapsedTime");
        c->last_kernel_ms = ms;
    }
    return PCSF_OK;
}

}  // namespace pcsf

namespace {

// Subtree tables of a context (ctx.hpp CherryTables): the reduced tree's program and facts, its fragment images, the table sets --
// leaves' rows from the host, the cherries' pair rows and the triples' rows by combine_table_kernel -- and the reduced program's all-gap
// messages, recorded like the full program's.  Leaves the context without them (tracks mode then runs the full program) when the tree
// has no cherry below the root.  PCSF_TABLE_BYTES bounds the rows of both models (default 8 GiB: every tree the reference ships gets its
// triples -- 58mammals 2.6 GB, 100vertebrates 4.4 GB); a tree whose triples do not fit keeps its cherries, one whose cherries do not fit
// runs the full program.  PCSF_TABLE_LEAVES=4 with a budget to match (9.1 GB per node and model) adds the nodes over two cherries / over a triple
// and a leaf.
int build_cherry_tables(pcsf_ctx* c, const pcsf::Vec prior[2])
{
    (void)prior;
    const int n = c->n_nodes, nl = c->n_leaves;
    size_t budget = size_t(8) << 30;
    if (const char* v = std::getenv("PCSF_TABLE_BYTES")) budget = static_cast<size_t>(std::strtoull(v, nullptr, 10));
    int max_leaves = 3;
    if (const char* v = std::getenv("PCSF_TABLE_LEAVES")) max_leaves = std::atoi(v);   // 2: cherries only (A/B)
    pcsf::CherryReduction red;
    size_t leaf_rows = 0;
    for (; max_leaves >= 2; --max_leaves) {
        try {
            red = pcsf::reduce_cherries(pcsf::TreeView{n, c->child1.data(), c->child2.data()}, max_leaves);
        } catch (const std::exception&) {
            return PCSF_OK;
        }
        if (!red.any()) return PCSF_OK;
        leaf_rows = static_cast<size_t>(pcsf::table_set_rows(nl, static_cast<int>(red.pairs.size()), static_cast<int>(red.triples.size()), static_cast<int>(red.quads.size())));
        // LDS of a work-group of the table form at its largest (fused front end, all-gap skipping; kernels.hip launch_impl has the same sum):
        // the codes it stages must neither pass the CU's 160 KiB nor cost the second work-group of a CU where the full program has one
        const size_t lds_base = ((2 * 32768 + ((c->schedule.program.size() + 3) & ~size_t(3)) * 4 + static_cast<size_t>(nl) * 64 + 15) & ~size_t(15)) + 256 + 64 * 9 + 16;
        const size_t lds_tab = ((lds_base + 15) & ~size_t(15)) + ((static_cast<size_t>(red.n_free_pairs) * 128 + 15) & ~size_t(15)) +
                               (static_cast<size_t>(red.n_free_triples) + red.quads.size()) * 256;
        const size_t two_per_cu = 160 * 1024 / 2 - 1560;   // (the skipping variants' static LDS beside the dynamic array)
        const bool lds_ok = lds_tab <= 158 * 1024 && (lds_tab <= two_per_cu || lds_base > two_per_cu);
        if (lds_ok && leaf_rows * 64 * sizeof(double) * 2 <= budget && leaf_rows < 0xffffffffull) {
            // the rows themselves: when the device has no room for them (other contexts, other processes) the next smaller form is tried
            const hipError_t ea = c->cherry.d_leaf_tab.reserve_exact(leaf_rows * 64 * sizeof(double) * 2);
            if (ea == hipSuccess) break;
            (void)hipGetLastError();
        }
        if (red.triples.empty() && red.quads.empty()) return PCSF_OK;   // not even the cherries fit
    }
    if (max_leaves < 2) return PCSF_OK;
    const size_t np = red.pairs.size(), nt = red.triples.size(), nq = red.quads.size();
    pcsf_ctx::CherryTables& t = c->cherry;
    try {
        t.schedule = pcsf::build_reduced_schedule(red);
    } catch (const std::exception&) {
        return PCSF_OK;
    }
    t.pairs = red.pairs;
    t.triples = red.triples;
    t.n_free_pairs = red.n_free_pairs;
    t.n_free_triples = red.n_free_triples;
    t.n_quads = static_cast<int>(nq);
    t.leaf_rows = static_cast<uint32_t>(leaf_rows);
    hipError_t e;
    const auto& prog = t.schedule.program;
    const int ne = static_cast<int>(t.schedule.slot_node.size());
    if ((e = t.d_program.reserve(prog.size() * sizeof(uint32_t))) != hipSuccess ||
        (e = hipMemcpy(t.d_program.ptr, prog.data(), prog.size() * sizeof(uint32_t), hipMemcpyHostToDevice)) != hipSuccess)
        return c->hip_fail(e, "pcsf_create (subtree tables: program)");
    const pcsf::ProgramFacts facts = pcsf::program_facts(t.schedule, &red, nl);
    if (facts.ok) {
        if ((e = t.d_op_leaves.reserve(facts.leaves.size() * sizeof(uint64_t))) != hipSuccess ||
            (e = hipMemcpy(t.d_op_leaves.ptr, facts.leaves.data(), facts.leaves.size() * sizeof(uint64_t), hipMemcpyHostToDevice)) != hipSuccess ||
            (e = t.d_op_meta.reserve(facts.meta.size() * sizeof(uint32_t))) != hipSuccess ||
            (e = hipMemcpy(t.d_op_meta.ptr, facts.meta.data(), facts.meta.size() * sizeof(uint32_t), hipMemcpyHostToDevice)) != hipSuccess)
            return c->hip_fail(e, "pcsf_create (subtree tables: program facts)");
    }
    // fragment images: the reduced program's slots, then (scratch) the table nodes' -- cherries, then triples -- for the table kernel
    const size_t n_tab = np + nt + nq;
    std::vector<double> frag(static_cast<size_t>(2) * std::max(ne, 1) * pcsf::FRAG_DOUBLES), cfrag(2 * n_tab * pcsf::FRAG_DOUBLES);
    std::vector<double> leaf(static_cast<size_t>(nl) * pcsf::LEAF_TABLE_DOUBLES);
    std::vector<uint16_t> pl(2 * np);
    for (size_t p = 0; p < np; ++p) {
        pl[2 * p] = static_cast<uint16_t>(red.pairs[p][0]);
        pl[2 * p + 1] = static_cast<uint16_t>(red.pairs[p][1]);
    }
    const uint32_t pair_row0 = static_cast<uint32_t>(nl) * 65u, trip_row0 = pair_row0 + static_cast<uint32_t>(np) * pcsf::PAIR_CODES;
    std::vector<uint32_t> ti(3 * std::max<size_t>(nt, 1), 0);
    std::vector<pcsf::CombineTable> pair_tabs(np), trip_tabs(nt), quad_tabs(nq);
    const uint32_t quad_row0 = trip_row0 + static_cast<uint32_t>(nt) * pcsf::TRIPLE_CODES;
    std::vector<uint32_t> qi(3 * std::max<size_t>(nq, 1), 0);
    for (size_t k = 0; k < nq; ++k) {
        const pcsf::CherryReduction::Quad& q = red.quads[k];
        const uint32_t base = quad_row0 + static_cast<uint32_t>(k * static_cast<size_t>(pcsf::QUAD_CODES));
        qi[3 * k] = static_cast<uint32_t>(q.leaves[0]) | (static_cast<uint32_t>(q.leaves[1]) << 16);
        qi[3 * k + 1] = static_cast<uint32_t>(q.leaves[2]) | (static_cast<uint32_t>(q.leaves[3]) << 16);
        qi[3 * k + 2] = base;
        if (q.x_triple >= 0)
            quad_tabs[k] = pcsf::CombineTable{trip_row0 + static_cast<uint32_t>(q.x_triple) * pcsf::TRIPLE_CODES, static_cast<uint32_t>(pcsf::TRIPLE_CODES), static_cast<uint32_t>(q.y) * 65u, 65u, base,
                                              static_cast<uint32_t>(np + nt + k)};
        else
            quad_tabs[k] = pcsf::CombineTable{pair_row0 + static_cast<uint32_t>(q.x_pair) * pcsf::PAIR_CODES, pcsf::PAIR_CODES, pair_row0 + static_cast<uint32_t>(q.y) * pcsf::PAIR_CODES, pcsf::PAIR_CODES,
                                              base, static_cast<uint32_t>(np + nt + k)};
    }
    for (size_t p = 0; p < np; ++p)
        pair_tabs[p] = pcsf::CombineTable{static_cast<uint32_t>(red.pairs[p][0]) * 65u, 65u, static_cast<uint32_t>(red.pairs[p][1]) * 65u, 65u,
                                          pair_row0 + static_cast<uint32_t>(p) * pcsf::PAIR_CODES, static_cast<uint32_t>(p)};
    for (size_t k = 0; k < nt; ++k) {
        const uint32_t p = static_cast<uint32_t>(red.triples[k][0]), leaf_c = static_cast<uint32_t>(red.triples[k][1]);
        const uint32_t base = trip_row0 + static_cast<uint32_t>(k) * pcsf::TRIPLE_CODES;
        ti[3 * k] = static_cast<uint32_t>(red.pairs[p][0]) | (static_cast<uint32_t>(red.pairs[p][1]) << 16);
        ti[3 * k + 1] = leaf_c;
        ti[3 * k + 2] = base;
        trip_tabs[k] = pcsf::CombineTable{pair_row0 + p * pcsf::PAIR_CODES, pcsf::PAIR_CODES, leaf_c * 65u, 65u, base, static_cast<uint32_t>(np + k)};
    }
    pcsf::DeviceBuffer d_cfrag, d_tabs;
    if ((e = t.d_frag.reserve(frag.size() * sizeof(double))) != hipSuccess || (e = d_cfrag.reserve(cfrag.size() * sizeof(double))) != hipSuccess ||
        (e = t.d_pair_leaves.reserve(pl.size() * sizeof(uint16_t))) != hipSuccess ||
        (e = t.d_trip_info.reserve(ti.size() * sizeof(uint32_t))) != hipSuccess || (e = t.d_quad_info.reserve(qi.size() * sizeof(uint32_t))) != hipSuccess ||
        (e = d_tabs.reserve(n_tab * sizeof(pcsf::CombineTable))) != hipSuccess)
        return c->hip_fail(e, "hipMalloc(subtree tables)");
    for (int m = 0; m < 2; ++m) {
        const double* Pm = c->host_P.data() + static_cast<size_t>(m) * (n - 1) * 4096;
        for (int s = 0; s < ne; ++s)
            pcsf::to_fragment_order(Pm + static_cast<size_t>(t.schedule.slot_node[s]) * 4096, frag.data() + (static_cast<size_t>(m) * ne + s) * pcsf::FRAG_DOUBLES);
        for (size_t p = 0; p < np; ++p)
            pcsf::to_fragment_order(Pm + static_cast<size_t>(red.pairs[p][2]) * 4096, cfrag.data() + (static_cast<size_t>(m) * n_tab + p) * pcsf::FRAG_DOUBLES);
        for (size_t k = 0; k < nt; ++k)
            pcsf::to_fragment_order(Pm + static_cast<size_t>(red.triples[k][2]) * 4096, cfrag.data() + (static_cast<size_t>(m) * n_tab + np + k) * pcsf::FRAG_DOUBLES);
        for (size_t k = 0; k < nq; ++k)
            pcsf::to_fragment_order(Pm + static_cast<size_t>(red.quads[k].node) * 4096, cfrag.data() + (static_cast<size_t>(m) * n_tab + np + nt + k) * pcsf::FRAG_DOUBLES);
        for (int l = 0; l < nl; ++l) pcsf::to_leaf_table(Pm + static_cast<size_t>(l) * 4096, leaf.data() + static_cast<size_t>(l) * pcsf::LEAF_TABLE_DOUBLES);
        // (a synchronous copy: `leaf` is filled again for the second model)
        if ((e = hipMemcpy(t.d_leaf_tab.as<double>() + static_cast<size_t>(m) * leaf_rows * 64, leaf.data(), leaf.size() * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess)
            return c->hip_fail(e, "hipMemcpy(subtree tables: leaf rows)");
    }
    if ((e = hipMemcpy(t.d_frag.ptr, frag.data(), frag.size() * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(d_cfrag.ptr, cfrag.data(), cfrag.size() * sizeof(double), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(t.d_pair_leaves.ptr, pl.data(), pl.size() * sizeof(uint16_t), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(t.d_trip_info.ptr, ti.data(), ti.size() * sizeof(uint32_t), hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(t.d_quad_info.ptr, qi.data(), qi.size() * sizeof(uint32_t), hipMemcpyHostToDevice)) != hipSuccess ||
        (nq && (e = hipMemcpy(d_tabs.as<pcsf::CombineTable>() + np + nt, quad_tabs.data(), nq * sizeof(pcsf::CombineTable), hipMemcpyHostToDevice)) != hipSuccess) ||
        (e = hipMemcpy(d_tabs.ptr, pair_tabs.data(), np * sizeof(pcsf::CombineTable), hipMemcpyHostToDevice)) != hipSuccess ||
        (nt && (e = hipMemcpy(d_tabs.as<pcsf::CombineTable>() + np, trip_tabs.data(), nt * sizeof(pcsf::CombineTable), hipMemcpyHostToDevice)) != hipSuccess))
        return c->hip_fail(e, "hipMemcpy(subtree tables)");
    // the cherries' rows, then the triples' (which read them)
    if ((e = pcsf::launch_combine_tables(d_tabs.as<pcsf::CombineTable>(), static_cast<int>(np), pcsf::PAIR_CODES, d_cfrag.as<double>(), static_cast<int>(n_tab), t.d_leaf_tab.as<double>(),
                                         t.leaf_rows, c->stream)) != hipSuccess ||
        (e = pcsf::launch_combine_tables(d_tabs.as<pcsf::CombineTable>() + np, static_cast<int>(nt), pcsf::TRIPLE_CODES, d_cfrag.as<double>(), static_cast<int>(n_tab),
                                         t.d_leaf_tab.as<double>(), t.leaf_rows, c->stream)) != hipSuccess ||
        (e = pcsf::launch_combine_tables(d_tabs.as<pcsf::CombineTable>() + np + nt, static_cast<int>(nq), static_cast<uint32_t>(pcsf::QUAD_CODES), d_cfrag.as<double>(), static_cast<int>(n_tab),
                                         t.d_leaf_tab.as<double>(), t.leaf_rows, c->stream)) != hipSuccess ||
        (e = hipStreamSynchronize(c->stream)) != hipSuccess)
        return c->hip_fail(e, "combine_table_kernel");
    t.on = true;
    if (c->skip_allgap && t.d_op_meta.ptr) {   // the reduced program's all-gap messages, by the reduced program itself (like the full program's above)
        const size_t doubles = 2 * (static_cast<size_t>(ne) + 1) * 64;
        pcsf::DeviceBuffer d_gaps;
        if ((e = t.d_gap_tab.reserve(doubles * sizeof(double))) != hipSuccess || (e = d_gaps.reserve(static_cast<size_t>(nl) * 64)) != hipSuccess ||
            (e = hipMemsetAsync(t.d_gap_tab.ptr, 0, doubles * sizeof(double), c->stream)) != hipSuccess ||
            (e = hipMemsetAsync(d_gaps.ptr, 64, static_cast<size_t>(nl) * 64, c->stream)) != hipSuccess) {
            t.on = false;
            return c->hip_fail(e, "pcsf_create (subtree tables: all-gap messages)");
        }
        int flag = 0;
        pcsf::PruneParams gp = pcsf::tracks_prune_params(c, &flag);
        gp.gap_tab = nullptr;
        gp.gap_dump = t.d_gap_tab.as<double>();
        gp.stamps = nullptr;
        gp.codons = d_gaps.as<uint8_t>();
        gp.n_sites = 64;
        gp.leaf_stride = 64;
        if ((e = pcsf::launch_prune_fixed(gp, 256 | (t.schedule.stack_depth > pcsf::MAX_STACK_DEPTH ? 66 : 2), c->stream, 0)) != hipSuccess ||
            (e = hipStreamSynchronize(c->stream)) != hipSuccess) {
            t.on = false;
            return c->hip_fail(e, "pcsf_create (subtree tables: all-gap messages)");
        }
    }
    return PCSF_OK;
}

}  // namespace

namespace pcsf {
// Branch-length scores of the columns in `spans` of the DEVICE-resident rows d_nucl[n_leaves][d_stride]: enqueues
// the kernel on the context's stream (no synchronisation) ...
static int bls_enqueue(pcsf_ctx* c, const unsigned char* d_nucl, size_t d_stride, size_t n_cols, const std::vector<BlsSpan>& spans)
{
    hipError_t e;
    if ((e = c->d_bls_out.reserve(n_cols * sizeof(double))) != hipSuccess) return c->hip_fail(e, "hipMalloc(bls out)");
    if ((e = c->d_bls_spans.reserve(spans.size() * sizeof(BlsSpan))) != hipSuccess) return c->hip_fail(e, "hipMalloc(bls spans)");
    // columns outside every span are reported as 0
    if ((e = hipMemsetAsync(c->d_bls_out.ptr, 0, n_cols * sizeof(double), c->stream)) != hipSuccess) return c->hip_fail(e, "hipMemsetAsync(bls out)");
    if ((e = hipMemsetAsync(c->d_bls_flag.ptr, 0xff, sizeof(unsigned long long), c->stream)) != hipSuccess) return c->hip_fail(e, "hipMemsetAsync(bls flag)");
    if ((e = hipMemcpyAsync(c->d_bls_spans.ptr, spans.data(), spans.size() * sizeof(BlsSpan), hipMemcpyHostToDevice, c->stream)) != hipSuccess)
        return c->hip_fail(e, "hipMemcpyAsync(bls spans)");
    for (size_t s0 = 0; s0 < spans.size(); s0 += 65535) {
        const uint32_t ns = static_cast<uint32_t>(std::min<size_t>(65535, spans.size() - s0));
        uint64_t max_len = 0;
        for (uint32_t k = 0; k < ns; ++k) max_len = std::max<uint64_t>(max_len, spans[s0 + k].len);
        if ((e = launch_bls(c->bls_tree, c->d_bls_spans.as<BlsSpan>() + s0, ns, max_len, d_nucl, d_stride, c->d_bls_out.as<double>(),
                            c->d_bls_flag.as<unsigned long long>(), c->stream)) != hipSuccess)
            return c->hip_fail(e, "launch bls");
    }
    return PCSF_OK;
}

// ... and the copies of columns [0, n_cols) and of the character check to the host, also without synchronising.
// Kept apart from the launch so that a caller with more kernels to run can queue all its downloads behind them
// (a download between two kernels idles the compute queue while the copy engine works).
static int bls_download(pcsf_ctx* c, size_t n_cols, double* host_out)
{
    hipError_t e;
    if ((e = hipMemcpyAsync(host_out, c->d_bls_out.ptr, n_cols * sizeof(double), hipMemcpyDeviceToHost, c->stream)) != hipSuccess)
        return c->hip_fail(e, "hipMemcpyAsync(bls out)");
    // into page-locked memory: a copy to pageable memory would block the host here until the kernel has run
    *c->bls_first_bad = BLS_NO_BAD;
    if ((e = hipMemcpyAsync(c->bls_first_bad, c->d_bls_flag.ptr, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream)) != hipSuccess)
        return c->hip_fail(e, "hipMemcpyAsync(bls flag)");
    return PCSF_OK;
}

// after the stream has been synchronised: was a character outside ACGTNacgtn-. seen?  (the reference exits on
// one, src/translation.hpp:29-88); host_nucl/host_stride locate the character for the message
static int bls_check(pcsf_ctx* c, const char* host_nucl, size_t host_stride, size_t d_stride)
{
    const unsigned long long bad = *c->bls_first_bad;  // copied out behind the kernel by bls_enqueue
    if (bad == BLS_NO_BAD) return PCSF_OK;
    const size_t row = bad / d_stride, col = bad % d_stride;
    return c->fail(PCSF_ERR_INVALID_ARG, std::string("unexpected character '") + host_nucl[row * host_stride + col] + "' in alignment");
}

// host rows in, host scores out (uploads the rows, synchronises)
static int bls_columns(pcsf_ctx* c, const char* nucleotides, size_t row_stride, size_t n_cols, const std::vector<BlsSpan>& spans, double* out)
{
    hipError_t e;
    const size_t nl = static_cast<size_t>(c->n_leaves);
    if ((e = c->d_nucl.reserve(nl * n_cols)) != hipSuccess) return c->hip_fail(e, "hipMalloc(nucleotides)");
    if ((e = upload_rows(c->d_nucl.ptr, nucleotides, row_stride, n_cols, nl, c->stream)) != hipSuccess) return c->hip_fail(e, "upload(nucleotides)");
    int rc = bls_enqueue(c, c->d_nucl.as<unsigned char>(), n_cols, n_cols, spans);
    if (rc == PCSF_OK) rc = bls_download(c, n_cols, out);
    if (rc != PCSF_OK) return rc;
    if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return c->hip_fail(e, "hipStreamSynchronize");
    return bls_check(c, nucleotides, row_stride, n_cols);
}
}  // namespace pcsf

extern "C" {

const char* pcsf_version(void) { return "phylocsf-amd 0.1.0 (gfx950)"; }

void* pcsf_host_alloc(size_t bytes)
{
    void* p = nullptr;
    // portable: page-locked for every device, so one pool can feed the contexts of several GPUs
    if (bytes == 0 || hipHostMalloc(&p, bytes, hipHostMallocPortable) != hipSuccess) return nullptr;
    return p;
}

void* pcsf_host_alloc_on(pcsf_ctx* c, size_t bytes)
{
    // page-locked memory is allocated through the CURRENT device's context: select the context's GPU first, so that a
    // run on device N does not create a primary context on device 0 (and works when device 0 is masked or exclusive)
    if (!c || hipSetDevice(c->device) != hipSuccess) return nullptr;
    return pcsf_host_alloc(bytes);
}

void pcsf_host_free(void* p)
{
    if (p) (void)hipHostFree(p);
}

// host-only helper: drives the lock-step optimiser's state machine (optimizer.hpp) with a caller-supplied
// objective; records the evaluation points.  Lets the tests check that it takes exactly the decisions
// of the reference's fit_find_init + GSL Brent without a GPU in the loop.
// the restart points a search over [lo, hi] draws from a std::mt19937 in state `gen` (optimizer.hpp DrawTable)
static std::vector<double> debug_draw_table(std::mt19937 gen, double lo, double hi, size_t n)
{
    std::vector<double> t(n);
    const double width = std::log(hi) - std::log(lo);
    for (double& x : t) {
        std::uniform_real_distribution<> dis(0.0, width);
        const double r = dis(gen);
        x = std::exp(std::log(lo) + r);
    }
    return t;
}

int pcsf_debug_maxlik(double (*objective)(double x, void* user), void* user, double init, double lo, double hi, uint32_t seed,
                      int skip_draws, double* xs_out, int cap, int* n_evals, double* last_lpr)
{
    if (!objective || !n_evals || !last_lpr) return PCSF_ERR_INVALID_ARG;
    std::mt19937 gen(seed);
    for (int i = 0; i < skip_draws; ++i) (void)gen();
    const std::vector<double> table = debug_draw_table(gen, lo, hi, 260);
    const pcsf::DrawTable dt{table.data(), static_cast<uint32_t>(table.size())};
    uint32_t next = 0;
    pcsf::MaxLik1D opt;
    opt.start(init, lo, hi);
    int n = 0;
    while (!opt.done()) {
        const double x = opt.request();
        if (xs_out && n < cap) xs_out[n] = x;
        ++n;
        opt.supply(objective(x, user), 0.0, dt, next);
    }
    *n_evals = n;
    *last_lpr = opt.last_lpr();
    return PCSF_OK;
}

// The same state machine ON THE DEVICE (chain.hip maxlik_replay_kernel: the code advance_blocks runs per block), fed with a recorded
// sequence of objective values: equal histories give equal requests, so a search that decides like the one that produced `values`
// asks for exactly its points.  *n_requests = -1 when it wanted more values than were recorded.
int pcsf_debug_maxlik_device(pcsf_ctx* c, double init, double lo, double hi, uint32_t seed, int skip_draws, const double* values, int n_values,
                             double* xs_out, int cap, int* n_requests, double* last_lpr)
{
    if (!c) return PCSF_ERR_INVALID_ARG;
    if (!values || n_values <= 0 || !xs_out || cap <= 0 || !n_requests || !last_lpr) return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_debug_maxlik_device: bad argument");
    hipError_t e;
    if ((e = hipSetDevice(c->device)) != hipSuccess) return c->hip_fail(e, "hipSetDevice");
    std::mt19937 gen(seed);
    for (int i = 0; i < skip_draws; ++i) (void)gen();
    const std::vector<double> table = debug_draw_table(gen, lo, hi, 260);
    pcsf::DeviceBuffer d_table, d_values, d_xs, d_n, d_last;
    if ((e = d_table.reserve(table.size() * 8)) != hipSuccess || (e = d_values.reserve(size_t(n_values) * 8)) != hipSuccess ||
        (e = d_xs.reserve(size_t(cap) * 8)) != hipSuccess || (e = d_n.reserve(sizeof(int))) != hipSuccess || (e = d_last.reserve(16)) != hipSuccess)
        return c->hip_fail(e, "hipMalloc");
    double last[2] = {0, 0};
    if ((e = hipMemcpyAsync(d_table.ptr, table.data(), table.size() * 8, hipMemcpyHostToDevice, c->stream)) != hipSuccess ||
        (e = hipMemcpyAsync(d_values.ptr, values, size_t(n_values) * 8, hipMemcpyHostToDevice, c->stream)) != hipSuccess ||
        (e = hipMemsetAsync(d_xs.ptr, 0, size_t(cap) * 8, c->stream)) != hipSuccess ||
        (e = pcsf::launch_maxlik_replay(init, lo, hi, d_table.as<double>(), static_cast<uint32_t>(table.size()), d_values.as<double>(), n_values, d_xs.as<double>(), cap,
                                        d_n.as<int>(), d_last.as<double>(), c->stream)) != hipSuccess ||
        (e = hipMemcpyAsync(xs_out, d_xs.ptr, size_t(cap) * 8, hipMemcpyDeviceToHost, c->stream)) != hipSuccess ||
        (e = hipMemcpyAsync(n_requests, d_n.ptr, sizeof(int), hipMemcpyDeviceToHost, c->stream)) != hipSuccess ||
        (e = hipMemcpyAsync(last, d_last.ptr, 16, hipMemcpyDeviceToHost, c->stream)) != hipSuccess || (e = hipStreamSynchronize(c->stream)) != hipSuccess)
        return c->hip_fail(e, "pcsf_debug_maxlik_device");
    *last_lpr = last[0];
    return PCSF_OK;
}

// wig.hip's number formatter on n values: text[32 * i ...] holds value i's digits, lens[i] their number (0xffffffff: a value the device
// leaves to the host).  decimals: 3 (raw tracks) or 4 (power track).
int pcsf_debug_wig_format(pcsf_ctx* c, const float* values, size_t n, int decimals, char* text, uint32_t* lens)
{
    if (!c) return PCSF_ERR_INVALID_ARG;
    if (!values || !text || !lens || (decimals != 3 && decimals != 4) || n >= 0xffffffffull) return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_debug_wig_format: bad argument");
    if (n == 0) return PCSF_OK;
    hipError_t e;
    if ((e = hipSetDevice(c->device)) != hipSuccess) return c->hip_fail(e, "hipSetDevice");
    pcsf::DeviceBuffer d_v, d_text, d_lens;
    if ((e = d_v.reserve(n * 4)) != hipSuccess || (e = d_text.reserve(n * 32)) != hipSuccess || (e = d_lens.reserve(n * 4)) != hipSuccess) return c->hip_fail(e, "hipMalloc");
    if ((e = hipMemcpyAsync(d_v.ptr, values, n * 4, hipMemcpyHostToDevice, c->stream)) != hipSuccess ||
        (e = hipMemsetAsync(d_text.ptr, 0, n * 32, c->stream)) != hipSuccess ||
        (e = pcsf::launch_wig_format_debug(d_v.as<float>(), static_cast<uint32_t>(n), decimals, d_text.as<char>(), d_lens.as<uint32_t>(), c->stream)) != hipSuccess ||
        (e = hipMemcpyAsync(text, d_text.ptr, n * 32, hipMemcpyDeviceToHost, c->stream)) != hipSuccess ||
        (e = hipMemcpyAsync(lens, d_lens.ptr, n * 4, hipMemcpyDeviceToHost, c->stream)) != hipSuccess || (e = hipStreamSynchronize(c->stream)) != hipSuccess)
        return c->hip_fail(e, "pcsf_debug_wig_format");
    return PCSF_OK;
}

// host-only helper (no device needed): the pruning program for a tree, for tests and tooling
int pcsf_build_schedule(int32_t n_nodes, const int16_t* child1, const int16_t* child2, uint32_t* program, int32_t program_cap,
                        int32_t* slot_node, int32_t* n_ops, int32_t* n_slots, int32_t* stack_depth)
{
    if (!child1 || !child2 || !program || !slot_node || !n_ops || !n_slots || !stack_depth) return PCSF_ERR_INVALID_ARG;
    try {
        const pcsf::Schedule s = pcsf::build_schedule(pcsf::TreeView{n_nodes, child1, child2});
        if (static_cast<int32_t>(s.program.size()) > program_cap) return PCSF_ERR_INVALID_ARG;
        std::copy(s.program.begin(), s.program.end(), program);
        std::copy(s.slot_node.begin(), s.slot_node.end(), slot_node);
        *n_ops = static_cast<int32_t>(s.program.size());
        *n_slots = static_cast<int32_t>(s.slot_node.size());
        *stack_depth = s.stack_depth;
    } catch (const std::exception& ex) {
        g_create_error = std::string("pcsf_build_schedule: ") + ex.what();
        return PCSF_ERR_INVALID_ARG;
    }
    return PCSF_OK;
}

int pcsf_tracks_program_info(const pcsf_ctx* c, int64_t* out)
{
    if (!c || !out) return PCSF_ERR_INVALID_ARG;
    out[0] = static_cast<int64_t>(c->schedule.slot_node.size());
    out[1] = c->cherry.on ? static_cast<int64_t>(c->cherry.schedule.slot_node.size()) : out[0];
    out[2] = c->cherry.on ? static_cast<int64_t>(c->cherry.pairs.size()) : 0;
    out[3] = c->cherry.on ? static_cast<int64_t>(c->cherry.leaf_rows) : static_cast<int64_t>(c->n_leaves) * 65;
    out[4] = c->cherry.on ? static_cast<int64_t>(c->cherry.triples.size()) : 0;
    out[5] = c->cherry.on ? static_cast<int64_t>(c->cherry.n_quads) : 0;
    return PCSF_OK;
}

int pcsf_debug_cherry_schedule(int32_t n_nodes, const int16_t* child1, const int16_t* child2, uint32_t* program, int32_t program_cap, int32_t* slot_node, int32_t* n_ops,
                               int32_t* n_slots, int32_t* stack_depth, int32_t* pairs, int32_t* n_pairs, int32_t* triples, int32_t* n_triples)
{
    if (!child1 || !child2 || !program || !slot_node || !n_ops || !n_slots || !stack_depth || !pairs || !n_pairs) return PCSF_ERR_INVALID_ARG;
    try {
        const pcsf::CherryReduction red = pcsf::reduce_cherries(pcsf::TreeView{n_nodes, child1, child2}, triples ? 3 : 2);
        *n_pairs = static_cast<int32_t>(red.pairs.size());
        if (n_triples) *n_triples = static_cast<int32_t>(red.triples.size());
        if (!red.any()) return PCSF_OK;
        const pcsf::Schedule s = pcsf::build_reduced_schedule(red);
        if (static_cast<int32_t>(s.program.size()) > program_cap) return PCSF_ERR_INVALID_ARG;
        std::copy(s.program.begin(), s.program.end(), program);
        std::copy(s.slot_node.begin(), s.slot_node.end(), slot_node);
        *n_ops = static_cast<int32_t>(s.program.size());
        *n_slots = static_cast<int32_t>(s.slot_node.size());
        *stack_depth = s.stack_depth;
        for (size_t p = 0; p < red.pairs.size(); ++p)
            for (int k = 0; k < 3; ++k) pairs[3 * p + k] = red.pairs[p][k];
        if (triples)
            for (size_t p = 0; p < red.triples.size(); ++p)
                for (int k = 0; k < 3; ++k) triples[3 * p + k] = red.triples[p][k];
    } catch (const std::exception& ex) {
        g_create_error = std::string("pcsf_debug_cherry_schedule: ") + ex.what();
        return PCSF_ERR_INVALID_ARG;
    }
    return PCSF_OK;
}

int pcsf_debug_table_schedule(int32_t n_nodes, const int16_t* child1, const int16_t* child2, int32_t max_leaves, uint32_t* program, int32_t program_cap, int32_t* slot_node,
                              int32_t* n_ops, int32_t* n_slots, int32_t* stack_depth, int32_t* operands, int32_t* n_operands)
{
    if (!child1 || !child2 || !program || !slot_node || !n_ops || !n_slots || !stack_depth || !operands || !n_operands || max_leaves < 2 || max_leaves > 4) return PCSF_ERR_INVALID_ARG;
    try {
        const pcsf::CherryReduction red = pcsf::reduce_cherries(pcsf::TreeView{n_nodes, child1, child2}, max_leaves);
        *n_operands = 0;
        if (!red.any()) return PCSF_OK;
        const pcsf::Schedule s = pcsf::build_reduced_schedule(red);
        if (static_cast<int32_t>(s.program.size()) > program_cap) return PCSF_ERR_INVALID_ARG;
        std::copy(s.program.begin(), s.program.end(), program);
        std::copy(s.slot_node.begin(), s.slot_node.end(), slot_node);
        *n_ops = static_cast<int32_t>(s.program.size());
        *n_slots = static_cast<int32_t>(s.slot_node.size());
        *stack_depth = s.stack_depth;
        int32_t k = 0;
        for (size_t r = 0; r < red.operand.size(); ++r)
            if (red.operand[r] >= pcsf::PAIR_OPERAND_BASE) {
                operands[2 * k] = red.operand[r];
                operands[2 * k + 1] = red.node[r];
                ++k;
            }
        *n_operands = k;
    } catch (const std::exception& ex) {
        g_create_error = std::string("pcsf_debug_table_schedule: ") + ex.what();
        return PCSF_ERR_INVALID_ARG;
    }
    return PCSF_OK;
}

int pcsf_debug_program_facts(int32_t n_nodes, const int16_t* child1, const int16_t* child2, uint64_t* leaves, uint32_t* meta, int32_t cap_ops)
{
    if (!child1 || !child2 || !leaves || !meta) return PCSF_ERR_INVALID_ARG;
    try {
        const pcsf::Schedule s = pcsf::build_schedule(pcsf::TreeView{n_nodes, child1, child2});
        const pcsf::ProgramFacts f = pcsf::program_facts(s);
        if (!f.ok) return PCSF_ERR_UNSUPPORTED;
        if (static_cast<int32_t>(f.meta.size()) > cap_ops) return PCSF_ERR_INVALID_ARG;
        std::copy(f.leaves.begin(), f.leaves.end(), leaves);
        std::copy(f.meta.begin(), f.meta.end(), meta);
    } catch (const std::exception& ex) {
        g_create_error = std::string("pcsf_debug_program_facts: ") + ex.what();
        return PCSF_ERR_INVALID_ARG;
    }
    return PCSF_OK;
}

int pcsf_create(pcsf_ctx** out, const pcsf_model_desc* m, int device_id)
{
    if (!out || !m || !m->child1 || !m->child2 || !m->branch_len || !m->S_coding || !m->pi_coding || !m->S_noncoding ||
        !m->pi_noncoding) {
        g_create_error = "pcsf_create: null argument";
        return PCSF_ERR_INVALID_ARG;
    }
    *out = nullptr;
    int n_dev = 0;
    hipError_t e = hipGetDeviceCount(&n_dev);
    if (e != hipSuccess || n_dev <= 0) {
        g_create_error = "pcsf_create: no HIP device visible (this library has no CPU path)";
        return PCSF_ERR_NO_DEVICE;
    }
    if (device_id < 0 || device_id >= n_dev) {
        g_create_error = "pcsf_create: device_id out of range";
        return PCSF_ERR_INVALID_ARG;
    }
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device_id)) != hipSuccess || std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        g_create_error = std::string("pcsf_create: device is not gfx950 (") + (e == hipSuccess ? prop.gcnArchName : "?") + ")";
        return PCSF_ERR_NO_DEVICE;
    }
    std::unique_ptr<pcsf_ctx> c(new (std::nothrow) pcsf_ctx);
    if (!c) return PCSF_ERR_HIP;
    c->device = device_id;
    c->n_nodes = m->n_nodes;
    try {
        c->schedule = pcsf::build_schedule(pcsf::TreeView{m->n_nodes, m->child1, m->child2});
    } catch (const std::exception& ex) {
        g_create_error = std::string("pcsf_create: invalid tree: ") + ex.what();
        return PCSF_ERR_INVALID_ARG;
    }
    const int n = m->n_nodes, nl = (n + 1) / 2;
    c->n_leaves = nl;
    c->child1.assign(m->child1, m->child1 + n);
    c->child2.assign(m->child2, m->child2 + n);
    c->branch_len.assign(m->branch_len, m->branch_len + n);
    for (int i = 0; i < n - 1; ++i)
        if (!(c->branch_len[i] >= 0.0f)) {
            g_create_error = "pcsf_create: negative or NaN branch length";
            return PCSF_ERR_INVALID_ARG;
        }
    std::memcpy(c->S[0].data(), m->S_coding, sizeof(double) * 4096);
    std::memcpy(c->S[1].data(), m->S_noncoding, sizeof(double) * 4096);
    std::memcpy(c->pi[0].data(), m->pi_coding, sizeof(double) * 64);
    std::memcpy(c->pi[1].data(), m->pi_noncoding, sizeof(double) * 64);

    // PhyloCSFModel_make for both models: Q, eigen-system, P(t_b) at scale 1
    c->host_P.resize(static_cast<size_t>(2) * (n - 1) * 4096);
    pcsf::Vec prior[2];
    for (int k = 0; k < 2; ++k) {
        pcsf::Mat Q;
        pcsf::rate_matrix_ecm(c->S[k].data(), c->pi[k].data(), Q);
        if (!pcsf::eigen_reversible(Q, c->pi[k].data(), c->eig[k])) {
            g_create_error = "pcsf_create: codon frequencies must be positive";
            return PCSF_ERR_INVALID_ARG;
        }
        prior[k] = c->eig[k].prior;
        for (int b = 0; b < n - 1; ++b) {
            const double t = pcsf::scaled_branch(c->branch_len[b], 1.0);
            if (!pcsf::transition_matrix(c->eig[k], t, c->host_P.data() + (static_cast<size_t>(k) * (n - 1) + b) * 4096)) {
                g_create_error = "pcsf_create: transition matrix failed validation (branch " + std::to_string(b) + ")";
                return PCSF_ERR_MODEL;
            }
        }
    }

    if ((e = hipSetDevice(device_id)) != hipSuccess) {
        g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(e);
        return PCSF_ERR_HIP;
    }
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess ||
        (e = hipEventCreateWithFlags(&c->ev_kernels, hipEventDisableTiming)) != hipSuccess || (e = hipEventCreate(&c->ev0)) != hipSuccess || (e = hipEventCreate(&c->ev1)) != hipSuccess) {
        g_create_error = std::string("stream/event creation: ") + hipGetErrorString(e);
        return PCSF_ERR_HIP;
    }
    c->branch_slot.assign(n, -1);
    for (size_t sl = 0; sl < c->schedule.slot_node.size(); ++sl) c->branch_slot[c->schedule.slot_node[sl]] = static_cast<int>(sl);
    const auto& prog = c->schedule.program;
    if ((e = c->d_program.reserve(prog.size() * sizeof(uint32_t))) != hipSuccess ||
        (e = hipMemcpy(c->d_program.ptr, prog.data(), prog.size() * sizeof(uint32_t), hipMemcpyHostToDevice)) != hipSuccess) {
        g_create_error = std::string("program upload: ") + hipGetErrorString(e);
        return PCSF_ERR_HIP;
    }
    {   // static facts about the ops (factored block-mode kernel: skipping the ops of all-gap subtrees); trees of <= 128 leaves
        const pcsf::ProgramFacts facts = pcsf::program_facts(c->schedule);
        if (facts.ok) {
            if ((e = c->d_op_leaves.reserve(facts.leaves.size() * sizeof(uint64_t))) != hipSuccess ||
                (e = hipMemcpy(c->d_op_leaves.ptr, facts.leaves.data(), facts.leaves.size() * sizeof(uint64_t), hipMemcpyHostToDevice)) != hipSuccess ||
                (e = c->d_op_meta.reserve(facts.meta.size() * sizeof(uint32_t))) != hipSuccess ||
                (e = hipMemcpy(c->d_op_meta.ptr, facts.meta.data(), facts.meta.size() * sizeof(uint32_t), hipMemcpyHostToDevice)) != hipSuccess) {
                g_create_error = std::string("pcsf_create (program facts): ") + hipGetErrorString(e);
                return PCSF_ERR_HIP;
            }
        }
    }
    int rc = upload_model_tables(c.get(), c->host_P.data(), prior);
    if (rc != PCSF_OK) {
        g_create_error = c->error;
        return rc;
    }
    c->stagger_first_blocks = 2u * static_cast<uint32_t>(prop.multiProcessorCount);
    if (const char* v = std::getenv("PCSF_STAGGER")) c->stagger_sleep = std::atoi(v);
    if (const char* v = std::getenv("PCSF_PRIO_MODE")) c->prio_mode = std::atoi(v);
    if (const char* v = std::getenv("PCSF_EXTRA_LDS")) c->extra_lds = static_cast<size_t>(std::atoi(v));
    if (const char* v = std::getenv("PCSF_FUSED_FRONTEND")) c->fused_frontend = std::atoi(v) != 0;
    if (const char* v = std::getenv("PCSF_SKIP_ALLGAP")) {
        c->skip_allgap = std::atoi(v) != 0;
        c->skip_allgap_forced = true;
    }
    if (const char* v = std::getenv("PCSF_DUMP_CODONS")) c->dump_codons = std::atoi(v) != 0;
    if (c->skip_allgap && c->d_op_meta.ptr) {
        // What the full program makes of subtrees that are all gaps -- column-independent vectors close to 1 -- recorded once, by the
        // full program itself (diagnostic variant, one work-group of all-gap columns): the skipping variants substitute these values,
        // so their scores are bit-identical to the full program's.
        const size_t ne = c->schedule.slot_node.size(), doubles = 2 * (ne + 1) * 64;
        pcsf::DeviceBuffer d_gaps;
        const size_t nl = static_cast<size_t>(c->n_leaves);
        if ((e = c->d_gap_tab.reserve(doubles * sizeof(double))) != hipSuccess || (e = d_gaps.reserve(nl * 64)) != hipSuccess ||
            // on the stream the kernel below runs on: a hipMemset is a fill kernel on the NULL stream that returns before it has run, and
            // c->stream (non-blocking) does not wait for the null stream -- on a device busy with other contexts' work the kernel read
            // the all-gap columns before they were written and the table was zeroed after it had been filled: a context set up beside
            // a running one (build-tracks --devices, two contexts per GPU) then scored the work-groups that skip subtrees wrongly
            // (found by tests/test_gpu_full_size.py: four contexts on one GPU)
            (e = hipMemsetAsync(c->d_gap_tab.ptr, 0, doubles * sizeof(double), c->stream)) != hipSuccess ||
            (e = hipMemsetAsync(d_gaps.ptr, 64, nl * 64, c->stream)) != hipSuccess) {
            g_create_error = std::string("pcsf_create (all-gap messages): ") + hipGetErrorString(e);
            return PCSF_ERR_HIP;
        }
        pcsf::PruneParams gp = pcsf::base_prune_params(c.get());
        gp.gap_tab = nullptr;
        gp.gap_dump = c->d_gap_tab.as<double>();
        gp.stamps = nullptr;
        gp.codons = d_gaps.as<uint8_t>();
        gp.n_sites = 64;
        gp.leaf_stride = 64;
        if ((e = pcsf::launch_prune_fixed(gp, c->schedule.stack_depth > pcsf::MAX_STACK_DEPTH ? 66 : 2, c->stream, 0)) != hipSuccess ||
            (e = hipStreamSynchronize(c->stream)) != hipSuccess) {
            g_create_error = std::string("pcsf_create (all-gap messages): ") + hipGetErrorString(e);
            return PCSF_ERR_HIP;
        }
    }
    if (const char* v = std::getenv("PCSF_PRUNE_VARIANT")) {
        const int vi = std::atoi(v);
        if (vi >= 0 && vi < pcsf::prune_variant_count()) c->variant = vi;
    }
    {   // cherry tables (PCSF_CHERRY_TABLES=0: tracks mode runs the full program -- the A/B switch; same bits either way)
        const char* v = std::getenv("PCSF_CHERRY_TABLES");
        if (!(v && v[0] == '0') && (rc = build_cherry_tables(c.get(), prior)) != PCSF_OK) {
            if (c->error.compare(0, 9, "hipMalloc") != 0) {
                g_create_error = c->error;
                return rc;
            }
            // no room for the rows on this device (other contexts, other processes): tracks mode runs the full program
            (void)hipGetLastError();
            c->cherry.reset();
            c->error.clear();
        }
    }
    *out = c.release();
    return PCSF_OK;
}

void pcsf_destroy(pcsf_ctx* c)
{
    if (!c) return;
    (void)hipSetDevice(c->device);
    {
        KernelTurn& turn = g_kernel_turn[c->device & 63];
        std::lock_guard<std::mutex> g(turn.lock);
        if (turn.owner == c) {
            turn.last = nullptr;
            turn.owner = nullptr;
        }
    }
    delete c;
}

const char* pcsf_last_error(const pcsf_ctx* c) { return c ? c->error.c_str() : g_create_error.c_str(); }

int pcsf_n_leaves(const pcsf_ctx* c) { return c ? c->n_leaves : PCSF_ERR_INVALID_ARG; }

int pcsf_set_timing(pcsf_ctx* c, int enable)
{
    if (!c) return PCSF_ERR_INVALID_ARG;
    c->timing = enable != 0;
    return PCSF_OK;
}

double pcsf_last_kernel_ms(const pcsf_ctx* c) { return c ? c->last_kernel_ms : -1.0; }

// diagnostic: arm in-kernel stamps for work-group `block` (needs variant 2) / read them back
int pcsf_debug_stamps(pcsf_ctx* c, int block, unsigned long long* out, int cap)
{
    if (!c) return PCSF_ERR_INVALID_ARG;
    const size_t n = 16;  // eight per-segment cycle sums and eight counts (kernels.hip, STAMP build)
    hipError_t e;
    if (!out) {  // arm
        if ((e = hipSetDevice(c->device)) != hipSuccess) return c->hip_fail(e, "hipSetDevice");
        if ((e = c->d_stamps.reserve(n * 8)) != hipSuccess) return c->hip_fail(e, "hipMalloc(stamps)");
        if ((e = hipMemsetAsync(c->d_stamps.ptr, 0, n * 8, c->stream)) != hipSuccess || (e = hipStreamSynchronize(c->stream)) != hipSuccess) return c->hip_fail(e, "hipMemset(stamps)");
        c->stamp_block = static_cast<uint32_t>(block);
        return static_cast<int>(n);
    }
    if (static_cast<size_t>(cap) < n || !c->d_stamps.ptr) return PCSF_ERR_INVALID_ARG;
    if ((e = hipDeviceSynchronize()) != hipSuccess) return c->hip_fail(e, "hipDeviceSynchronize");
    if ((e = hipMemcpy(out, c->d_stamps.ptr, n * 8, hipMemcpyDeviceToHost)) != hipSuccess) return c->hip_fail(e, "hipMemcpy(stamps)");
    return static_cast<int>(n);
}

// diagnostics: the leaf-major codon matrix of the last six-frame call ([n_leaves][n_sites]; the separate extract kernel's, or,
// with PCSF_DUMP_CODONS=1, what the fused front end formed)
void pcsf_debug_eigen_schedule(int32_t* pairs, int32_t* located)
{
    pcsf::eigen_schedule_for_tests(pairs, located);
}

int pcsf_debug_eigen_schedule_quads(int32_t* pairs)
{
    return pcsf::eigen_schedule_quads_for_tests(pairs);
}

int pcsf_debug_codon_matrix(pcsf_ctx* c, uint8_t* out, size_t capacity, size_t* n_sites)
{
    if (!c || !out || !n_sites) return PCSF_ERR_INVALID_ARG;
    *n_sites = c->last_total_sites;
    const size_t bytes = static_cast<size_t>(c->n_leaves) * c->last_total_sites;
    if (bytes > capacity || !c->d_codons.ptr || c->d_codons.bytes < bytes) return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_debug_codon_matrix: no matrix / capacity");
    hipError_t e;
    if ((e = hipSetDevice(c->device)) != hipSuccess) return c->hip_fail(e, "hipSetDevice");
    if ((e = hipMemcpy(out, c->d_codons.ptr, bytes, hipMemcpyDeviceToHost)) != hipSuccess) return c->hip_fail(e, "hipMemcpy(codons)");
    return PCSF_OK;
}

int pcsf_get_branch_matrices(pcsf_ctx* c, double* out)
{
    if (!c || !out) return PCSF_ERR_INVALID_ARG;
    std::memcpy(out, c->host_P.data(), c->host_P.size() * sizeof(double));
    return PCSF_OK;
}

int pcsf_score_sites_fixed_dev(pcsf_ctx* c, const uint8_t* d_codons, size_t n_sites, size_t leaf_stride, double* d_out,
                               double* d_lnz, void* hip_stream)
{
    if (!c) return PCSF_ERR_INVALID_ARG;
    if (n_sites == 0) return PCSF_OK;
    if (!d_codons || (!d_out && !d_lnz) || leaf_stride < n_sites) return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_score_sites_fixed_dev: bad argument");
    hipError_t e;
    if ((e = hipSetDevice(c->device)) != hipSuccess) return c->hip_fail(e, "hipSetDevice");
    return pcsf::launch_fixed(c, d_codons, n_sites, leaf_stride, d_out, d_lnz, nullptr, static_cast<hipStream_t>(hip_stream));
}

int pcsf_score_sites_fixed(pcsf_ctx* c, const uint8_t* codons, size_t n_sites, size_t leaf_stride, double* out)
{
    if (!c) return PCSF_ERR_INVALID_ARG;
    if (n_sites == 0) return PCSF_OK;
    if (!codons || !out || leaf_stride < n_sites) return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_score_sites_fixed: bad argument");
    hipError_t e;
    if ((e = hipSetDevice(c->device)) != hipSuccess) return c->hip_fail(e, "hipSetDevice");
    const size_t nl = static_cast<size_t>(c->n_leaves);
    if ((e = c->d_codons.reserve(nl * n_sites)) != hipSuccess) return c->hip_fail(e, "hipMalloc(codons)");
    if ((e = c->d_out.reserve(n_sites * sizeof(double))) != hipSuccess) return c->hip_fail(e, "hipMalloc(out)");
    // compact the leaf-major matrix while copying (device stride = n_sites)
    if ((e = pcsf::upload_rows(c->d_codons.ptr, codons, leaf_stride, n_sites, nl, c->stream)) != hipSuccess) return c->hip_fail(e, "upload(codons)");
    int rc = pcsf::launch_fixed(c, c->d_codons.as<uint8_t>(), n_sites, n_sites, c->d_out.as<double>(), nullptr, nullptr, c->stream);
    if (rc != PCSF_OK) return rc;
    if ((e = hipMemcpyAsync(out, c->d_out.ptr, n_sites * sizeof(double), hipMemcpyDeviceToHost, c->stream)) != hipSuccess)
        return c->hip_fail(e, "hipMemcpyAsync(out)");
    if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return c->hip_fail(e, "hipStreamSynchronize");
    return PCSF_OK;
}

int pcsf_score_alignments_6frames(pcsf_ctx* c, const char* nucleotides, size_t row_stride, const pcsf_alignment_desc* alns, size_t n_alns,
                                  double* out_scores, size_t out_capacity, uint64_t* out_offsets, uint32_t* n_codons, uint32_t* skip_bases)
{
    return pcsf_score_alignments_6frames_bls(c, nucleotides, row_stride, alns, n_alns, out_scores, out_capacity, out_offsets, n_codons, skip_bases,
                                             nullptr);
}

namespace {
struct WigRequest {   // pcsf_score_alignments_6frames_wig: the call's wig text instead of its scores
    const char* const* chrom_names;
    float power_threshold;
    bool want_power, want_raw;
    uint64_t* out_bytes;       // [7]
    size_t* n_codon_frames;
    char* text;                // host destination of the seven texts back to back (may be null)
    size_t text_capacity;
};
}  // namespace

static int six_frames_impl(pcsf_ctx* c, const char* nucleotides, size_t row_stride, const pcsf_alignment_desc* alns, size_t n_alns,
                           double* out_scores, size_t out_capacity, uint64_t* out_offsets, uint32_t* n_codons, uint32_t* skip_bases,
                           double* out_bls, const WigRequest* wig);

int pcsf_score_alignments_6frames_bls(pcsf_ctx* c, const char* nucleotides, size_t row_stride, const pcsf_alignment_desc* alns, size_t n_alns,
                                      double* out_scores, size_t out_capacity, uint64_t* out_offsets, uint32_t* n_codons, uint32_t* skip_bases,
                                      double* out_bls)
{
    if (!c) return PCSF_ERR_INVALID_ARG;
    if (n_alns == 0) return PCSF_OK;
    if (!nucleotides || !alns || !out_scores || !out_offsets || !n_codons) return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_score_alignments_6frames: null argument");
    return six_frames_impl(c, nucleotides, row_stride, alns, n_alns, out_scores, out_capacity, out_offsets, n_codons, skip_bases, out_bls, nullptr);
}

// The six-frame call that keeps its results on the device and formats the wig text there (wig.hip): out_bytes[f] = bytes of file f's
// text of this call (0: PhyloCSFpower.wig, 1..6: PhyloCSFRaw+1 .. PhyloCSFRaw-3); pcsf_fetch_wig_text copies them out.
int pcsf_score_alignments_6frames_wig(pcsf_ctx* c, const char* nucleotides, size_t row_stride, const pcsf_alignment_desc* alns, size_t n_alns,
                                      const char* const* chrom_names, float power_threshold, int want_power, int want_raw, char* text, size_t text_capacity,
                                      uint64_t* out_bytes, size_t* n_codon_frames)
{
    if (!c) return PCSF_ERR_INVALID_ARG;
    if (!out_bytes) return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_score_alignments_6frames_wig: null argument");
    for (int f = 0; f < 7; ++f) out_bytes[f] = 0;
    if (n_codon_frames) *n_codon_frames = 0;
    c->wig.ready = false;
    if (n_alns == 0) return PCSF_OK;
    if (!nucleotides || !alns || !chrom_names) return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_score_alignments_6frames_wig: null argument");
    if (!c->bls_ready) return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_score_alignments_6frames_wig: call pcsf_set_bls_branch_lengths first");
    const WigRequest req{chrom_names, power_threshold, want_power != 0, want_raw != 0, out_bytes, n_codon_frames, text, text ? text_capacity : 0};
    std::vector<uint64_t> offsets(n_alns * 6);
    std::vector<uint32_t> counts(n_alns * 6);
    return six_frames_impl(c, nucleotides, row_stride, alns, n_alns, nullptr, ~size_t(0), offsets.data(), counts.data(), nullptr, nullptr, &req);
}

int pcsf_fetch_wig_text(pcsf_ctx* c, char* const* dst)
{
    if (!c) return PCSF_ERR_INVALID_ARG;
    if (!dst) return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_fetch_wig_text: null argument");
    if (!c->wig.ready) return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_fetch_wig_text: no text (call pcsf_score_alignments_6frames_wig first)");
    hipError_t e;
    if ((e = hipSetDevice(c->device)) != hipSuccess) return c->hip_fail(e, "hipSetDevice");
    bool contiguous = true;   // the seven destinations back to back, like the text on the device: one copy
    const char* first = nullptr;
    for (int f = 0; f < 7; ++f) {
        const uint64_t n = c->wig.file_off[f + 1] - c->wig.file_off[f];
        if (!n) continue;
        if (!dst[f]) return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_fetch_wig_text: null destination for a file with text");
        if (!first) first = dst[f] - (c->wig.file_off[f] - c->wig.file_off[0]);
        contiguous = contiguous && dst[f] == first + (c->wig.file_off[f] - c->wig.file_off[0]);
    }
    if (contiguous && first) {
        if ((e = hipMemcpyAsync(const_cast<char*>(first), c->wig.text.as<char>() + c->wig.file_off[0], c->wig.file_off[7] - c->wig.file_off[0], hipMemcpyDeviceToHost, c->stream)) != hipSuccess)
            return c->hip_fail(e, "hipMemcpyAsync(wig text)");
    } else {
        for (int f = 0; f < 7; ++f) {
            const uint64_t n = c->wig.file_off[f + 1] - c->wig.file_off[f];
            if (n && (e = hipMemcpyAsync(dst[f], c->wig.text.as<char>() + c->wig.file_off[f], n, hipMemcpyDeviceToHost, c->stream)) != hipSuccess)
                return c->hip_fail(e, "hipMemcpyAsync(wig text)");
        }
    }
    if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return c->hip_fail(e, "hipStreamSynchronize");
    c->wig.ready = false;
    return PCSF_OK;
}

// bytes of text buffer per line candidate before the sizes are known (PCSF_WIG_TEXT_ESTIMATE: tests make it too small on purpose)
static uint64_t wig_text_estimate()
{
    static const uint64_t v = [] {
        const char* s = std::getenv("PCSF_WIG_TEXT_ESTIMATE");
        return s ? std::max<uint64_t>(strtoull(s, nullptr, 10), 0) : uint64_t(12);
    }();
    return v;
}

size_t pcsf_wig_text_estimate(size_t n_columns, size_t n_alignments)
{
    // line candidates: six frames of at most n_columns / 3 codons, and the power track's n_columns / 3
    return static_cast<size_t>(wig_text_estimate()) * (2 * n_columns + n_columns / 3 + 1) + 128 * 7 * n_alignments + 64;
}

static int six_frames_impl(pcsf_ctx* c, const char* nucleotides, size_t row_stride, const pcsf_alignment_desc* alns, size_t n_alns,
                           double* out_scores, size_t out_capacity, uint64_t* out_offsets, uint32_t* n_codons, uint32_t* skip_bases,
                           double* out_bls, const WigRequest* wig)
{
    static const bool TIMING = std::getenv("PCSF_TIME_6F") != nullptr;   // one line per call on stderr: host milliseconds of its phases
    const auto T0 = std::chrono::steady_clock::now();
    auto since = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - T0).count() * 1e3; };
    double t_enq = 0, t_turn = 0, t_sync = 0;
    std::vector<pcsf::FrameSegment> seg;
    std::vector<pcsf::BlsSpan> spans;  // like seg: source of an asynchronous copy, alive until the stream is synchronised
    seg.reserve(n_alns * 6);
    uint64_t total = 0, max_col = 0;
    uint32_t max_codons = 0;
    for (size_t k = 0; k < n_alns; ++k) {
        const pcsf_alignment_desc& a = alns[k];
        if (a.col_offset + a.len > row_stride) return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_score_alignments_6frames: alignment exceeds row_stride");
        max_col = std::max<uint64_t>(max_col, a.col_offset + a.len);
        if ((out_bls || wig) && a.len) spans.push_back(pcsf::BlsSpan{a.col_offset, a.len});
        for (int f = 0; f < 6; ++f) {
            const pcsf::FrameView v = pcsf::frame_view(a.start_pos, a.chrom_len, a.len, f < 3 ? '+' : '-', static_cast<unsigned>(f % 3 + 1));
            pcsf::FrameSegment s{};
            s.col0 = a.col_offset;
            s.dst0 = total;
            s.len = a.len;
            s.skip = v.skip;
            s.n_codons = static_cast<uint32_t>(v.n_codons);
            s.minus = f >= 3;
            out_offsets[k * 6 + f] = total;
            n_codons[k * 6 + f] = s.n_codons;
            if (skip_bases) skip_bases[k * 6 + f] = v.skip;
            total += s.n_codons;
            max_codons = std::max(max_codons, s.n_codons);
            if (s.n_codons) seg.push_back(s);
        }
    }
    if (total > out_capacity) return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_score_alignments_6frames: out_capacity too small");
    hipError_t e;
    if ((e = hipSetDevice(c->device)) != hipSuccess) return c->hip_fail(e, "hipSetDevice");
    // every return below that leaves work queued -- an error half way -- waits for it first: the queued copies read vectors of this frame
    // and write the caller's buffers
    struct Drain {
        hipStream_t s;
        bool armed = true;
        ~Drain() { if (armed) (void)hipStreamSynchronize(s); }
    } drain{c->stream};
    if (total == 0 && !wig) {  // alignments shorter than a codon still have branch-length scores
        if (!out_bls || max_col == 0) return PCSF_OK;
        return pcsf::bls_columns(c, nucleotides, row_stride, max_col, spans, out_bls);
    }
    if (wig && max_col == 0) {   // nothing but empty alignments: seven empty texts
        for (uint64_t& v : c->wig.file_off) v = 0;
        c->wig.ready = true;
        return PCSF_OK;
    }
    // One stream, in order: rows up, branch-length kernel, frame extraction, pruning, then ALL downloads (a download
    // between two kernels idles the compute queue while the copy engine works).  Splitting the call into chunks on
    // two streams so that transfers overlap kernels was measured and gains nothing: with page-locked buffers the
    // transfers are 0.5 ms of a 12 ms call and the extra launches and kernel tails cost as much.
    const size_t nl = static_cast<size_t>(c->n_leaves);
    // The six-frame translation runs inside the pruning kernel (kernels.hip, FUSED): its work-groups form their codon
    // columns from the resident nucleotide rows, so no codon matrix is written or re-read.  PCSF_FUSED_FRONTEND=0 keeps
    // the separate extract_frames_kernel (A/B measurements; results are identical).
    const bool fused = c->fused_frontend;
    std::vector<uint32_t> wg_first_seg;
    if (fused && total) {  // segment that holds site 64 b, for every work-group b
        wg_first_seg.resize((total + 63) / 64);
        size_t k = 0;
        for (size_t b = 0; b < wg_first_seg.size(); ++b) {
            while (b * 64 >= seg[k].dst0 + seg[k].n_codons) ++k;
            wg_first_seg[b] = static_cast<uint32_t>(k);
        }
    }
    if ((e = c->d_nucl.reserve(nl * max_col)) != hipSuccess) return c->hip_fail(e, "hipMalloc(nucleotides)");
    if ((e = c->d_segments.reserve(std::max<size_t>(seg.size(), 1) * sizeof(pcsf::FrameSegment))) != hipSuccess) return c->hip_fail(e, "hipMalloc(segments)");
    if ((!fused || c->dump_codons) && (e = c->d_codons.reserve(nl * total)) != hipSuccess) return c->hip_fail(e, "hipMalloc(codons)");
    c->last_total_sites = total;
    if (fused && (e = c->d_wg_seg.reserve(std::max<size_t>(wg_first_seg.size(), 1) * sizeof(uint32_t))) != hipSuccess) return c->hip_fail(e, "hipMalloc(work-group segments)");
    if ((e = c->d_out.reserve(std::max<uint64_t>(total, 1) * sizeof(double))) != hipSuccess) return c->hip_fail(e, "hipMalloc(out)");
    if ((e = pcsf::upload_rows(c->d_nucl.ptr, nucleotides, row_stride, max_col, nl, c->stream)) != hipSuccess) return c->hip_fail(e, "upload(nucleotides)");
    if (!seg.empty() && (e = hipMemcpyAsync(c->d_segments.ptr, seg.data(), seg.size() * sizeof(pcsf::FrameSegment), hipMemcpyHostToDevice, c->stream)) != hipSuccess)
        return c->hip_fail(e, "hipMemcpyAsync(segments)");
    if (fused && !wg_first_seg.empty() && (e = hipMemcpyAsync(c->d_wg_seg.ptr, wg_first_seg.data(), wg_first_seg.size() * sizeof(uint32_t), hipMemcpyHostToDevice, c->stream)) != hipSuccess)
        return c->hip_fail(e, "hipMemcpyAsync(work-group segments)");
    // ---- wig text on the device: the line candidates of the seven files (records and names uploaded before this call's turn on the device)
    pcsf::WigParams wp{};
    uint64_t text_copied = 0;               // bytes of text queued for the caller's buffer before the sizes were known
    size_t wig_temp_bytes = 0;              // scratch of the prefix sum
    std::vector<pcsf::WigFrame> frames;     // sources of asynchronous copies: alive until the stream is synchronised below
    std::vector<char> chroms;
    if (wig) {
        std::vector<uint32_t> chrom0(n_alns), chrom_len(n_alns);
        for (size_t k = 0; k < n_alns; ++k) {
            const char* name = wig->chrom_names[k] ? wig->chrom_names[k] : "";
            chrom0[k] = static_cast<uint32_t>(chroms.size());
            chrom_len[k] = static_cast<uint32_t>(strlen(name));
            chroms.insert(chroms.end(), name, name + chrom_len[k]);
        }
        uint64_t elems = 0;
        for (int file = 0; file < 7; ++file) {
            wp.file_frame0[file] = static_cast<uint32_t>(frames.size());
            wp.file_elem0[file] = elems;
            if (file == 0 ? !wig->want_power : !wig->want_raw) continue;
            for (size_t k = 0; k < n_alns; ++k) {
                const pcsf_alignment_desc& a = alns[k];
                pcsf::WigFrame fr{};
                fr.elem0 = elems;
                fr.chrom0 = chrom0[k];
                fr.chrom_len = chrom_len[k];
                if (file == 0) {   // src/phylocsf++build_tracks.hpp:139-156: frame 3 of the + strand, no filter
                    int64_t skip = static_cast<int64_t>(3 - static_cast<int64_t>(a.start_pos)) % 3;
                    if (skip < 0) skip += 3;
                    fr.n = a.len > static_cast<uint64_t>(skip) + 2 ? static_cast<uint32_t>((a.len - static_cast<uint64_t>(skip)) / 3) : 0u;
                    fr.bls0 = a.col_offset + static_cast<uint64_t>(skip);
                    fr.start_pos = static_cast<int64_t>(a.start_pos) + skip;
                } else {
                    const int fi = file - 1;
                    const pcsf::FrameView v = pcsf::frame_view(a.start_pos, a.chrom_len, a.len, fi < 3 ? '+' : '-', static_cast<unsigned>(fi % 3 + 1));
                    fr.n = static_cast<uint32_t>(v.n_codons);
                    fr.score0 = out_offsets[k * 6 + fi];
                    fr.bls0 = a.col_offset + v.bls_offset;
                    fr.start_pos = static_cast<int64_t>(v.start_pos);
                }
                if (fr.n == 0) continue;
                elems += fr.n;
                frames.push_back(fr);
            }
        }
        wp.file_frame0[7] = static_cast<uint32_t>(frames.size());
        wp.file_elem0[7] = elems;
        wp.n_elems = elems;
        {   // byte offsets are 32-bit: a call whose text could pass 4 GB (far beyond build-tracks' batches) is the host's
            uint32_t longest_name = 0;
            for (uint32_t v : chrom_len) longest_name = std::max(longest_name, v);
            if (elems >= 0xffffffffull / (static_cast<uint64_t>(longest_name) + 96))
                return c->fail(PCSF_ERR_UNSUPPORTED, "pcsf_score_alignments_6frames_wig: too many lines in one call -- format this batch on the host");
        }
        pcsf_ctx::WigBuffers& W = c->wig;
        const size_t temp_bytes = wig_temp_bytes = pcsf::wig_scan_temp_bytes(elems);
        if (!W.h_bounds && (e = hipHostMalloc(reinterpret_cast<void**>(&W.h_bounds), 16 * sizeof(uint32_t), hipHostMallocDefault)) != hipSuccess)
            return c->hip_fail(e, "hipHostMalloc(wig bounds)");
        if ((e = W.frames.reserve(std::max<size_t>(frames.size(), 1) * sizeof(pcsf::WigFrame))) != hipSuccess || (e = W.chroms.reserve(std::max<size_t>(chroms.size(), 1))) != hipSuccess ||
            (e = W.lens.reserve((elems + 1) * sizeof(uint32_t))) != hipSuccess || (e = W.offsets.reserve((elems + 1) * sizeof(uint32_t))) != hipSuccess ||
            (e = W.temp.reserve(std::max<size_t>(temp_bytes, 16))) != hipSuccess || (e = W.flag.reserve(9 * sizeof(uint32_t))) != hipSuccess ||
            (e = W.text.reserve(wig_text_estimate() * elems + 128 * frames.size() + 64)) != hipSuccess)   // an estimate: a line is ~8.4 bytes (wig.hip)
            return c->hip_fail(e, "hipMalloc(wig)");
        if (!frames.empty() && (e = hipMemcpyAsync(W.frames.ptr, frames.data(), frames.size() * sizeof(pcsf::WigFrame), hipMemcpyHostToDevice, c->stream)) != hipSuccess)
            return c->hip_fail(e, "hipMemcpyAsync(wig frames)");
        if (!chroms.empty() && (e = hipMemcpyAsync(W.chroms.ptr, chroms.data(), chroms.size(), hipMemcpyHostToDevice, c->stream)) != hipSuccess)
            return c->hip_fail(e, "hipMemcpyAsync(wig chromosome names)");
        if ((e = hipMemsetAsync(W.flag.ptr, 0, 9 * sizeof(uint32_t), c->stream)) != hipSuccess) return c->hip_fail(e, "hipMemsetAsync(wig bounds)");
        wp.frames = W.frames.as<pcsf::WigFrame>();
        wp.scores = c->d_out.as<double>();
        wp.bls = c->d_bls_out.as<double>();
        wp.chroms = W.chroms.as<char>();
        wp.threshold3 = wig->power_threshold * 3;
        wp.lens = W.lens.as<uint32_t>();
        wp.offsets = W.offsets.as<uint32_t>();
        wp.bounds = W.flag.as<uint32_t>();
        wp.needs_host = wp.bounds + 8;
        wp.text = W.text.as<char>();
        wp.text_capacity = W.text.bytes;
    }
    int rc;
    KernelTurn& turn = g_kernel_turn[c->device & 63];
    t_enq = since();
    std::unique_lock<std::mutex> my_turn(turn.lock);
    t_turn = since();
    if (turn.last && turn.owner != c && (e = hipStreamWaitEvent(c->stream, turn.last, 0)) != hipSuccess) return c->hip_fail(e, "hipStreamWaitEvent");
    // the power track's per-base scores from the same resident rows (that kernel also checks every character); without
    // them the translation does the character check itself, so that both entry points reject the same input
    unsigned long long* d_char_flag = nullptr;
    if (out_bls || wig) {
        if ((rc = pcsf::bls_enqueue(c, c->d_nucl.as<unsigned char>(), max_col, max_col, spans)) != PCSF_OK) return rc;
    } else {
        if ((e = c->d_bls_flag.reserve(sizeof(unsigned long long))) != hipSuccess) return c->hip_fail(e, "hipMalloc(character flag)");
        if (!c->bls_first_bad && (e = hipHostMalloc(reinterpret_cast<void**>(&c->bls_first_bad), sizeof(unsigned long long), hipHostMallocDefault)) != hipSuccess)
            return c->hip_fail(e, "hipHostMalloc(character flag)");
        if ((e = hipMemsetAsync(c->d_bls_flag.ptr, 0xff, sizeof(unsigned long long), c->stream)) != hipSuccess) return c->hip_fail(e, "hipMemsetAsync(character flag)");
        d_char_flag = c->d_bls_flag.as<unsigned long long>();
    }
    if (total == 0) {
        // (wig request on alignments shorter than a codon: only the power track has text)
    } else if (fused) {
        pcsf::FusedFrontEnd fe{c->d_nucl.as<unsigned char>(), max_col, c->d_segments.as<pcsf::FrameSegment>(), c->d_wg_seg.as<uint32_t>(), d_char_flag,
                               c->dump_codons ? c->d_codons.as<unsigned char>() : nullptr};
        if ((rc = pcsf::launch_fixed(c, nullptr, total, total, c->d_out.as<double>(), nullptr, nullptr, c->stream, &fe)) != PCSF_OK) return rc;
    } else {
        // at most 65535 segments per launch (grid.y)
        for (size_t s0 = 0; s0 < seg.size(); s0 += 65535) {
            const uint32_t ns = static_cast<uint32_t>(std::min<size_t>(65535, seg.size() - s0));
            if ((e = pcsf::launch_extract_frames(c->d_segments.as<pcsf::FrameSegment>() + s0, ns, max_codons, c->d_nucl.as<unsigned char>(), max_col,
                                                 c->n_leaves, c->d_codons.as<unsigned char>(), total, d_char_flag, c->stream)) != hipSuccess)
                return c->hip_fail(e, "launch extract_frames");
        }
        if ((rc = pcsf::launch_fixed(c, c->d_codons.as<uint8_t>(), total, total, c->d_out.as<double>(), nullptr, nullptr, c->stream)) != PCSF_OK) return rc;
    }
    if (wig) {   // the line candidates' lengths, their byte offsets, the text: behind the scoring kernels, no host in between
        pcsf_ctx::WigBuffers& W = c->wig;
        wp.scores = c->d_out.as<double>();   // (the branch-length kernel's output buffer may have grown since the records were made)
        wp.bls = c->d_bls_out.as<double>();
        if ((e = pcsf::launch_wig_lengths(wp, W.temp.ptr, wig_temp_bytes, c->stream)) != hipSuccess) return c->hip_fail(e, "launch wig lengths");
        if ((e = pcsf::launch_wig_write(wp, c->stream)) != hipSuccess) return c->hip_fail(e, "launch wig write");
    }
    if ((e = hipEventRecord(c->ev_kernels, c->stream)) != hipSuccess) return c->hip_fail(e, "hipEventRecord");
    turn.last = c->ev_kernels;
    turn.owner = c;
    my_turn.unlock();
    if (!wig && (e = hipMemcpyAsync(out_scores, c->d_out.ptr, total * sizeof(double), hipMemcpyDeviceToHost, c->stream)) != hipSuccess)
        return c->hip_fail(e, "hipMemcpyAsync(out)");
    if (out_bls && (rc = pcsf::bls_download(c, max_col, out_bls)) != PCSF_OK) return rc;
    if (wig) {   // the character check of bls_kernel, the byte offsets of the eight file boundaries, the needs-host flag
        pcsf_ctx::WigBuffers& W = c->wig;
        *c->bls_first_bad = pcsf::BLS_NO_BAD;
        if ((e = hipMemcpyAsync(c->bls_first_bad, c->d_bls_flag.ptr, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream)) != hipSuccess)
            return c->hip_fail(e, "hipMemcpyAsync(bls flag)");
        if ((e = hipMemcpyAsync(W.h_bounds, W.flag.ptr, 9 * sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream)) != hipSuccess) return c->hip_fail(e, "hipMemcpyAsync(wig bounds)");
        // the text itself, before its size is known: as much as the caller's buffer and the device's hold.  (A copy queued once the
        // sizes are on the host waits behind the OTHER context's pruning kernel, which was queued meanwhile: 7.5 ms per call measured.)
        text_copied = std::min<uint64_t>(wig->text_capacity, wp.n_elems ? wp.text_capacity : 0);
        if (text_copied && (e = hipMemcpyAsync(wig->text, wp.text, text_copied, hipMemcpyDeviceToHost, c->stream)) != hipSuccess) return c->hip_fail(e, "hipMemcpyAsync(wig text)");
    }
    if (d_char_flag) {
        *c->bls_first_bad = pcsf::BLS_NO_BAD;
        if ((e = hipMemcpyAsync(c->bls_first_bad, d_char_flag, sizeof(unsigned long long), hipMemcpyDeviceToHost, c->stream)) != hipSuccess)
            return c->hip_fail(e, "hipMemcpyAsync(character flag)");
    }
    const double t_unl = since();
    if ((e = hipStreamSynchronize(c->stream)) != hipSuccess) return c->hip_fail(e, "hipStreamSynchronize");
    t_sync = since();
    if (TIMING) fprintf(stderr, "6f %s ctx %p: cols %zu, before turn %.3f, wait turn %.3f, enqueue in turn %.3f, sync %.3f ms\n", wig ? "wig" : "dl", (void*)c, (size_t)max_col, t_enq, t_turn - t_enq, t_unl - t_turn, t_sync - t_unl);
    if ((rc = pcsf::bls_check(c, nucleotides, row_stride, max_col)) != PCSF_OK) return rc;
    if (wig) {
        pcsf_ctx::WigBuffers& W = c->wig;
        if (W.h_bounds[8]) return c->fail(PCSF_ERR_UNSUPPORTED, "pcsf_score_alignments_6frames_wig: a value of 9e14 or more -- format this batch on the host");
        for (int f = 0; f < 8; ++f) W.file_off[f] = W.h_bounds[f];
        const uint64_t bytes = W.file_off[7];
        if (bytes > wp.text_capacity) {   // the estimate was too small (lines of twelve bytes and more on average): write again into enough room
            if ((e = W.text.reserve(bytes)) != hipSuccess) return c->hip_fail(e, "hipMalloc(wig text)");
            wp.text = W.text.as<char>();
            wp.text_capacity = W.text.bytes;
            if ((e = pcsf::launch_wig_write(wp, c->stream)) != hipSuccess) return c->hip_fail(e, "launch wig write");
            text_copied = 0;
        }
        if (bytes > text_copied && bytes <= wig->text_capacity) {   // the caller's buffer holds it but the early copy did not bring all of it
            if ((e = hipMemcpyAsync(wig->text, wp.text, bytes, hipMemcpyDeviceToHost, c->stream)) != hipSuccess || (e = hipStreamSynchronize(c->stream)) != hipSuccess)
                return c->hip_fail(e, "hipMemcpyAsync(wig text)");
        }
        for (int f = 0; f < 7; ++f) wig->out_bytes[f] = W.file_off[f + 1] - W.file_off[f];
        if (wig->n_codon_frames) *wig->n_codon_frames = total;
        W.ready = true;
    }
    drain.armed = false;   // (everything has been waited for, but a second write pass of the text, which pcsf_fetch_wig_text waits for)
    return PCSF_OK;
}

// Gets the context ready for six-frame calls of up to max_columns alignment columns: device scratch reserved, and one
// call on a dummy alignment so that the kernels' code objects are loaded -- what the first real call of a short run
// (one 1 Mb file = three calls) would otherwise pay for.  Optional; results never depend on it.
int pcsf_prepare_tracks(pcsf_ctx* c, size_t max_columns)
{
    if (!c) return PCSF_ERR_INVALID_ARG;
    hipError_t e;
    if ((e = hipSetDevice(c->device)) != hipSuccess) return c->hip_fail(e, "hipSetDevice");
    const size_t nl = static_cast<size_t>(c->n_leaves), cols = std::max<size_t>(max_columns, 192);
    if ((e = c->d_nucl.reserve(nl * cols)) != hipSuccess) return c->hip_fail(e, "hipMalloc(nucleotides)");
    if ((e = c->d_segments.reserve(size_t(1) << 20)) != hipSuccess) return c->hip_fail(e, "hipMalloc(segments)");
    if ((e = c->d_out.reserve(2 * cols * sizeof(double))) != hipSuccess) return c->hip_fail(e, "hipMalloc(out)");
    if (c->fused_frontend) {
        if ((e = c->d_wg_seg.reserve((2 * cols / 64 + 1) * sizeof(uint32_t))) != hipSuccess) return c->hip_fail(e, "hipMalloc(work-group segments)");
    } else if ((e = c->d_codons.reserve(nl * 2 * cols)) != hipSuccess) {
        return c->hip_fail(e, "hipMalloc(codons)");
    }
    if (c->bls_ready) {
        if ((e = c->d_bls_out.reserve(cols * sizeof(double))) != hipSuccess) return c->hip_fail(e, "hipMalloc(bls out)");
        if ((e = c->d_bls_spans.reserve(size_t(1) << 18)) != hipSuccess) return c->hip_fail(e, "hipMalloc(bls spans)");
    }
    std::vector<char> rows(nl * 192, 'A');
    std::vector<double> scores(2 * 192), bls(192);
    const pcsf_alignment_desc aln{0, 192, 1, 1000};
    uint64_t offsets[6];
    uint32_t n_codons[6];
    if (c->bls_ready) {
        // ... and the text path (wig.hip): its buffers for max_columns, its kernels and the scan's loaded by a call of that form -- the first
        // real call of a 1 Mb run otherwise spent 6-7 ms loading them and the next two 5 + 1 ms growing the buffers
        const uint64_t elems = 2 * static_cast<uint64_t>(cols) + cols / 3 + 8;
        pcsf_ctx::WigBuffers& W = c->wig;
        if ((e = W.lens.reserve((elems + 1) * sizeof(uint32_t))) != hipSuccess || (e = W.offsets.reserve((elems + 1) * sizeof(uint32_t))) != hipSuccess ||
            (e = W.temp.reserve(std::max<size_t>(pcsf::wig_scan_temp_bytes(elems), 16))) != hipSuccess ||
            (e = W.text.reserve(pcsf_wig_text_estimate(cols, 4096))) != hipSuccess || (e = W.frames.reserve(7 * 4096 * sizeof(pcsf::WigFrame))) != hipSuccess ||
            (e = W.chroms.reserve(1 << 16)) != hipSuccess)
            return c->hip_fail(e, "hipMalloc(wig)");
        const char* name = "x";
        uint64_t bytes[7];
        size_t frames = 0;
        const int rc = pcsf_score_alignments_6frames_wig(c, rows.data(), 192, &aln, 1, &name, 0.1f, 1, 1, nullptr, 0, bytes, &frames);
        if (rc != PCSF_OK) return rc;
    }
    return pcsf_score_alignments_6frames_bls(c, rows.data(), 192, &aln, 1, scores.data(), scores.size(), offsets, n_codons, nullptr,
                                             c->bls_ready ? bls.data() : nullptr);
}

// ---- branch-length score (src/additional_scores.hpp:5-84) --------------------------------------------------
int pcsf_set_bls_branch_lengths(pcsf_ctx* c, const double* branch_len)
{
    if (!c) return PCSF_ERR_INVALID_ARG;
    if (!branch_len) return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_set_bls_branch_lengths: null argument");
    const int n = c->n_nodes, nl = c->n_leaves;
    if (nl > 128 || n > pcsf::BLS_MAX_NODES) return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_set_bls_branch_lengths: more than 128 species");
    hipError_t e;
    if ((e = hipSetDevice(c->device)) != hipSuccess) return c->hip_fail(e, "hipSetDevice");
    // one allocation: len[n] | range[n][2] | child1[n] | child2[n] | prog[n - nl]
    std::vector<uint64_t> range(static_cast<size_t>(n) * 2, 0);
    for (int i = 0; i < n; ++i) {
        if (i < nl) {
            range[2 * i + (i >> 6)] = uint64_t(1) << (i & 63);
        } else {
            for (int w = 0; w < 2; ++w) range[2 * i + w] = range[2 * c->child1[i] + w] | range[2 * c->child2[i] + w];
        }
    }
    // The internal nodes as a stack program for bls_kernel: a node runs when both children's values exist; of two internal children
    // the one whose subtree needs the deeper stack goes first (Sethi-Ullman), so at most log2(internal nodes) + 1 values wait.
    std::vector<uint32_t> prog;
    {
        std::vector<int> need(static_cast<size_t>(n), 0);   // stack entries the subtree of an internal node needs, its own result included
        for (int i = nl; i < n; ++i) {
            const int l = c->child1[i], r = c->child2[i];
            const int a = l >= nl ? need[l] : 0, b = r >= nl ? need[r] : 0;
            need[i] = std::max(1, a == 0 || b == 0 ? std::max(a, b) : std::max(std::max(a, b), std::min(a, b) + 1));
        }
        if (n > nl && need[n - 1] > pcsf::BLS_STACK_DEPTH) return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_set_bls_branch_lengths: tree too deep for the branch-length kernel");
        std::vector<std::pair<int, int>> todo;   // (node, stage) -- an explicit stack: the tree may be a chain of 127 internal nodes
        if (n > nl) todo.push_back({n - 1, 0});
        while (!todo.empty()) {
            auto [i, stage] = todo.back();
            todo.pop_back();
            const int l = c->child1[i], r = c->child2[i];
            const bool li = l >= nl, ri = r >= nl;
            const bool left_first = !(li && ri) || need[l] >= need[r];
            if (stage == 0) {
                todo.push_back({i, 1});
                // pushed in reverse: the child to run first is popped first
                if (left_first) {
                    if (ri) todo.push_back({r, 0});
                    if (li) todo.push_back({l, 0});
                } else {
                    todo.push_back({l, 0});
                    todo.push_back({r, 0});
                }
            } else {
                uint32_t flags = (li ? 0u : pcsf::BLS_LEFT_LEAF) | (ri ? 0u : pcsf::BLS_RIGHT_LEAF);
                if (li && ri && left_first) flags |= pcsf::BLS_RIGHT_ON_TOP;
                prog.push_back(static_cast<uint32_t>(i) | static_cast<uint32_t>(l) << 8 | static_cast<uint32_t>(r) << 16 | flags << 24);
            }
        }
    }
    const size_t off_range = sizeof(double) * n, off_c1 = off_range + sizeof(uint64_t) * 2 * n, off_c2 = off_c1 + ((sizeof(int16_t) * n + 7) & ~size_t(7));
    const size_t off_prog = off_c2 + ((sizeof(int16_t) * n + 7) & ~size_t(7));
    const size_t bytes = off_prog + sizeof(uint32_t) * std::max<size_t>(prog.size(), 1);
    std::vector<unsigned char> host(bytes, 0);
    std::memcpy(host.data() + off_prog, prog.data(), sizeof(uint32_t) * prog.size());
    std::memcpy(host.data(), branch_len, sizeof(double) * n);
    std::memcpy(host.data() + off_range, range.data(), sizeof(uint64_t) * 2 * n);
    std::memcpy(host.data() + off_c1, c->child1.data(), sizeof(int16_t) * n);
    std::memcpy(host.data() + off_c2, c->child2.data(), sizeof(int16_t) * n);
    if ((e = c->d_bls_tree.reserve(bytes)) != hipSuccess) return c->hip_fail(e, "hipMalloc(bls tree)");
    if ((e = c->d_bls_flag.reserve(sizeof(unsigned long long))) != hipSuccess) return c->hip_fail(e, "hipMalloc(bls flag)");
    if (!c->bls_first_bad && (e = hipHostMalloc(reinterpret_cast<void**>(&c->bls_first_bad), sizeof(unsigned long long), hipHostMallocDefault)) != hipSuccess)
        return c->hip_fail(e, "hipHostMalloc(bls flag)");
    if ((e = hipMemcpy(c->d_bls_tree.ptr, host.data(), bytes, hipMemcpyHostToDevice)) != hipSuccess) return c->hip_fail(e, "hipMemcpy(bls tree)");
    unsigned char* base = c->d_bls_tree.as<unsigned char>();
    c->bls_tree.len = reinterpret_cast<const double*>(base);
    c->bls_tree.range = reinterpret_cast<const uint64_t*>(base + off_range);
    c->bls_tree.child1 = reinterpret_cast<const int16_t*>(base + off_c1);
    c->bls_tree.child2 = reinterpret_cast<const int16_t*>(base + off_c2);
    c->bls_tree.n_nodes = n;
    c->bls_tree.n_leaves = nl;
    c->bls_tree.prog = reinterpret_cast<const uint32_t*>(base + off_prog);
    c->bls_tree.n_prog = static_cast<int>(prog.size());
    // the divisor is the score of the full species set: one column of 'A's through the same kernel (divided by 1)
    c->bls_tree.total = 1.0;
    c->bls_ready = true;
    std::vector<char> all(static_cast<size_t>(nl), 'A');
    double total = 0.0;
    const int rc = pcsf::bls_columns(c, all.data(), 1, 1, {pcsf::BlsSpan{0, 1}}, &total);
    if (rc != PCSF_OK) {
        c->bls_ready = false;
        return rc;
    }
    if (nl < 2) total = 1.0;  // a single species never scores; avoid 0/0 below
    c->bls_tree.total = total;
    return PCSF_OK;
}

int pcsf_branch_length_scores(pcsf_ctx* c, const char* nucleotides, size_t row_stride, size_t n_columns, double* out_bls)
{
    if (!c) return PCSF_ERR_INVALID_ARG;
    if (n_columns == 0) return PCSF_OK;
    if (!nucleotides || !out_bls || row_stride < n_columns) return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_branch_length_scores: bad argument");
    if (!c->bls_ready) return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_branch_length_scores: call pcsf_set_bls_branch_lengths first");
    hipError_t e;
    if ((e = hipSetDevice(c->device)) != hipSuccess) return c->hip_fail(e, "hipSetDevice");
    return pcsf::bls_columns(c, nucleotides, row_stride, n_columns, {pcsf::BlsSpan{0, n_columns}}, out_bls);
}

int pcsf_score_blocks(pcsf_ctx* c, int algo, const uint8_t* codons, const uint64_t* site_offsets, size_t n_blocks,
                      size_t leaf_stride, int compute_anc, uint32_t seed, float* out_phylo, float* out_anc)
{
    if (!c) return PCSF_ERR_INVALID_ARG;
    if (n_blocks == 0) return PCSF_OK;
    if (!site_offsets || !out_phylo || (algo != PCSF_MLE && algo != PCSF_FIXED && algo != PCSF_OMEGA))
        return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_score_blocks: bad argument");
    for (size_t b = 0; b < n_blocks; ++b)
        if (site_offsets[b + 1] < site_offsets[b]) return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_score_blocks: site_offsets must be non-decreasing");
    const size_t n_sites = site_offsets[n_blocks] - site_offsets[0];
    if (n_sites > 0 && (!codons || leaf_stride < site_offsets[n_blocks])) return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_score_blocks: bad codon matrix");
    hipError_t e;
    if ((e = hipSetDevice(c->device)) != hipSuccess) return c->hip_fail(e, "hipSetDevice");
    try {
        return pcsf::score_blocks_impl(c, algo, codons, site_offsets, n_blocks, leaf_stride, compute_anc, seed, out_phylo, out_anc);
    } catch (const std::exception& ex) {
        return c->fail(PCSF_ERR_HIP, std::string("pcsf_score_blocks: ") + ex.what());
    }
}

int pcsf_set_block_kernel(pcsf_ctx* c, int kernel)
{
    if (!c) return PCSF_ERR_INVALID_ARG;
    if (kernel != PCSF_BLOCK_KERNEL_AUTO && kernel != PCSF_BLOCK_KERNEL_TILES && kernel != PCSF_BLOCK_KERNEL_FACTORED)
        return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_set_block_kernel: bad argument");
    c->block_kernel = kernel;
    return PCSF_OK;
}

int pcsf_choose_block_kernel(int32_t n_leaves, const uint8_t* codons, const uint64_t* site_offsets, size_t n_blocks, size_t leaf_stride)
{
    if (n_leaves <= 0 || (n_blocks > 0 && !site_offsets)) return PCSF_ERR_INVALID_ARG;
    for (size_t b = 0; b < n_blocks; ++b)
        if (site_offsets[b + 1] < site_offsets[b]) return PCSF_ERR_INVALID_ARG;
    if (n_blocks > 0 && site_offsets[n_blocks] > site_offsets[0] && (!codons || leaf_stride < site_offsets[n_blocks])) return PCSF_ERR_INVALID_ARG;
    try {
        return pcsf::list_prefers_tiles(static_cast<size_t>(n_leaves), codons, leaf_stride, site_offsets, n_blocks) ? PCSF_BLOCK_KERNEL_TILES
                                                                                                                        : PCSF_BLOCK_KERNEL_FACTORED;
    } catch (...) {
        return PCSF_ERR_HIP;
    }
}

int pcsf_prepare_block_scoring(pcsf_ctx* c, int algo)
{
    if (!c) return PCSF_ERR_INVALID_ARG;
    if (algo != PCSF_MLE && algo != PCSF_FIXED && algo != PCSF_OMEGA) return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_prepare_block_scoring: bad argument");
    if (algo == PCSF_FIXED) return PCSF_OK;   // no engine, no streams of its own
    if (const char* v = std::getenv("PCSF_PREPARE_BLOCKS"))   // A/B knob: 0 = create the streams when an engine asks for them
        if (v[0] == '0') return PCSF_OK;
    c->block_streams.warm_up(c->device, 4);    // two groups of blocks, a main and a side stream each (score_blocks_impl)
    return PCSF_OK;
}

// log-likelihood of every block at caller-chosen parameters: ONE objective evaluation of the MLE / OMEGA
// strategies without the prior term (minimizer_lpr_leaves, src/fixed_lik.hpp:460-467; lpr_leaves_omega,
// src/omega.hpp:151-202).  model: 0 coding ECM, 1 non-coding ECM, 2 kappa/omega F3x4 model with
// q_settings[n_blocks][12] = kappa, omega, sigma, nine nucleotide-frequency ratios (src/run.hpp:75-79).
int pcsf_block_loglik(pcsf_ctx* c, int model, const double* q_settings, const double* tree_scale, const uint8_t* codons,
                      const uint64_t* site_offsets, size_t n_blocks, size_t leaf_stride, double* out_lpr, int32_t* out_failed)
{
    if (!c) return PCSF_ERR_INVALID_ARG;
    if (n_blocks == 0) return PCSF_OK;
    if (model < 0 || model > 2 || !tree_scale || !site_offsets || !out_lpr || (model == 2 && !q_settings))
        return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_block_loglik: bad argument");
    for (size_t b = 0; b < n_blocks; ++b)
        if (site_offsets[b + 1] < site_offsets[b] || !(tree_scale[b] >= 0.0)) return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_block_loglik: bad block");
    hipError_t e;
    if ((e = hipSetDevice(c->device)) != hipSuccess) return c->hip_fail(e, "hipSetDevice");
    try {
        return pcsf::block_loglik_impl(c, model, q_settings, tree_scale, codons, site_offsets, n_blocks, leaf_stride, out_lpr, out_failed);
    } catch (const std::exception& ex) {
        return c->fail(PCSF_ERR_HIP, std::string("pcsf_block_loglik: ") + ex.what());
    }
}

// parity tests: the same evaluation with caller-supplied eigen-systems (no eigen-solver of this library involved)
int pcsf_block_loglik_eig(pcsf_ctx* c, const double* right, const double* left, const double* lambda, const double* prior, size_t n_eig,
                          const double* tree_scale, const uint8_t* codons, const uint64_t* site_offsets, size_t n_blocks, size_t leaf_stride,
                          double* out_lpr, int32_t* out_failed)
{
    if (!c) return PCSF_ERR_INVALID_ARG;
    if (n_blocks == 0) return PCSF_OK;
    if (!right || !left || !lambda || !prior || !tree_scale || !site_offsets || !out_lpr || (n_eig != 1 && n_eig != n_blocks))
        return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_block_loglik_eig: bad argument (n_eig must be 1 or n_blocks)");
    for (size_t b = 0; b < n_blocks; ++b)
        if (site_offsets[b + 1] < site_offsets[b] || !(tree_scale[b] >= 0.0)) return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_block_loglik_eig: bad block");
    hipError_t e;
    if ((e = hipSetDevice(c->device)) != hipSuccess) return c->hip_fail(e, "hipSetDevice");
    try {
        std::vector<pcsf::EigenSystem> es(n_eig);
        for (size_t k = 0; k < n_eig; ++k) {
            std::memcpy(es[k].right.data(), right + k * 4096, 4096 * sizeof(double));
            std::memcpy(es[k].left.data(), left + k * 4096, 4096 * sizeof(double));
            std::memcpy(es[k].lambda.data(), lambda + k * 64, 64 * sizeof(double));
            std::memcpy(es[k].prior.data(), prior + k * 64, 64 * sizeof(double));
        }
        return pcsf::block_loglik_impl(c, 3, nullptr, tree_scale, codons, site_offsets, n_blocks, leaf_stride, out_lpr, out_failed, es.data(), n_eig);
    } catch (const std::exception& ex) {
        return c->fail(PCSF_ERR_HIP, std::string("pcsf_block_loglik_eig: ") + ex.what());
    }
}

// diagnostics/tests: the device eigen-solver of the kappa/omega rate matrix (eigen.hip)
int pcsf_debug_eigen_omega(pcsf_ctx* c, const double* q_settings, const double* warm_settings, size_t n, double* right, double* left,
                           double* lambda, double* prior, int32_t* failed)
{
    if (!c) return PCSF_ERR_INVALID_ARG;
    if (n == 0) return PCSF_OK;
    if (!q_settings || !right || !left || !lambda || !prior || !failed) return c->fail(PCSF_ERR_INVALID_ARG, "pcsf_debug_eigen_omega: bad argument");
    hipError_t e;
    if ((e = hipSetDevice(c->device)) != hipSuccess) return c->hip_fail(e, "hipSetDevice");
    try {
        return pcsf::debug_eigen_omega_impl(c, q_settings, warm_settings, n, right, left, lambda, prior, failed);
    } catch (const std::exception& ex) {
        return c->fail(PCSF_ERR_HIP, std::string("pcsf_debug_eigen_omega: ") + ex.what());
    }
}

}  // extern "C"
