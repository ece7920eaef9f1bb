// kernels.hip -- gfx950 (CDNA4) kernels of the PhyloCSF scoring core.
//
// prune_fixed<NW,G>: Felsenstein pruning + log-likelihood ratio for the FIXED strategy
// (reference: ensure_alpha / lpr_leaves / run_tracks, src/fixed_lik.hpp:125-164,362-449,
// src/run.hpp:35-55).  One wave owns G tiles of 16 alignment columns ("sites"); the partial
// likelihoods of a tile (64 states x 16 sites, fp64) live in 16 registers per lane in the
// operand layout of v_mfma_f64_16x16x4_f64, so a node's result feeds the next product without
// leaving the register file.  The 64x64 branch matrix of the edge being processed is staged
// once per work-group in LDS (fragment order, double-buffered); leaf edges are table look-ups
// in HBM/L2 (one 128-byte row segment per lane).  See DESIGN.md for layouts and the roofline.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <type_traits>

#include "kernels.h"

namespace pcsf {

typedef double double4_t __attribute__((ext_vector_type(4)));
typedef double double2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double4_t mfma_f64(double a, double b, double4_t c)
{
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// Leaf-message rows: table[leaf][code][q2][g][e] = P[4(2 q2 + e) + g][code].  A lane (site sl, group g) needs
// the sixteen states 4q+g; piece i (q = 2i, 2i+1) is 16 bytes at row + (i*4 + g)*16, so the four lanes of a site
// read 64 contiguous bytes per instruction and one wave-instruction touches 16 half cache lines.  (With the
// lane's 128 bytes stored contiguously every instruction touched 64 different lines and the L1 tag pipe, not
// the matrix pipe, set the pace of the leaf arms.)
struct LeafRow {
    double2_t v[8];
};
constexpr int ROW_PIECE_STRIDE = 4;  // in 16-byte units
#ifndef PCSF_STAGE_Q0
#define PCSF_STAGE_Q0 4               // the k-step whose slot 1 requests the first LDS-DMA piece of the next product's matrix (round 6: 4 instead of 0, +1 % on the table form; tools/cherry_ab.sh)
#endif
#ifndef PCSF_LEAF_ROW_Q0
#define PCSF_LEAF_ROW_Q0 4            // MM_LEAF: the k-step whose slot 2 requests the first piece of the leaf row (A/B builds: tools/build_variant.sh)
#endif

__device__ __forceinline__ const double2_t* leaf_row_ptr(const double* __restrict__ leaf_tab, int leaf, uint32_t code, int g4)
{
    return reinterpret_cast<const double2_t*>(leaf_tab + (static_cast<size_t>(leaf) * 65 + code) * 64) + g4;
}

// PAIRS: `code` is the row of the table set (leaves' rows, then the cherries' pair rows)
template <bool PAIRS>
__device__ __forceinline__ const double2_t* operand_row_ptr(const double* __restrict__ leaf_tab, int leaf, uint32_t code, int g4)
{
    if constexpr (PAIRS) return reinterpret_cast<const double2_t*>(leaf_tab + static_cast<size_t>(code) * 64) + g4;
    else return leaf_row_ptr(leaf_tab, leaf, code, g4);
}

__device__ __forceinline__ LeafRow load_leaf_row(const double* __restrict__ leaf_tab, int leaf, uint32_t code, int g4)
{
    const double2_t* r2 = leaf_row_ptr(leaf_tab, leaf, code, g4);
    LeafRow r;
#pragma unroll
    for (int i = 0; i < 8; ++i) r.v[i] = r2[i * ROW_PIECE_STRIDE];
    return r;
}

// acc[t] (t = 0..3, output states 16t..16t+15) <- P x cur, with P read from its LDS fragment image:
// 64 x v_mfma_f64_16x16x4_f64 in four independent accumulator chains.
//
// Issue model (measured with in-kernel stamps, tools/stamps.py): one MFMA occupies the matrix
// pipe for 64 cycles and a wave issues in order, so the next MFMA -- and everything behind it in
// the instruction stream -- waits for the pipe.  Anything placed AFTER a group of four MFMAs
// therefore starts ~200 cycles late and, if it takes longer than the 56 idle issue cycles of
// one MFMA, opens a bubble.  So every non-MFMA instruction of the product is slotted BETWEEN two
// MFMAs, at most one memory instruction per slot:
//   slot 0 (after MFMA 0): the two ds_read_b128 of the next k-step's fragments
//   slot 1..3: per_slot(q, j) -- LDS-DMA piece of the next slot, leaf-row loads, cherry multiplies
// Scheduling barriers pin this order (and keep the compiler from hoisting all 32 LDS reads).
template <class PerSlot>
__device__ __forceinline__ void product(const double2_t* __restrict__ buf, int lane, const double (&cur)[16], double4_t (&acc)[4],
                                        PerSlot&& per_slot)
{
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = double4_t{0.0, 0.0, 0.0, 0.0};
    double2_t n01 = buf[lane], n23 = buf[64 + lane];
    auto step = [&](auto qc) {
        constexpr int q = decltype(qc)::value;
        const double2_t f01 = n01, f23 = n23;
        __builtin_amdgcn_sched_barrier(0);
        acc[0] = mfma_f64(f01.x, cur[q], acc[0]);
        __builtin_amdgcn_sched_barrier(0);
        if (q + 1 < 16) {
            n01 = buf[((q + 1) * 2 + 0) * 64 + lane];
            n23 = buf[((q + 1) * 2 + 1) * 64 + lane];
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[1] = mfma_f64(f01.y, cur[q], acc[1]);
        __builtin_amdgcn_sched_barrier(0);
        per_slot(qc, std::integral_constant<int, 1>{});
        __builtin_amdgcn_sched_barrier(0);
        acc[2] = mfma_f64(f23.x, cur[q], acc[2]);
        __builtin_amdgcn_sched_barrier(0);
        per_slot(qc, std::integral_constant<int, 2>{});
        __builtin_amdgcn_sched_barrier(0);
        acc[3] = mfma_f64(f23.y, cur[q], acc[3]);
        __builtin_amdgcn_sched_barrier(0);
        per_slot(qc, std::integral_constant<int, 3>{});
        __builtin_amdgcn_sched_barrier(0);
    };
#define PCSF_STEP(Q) step(std::integral_constant<int, Q>{});
    PCSF_STEP(0) PCSF_STEP(1) PCSF_STEP(2) PCSF_STEP(3) PCSF_STEP(4) PCSF_STEP(5) PCSF_STEP(6) PCSF_STEP(7)
    PCSF_STEP(8) PCSF_STEP(9) PCSF_STEP(10) PCSF_STEP(11) PCSF_STEP(12) PCSF_STEP(13) PCSF_STEP(14) PCSF_STEP(15)
#undef PCSF_STEP
}

// prune_fixed<NW>: NW waves per work-group, 16 sites per wave.
// LDS (dynamic, one array): [2][4096] doubles fragment double-buffer, [n_ops] program words,
// then [n_leaves][NW*16] codon bytes.  The program is read from LDS, not with scalar loads: an
// s_load in flight makes every lgkmcnt wait of the product a full drain (SMEM returns out of
// order), which stalls each k-step on the fragment reads it has just issued.
// STAMP: diagnostic build only (variant 2): wave 0 of one work-group records s_memtime at the
// start of every op, after its barrier, after its product and after its epilogue into p.stamps.
// Stamps never touch an output value; the shipped variants are compiled with STAMP = false.
__device__ __forceinline__ unsigned long long stamp_now()
{
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    return t;
}

// BLOCKS: per-work-group site window and table set (whole-block scoring with per-block branch
// matrices, MLE / OMEGA); otherwise work-group b owns sites [b*SITES, (b+1)*SITES) and passes
// 0..n_passes-1 use table sets 0..n_passes-1 (coding, non-coding).
// FUSED (tracks mode): the codon columns are not read from a codon matrix but formed here from the resident
// nucleotide rows -- the six-frame front end (alignment_t::update_seqs / translate and the reverse complement between
// the strands, src/parallel_file_reader.hpp:61-113, src/phylocsf++build_tracks.hpp:162-227) runs in the prologue of the
// work-group that scores the columns, and the leaf-major codon matrix never exists in HBM.
// DEEP: the tree needs more than MAX_STACK_DEPTH_DEV parked tiles (no tree the reference ships does): pushes and pops
// nest, so the kernel keeps a stack pointer of its own and parks the tiles in scratch memory.
// SKIP (tracks mode, trees of at most 128 leaves): subtrees that are gaps in all of the work-group's columns cost nothing.  Real
// multi-species alignments leave most species out of most blocks (they reach the kernel as N-padded rows, codon 64).  The message of
// an all-gap subtree does not depend on the column -- it is what the full program computes from the tables' row-sum rows, a vector
// close to 1 (src/instance.hpp:625-639, src/fixed_lik.hpp:111-117) -- so the context records it once per edge and model with the
// full kernel itself (p.gap_dump, diagnostic variant) and the work-group compacts the program in its prologue: ops whose result is
// an all-gap message and that do not touch the stack of parked tiles disappear, and any op above an all-gap subtree takes the recorded
// message instead of its product (`gap_before`).  Every score is BIT-IDENTICAL to the full program's: the recorded vectors are the
// full program's own values.

// PAIRS (tracks mode, the context's own branch lengths): cherry tables.  A cherry's message to its parent depends on the column only
// through its two codons: the context holds it per codon pair (pair_table_kernel below made the rows with this kernel's own product), the
// program is the one of the tree whose cherries are leaves, and a leaf operand is a ROW of the table set -- leaf l: row 65 l + codon; cherry
// p: row 65 n_leaves + 4225 p + 65 a + b.  One product less per cherry and model; the same bits as the full program.
template <int NW, bool STAMP, bool BLOCKS, bool FUSED, bool DEEP, bool SKIP, bool PAIRS>
__global__ __launch_bounds__(NW * 64, 2) void prune_fixed(const PruneParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    double2_t* lfrag = reinterpret_cast<double2_t*>(lds_raw);
    uint32_t* lprog = reinterpret_cast<uint32_t*>(lds_raw + 2 * FRAG_DOUBLES_DEV * sizeof(double));
    unsigned char* lcod = reinterpret_cast<unsigned char*>(lprog + ((p.n_ops + 3) & ~3));

    constexpr int SITES = NW * 16;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int sl = lane & 15, g4 = lane >> 4;
    const size_t block_site0 = BLOCKS ? p.wg_site0[blockIdx.x] : static_cast<size_t>(blockIdx.x) * SITES;
    const size_t site_end = BLOCKS ? p.wg_site_end[blockIdx.x] : p.n_sites;
    const int table0 = BLOCKS ? static_cast<int>(p.wg_table[blockIdx.x]) : 0;
    const int prior0 = (BLOCKS && p.wg_prior) ? static_cast<int>(p.wg_prior[blockIdx.x]) : table0;
    const size_t my_site = block_site0 + wave * 16 + sl;
    const bool valid = my_site < site_end;
    const unsigned long long stamp_t0 = STAMP ? stamp_now() : 0ull;
    (void)stamp_t0;

    // ---- fragment staging: 32 KiB per slot, moved by LDS-DMA in 1 KiB wave-instructions
    constexpr int CHUNKS_PER_WAVE = 32 / NW;
    const int total_slots = p.n_passes * p.n_edges;  // pass 0 slots, then pass 1 slots (consecutive table sets)
    // each wave moves CHUNKS_PER_WAVE consecutive 1 KiB pieces; the instruction's immediate offset
    // advances the global and the LDS address together, so one address register serves 4 pieces
    const unsigned char* src_lane = reinterpret_cast<const unsigned char*>(p.frag) +
                                    static_cast<size_t>(table0) * p.n_edges * (FRAG_DOUBLES_DEV * 8) + wave * (CHUNKS_PER_WAVE * 1024) + lane * 16;
    // stage_piece<J>(slot): J-th of this wave's CHUNKS_PER_WAVE pieces of `slot`
    auto stage_piece = [&](auto jc, int slot, int buffer) {
        constexpr int J = decltype(jc)::value;
        const unsigned char* src = src_lane + static_cast<size_t>(slot) * (FRAG_DOUBLES_DEV * 8) + (J / 4) * 4096;
        unsigned char* dst = lds_raw + buffer * (FRAG_DOUBLES_DEV * 8) + wave * (CHUNKS_PER_WAVE * 1024) + (J / 4) * 4096;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)dst, 16, (J % 4) * 1024, 0);
    };
    auto stage_slot = [&](int slot) {
        stage_piece(std::integral_constant<int, 0>{}, slot, slot & 1);
        stage_piece(std::integral_constant<int, 1>{}, slot, slot & 1);
        stage_piece(std::integral_constant<int, 2>{}, slot, slot & 1);
        stage_piece(std::integral_constant<int, 3>{}, slot, slot & 1);
        if constexpr (CHUNKS_PER_WAVE > 4) {
            stage_piece(std::integral_constant<int, 4>{}, slot, slot & 1);
            stage_piece(std::integral_constant<int, 5>{}, slot, slot & 1);
            stage_piece(std::integral_constant<int, 6>{}, slot, slot & 1);
            stage_piece(std::integral_constant<int, 7>{}, slot, slot & 1);
        }
    };
    // (the first product's matrix is requested before anything else: its 32 KiB arrive while the codons are staged -- the skipping variants
    // only know their first product after the compaction below)
    if constexpr (!SKIP) {
        if (total_slots > 0) stage_slot(0);
    }
    // ---- stage this work-group's codon columns into LDS (clamped to 0..64; 64 beyond the end)
    // A full, 16-byte aligned window moves as one 16-byte load per thread (58 leaves x 64 sites = 232 pieces):
    // the byte-wise loop below costs ~15 dependent round trips per work-group while the CU's other group runs alone.
    const bool whole_window = !FUSED && block_site0 + SITES <= site_end && ((block_site0 | p.leaf_stride | reinterpret_cast<size_t>(p.codons)) & 15) == 0;
    if constexpr (FUSED) {
        // phase 1, one thread per site: which (alignment, strand, frame) segment the site belongs to and where its codon
        // starts.  Segments follow each other in the site stream; the host gives the segment of the window's first site.
        // Site i of a '-' segment holds codon n-1-i (scores come out in forward genome order); its three columns are read
        // right to left and complemented.
        unsigned char* lut = lcod + ((p.n_leaves * SITES + 15) & ~15);                   // [256] character classes
        long long* lfirst = reinterpret_cast<long long*>(lut + 256);                     // [SITES] first column of the codon, -1: no site
        unsigned char* lminus = reinterpret_cast<unsigned char*>(lfirst + SITES);        // [SITES]
        for (int ch = tid; ch < 256; ch += NW * 64) {
            unsigned v = 5u;                                                             // a character the reference exits on
            switch (ch) {
                case 'A': case 'a': v = 0u; break;
                case 'C': case 'c': v = 1u; break;
                case 'G': case 'g': v = 2u; break;
                case 'T': case 't': v = 3u; break;
                case 'N': case 'n': case '-': case '.': v = 4u; break;
                default: break;
            }
            lut[ch] = static_cast<unsigned char>(v);
        }
        if (tid < SITES) {
            const size_t site = block_site0 + tid;
            long long first = -1;
            unsigned minus = 0;
            if (site < site_end) {
                uint32_t k = p.wg_first_seg[blockIdx.x];
                FrameSegment sg = p.segments[k];
                while (site >= sg.dst0 + sg.n_codons) sg = p.segments[++k];
                const size_t i = site - sg.dst0;
                minus = sg.minus;
                const size_t c = minus ? sg.n_codons - 1 - i : i;
                first = static_cast<long long>(minus ? sg.col0 + sg.len - 1 - sg.skip - 3 * c : sg.col0 + sg.skip + 3 * c);
            }
            lfirst[tid] = first;
            lminus[tid] = static_cast<unsigned char>(minus);
        }
        __syncthreads();
        // phase 2: (leaf, site) pairs; consecutive threads read consecutive codons of one species row
        for (int idx = tid; idx < p.n_leaves * SITES; idx += NW * 64) {
            const int leaf = idx / SITES, s = idx - leaf * SITES;
            const long long first = lfirst[s];
            unsigned cod = 64u;
            if (first >= 0) {
                const unsigned char* row = p.nucl + static_cast<size_t>(leaf) * p.nucl_stride;
                const long long step = lminus[s] ? -1 : 1;
                unsigned a = lut[row[first]], b = lut[row[first + step]], d = lut[row[first + 2 * step]];
                if ((a == 5u || b == 5u || d == 5u) && p.first_bad) {  // outside ACGTNacgtn-.: the call fails with the character's position
                    const long long where = a == 5u ? first : (b == 5u ? first + step : first + 2 * step);
                    atomicMin(p.first_bad, static_cast<unsigned long long>(static_cast<size_t>(leaf) * p.nucl_stride + static_cast<size_t>(where)));
                }
                if (lminus[s]) {  // complement: A<->T, C<->G
                    a = a < 4u ? 3u - a : a;
                    b = b < 4u ? 3u - b : b;
                    d = d < 4u ? 3u - d : d;
                }
                cod = (a >= 4u || b >= 4u || d >= 4u) ? 64u : 16u * a + 4u * b + d;
            }
            lcod[idx] = static_cast<unsigned char>(cod);
            if (p.dump_codons && first >= 0) p.dump_codons[static_cast<size_t>(leaf) * p.leaf_stride + block_site0 + s] = static_cast<unsigned char>(cod);
        }
    } else if (whole_window) {
        constexpr int PIECES = SITES / 16;
        for (int idx = tid; idx < p.n_leaves * PIECES; idx += NW * 64) {
            const int leaf = idx / PIECES, piece = idx % PIECES;
            uint4 v = *reinterpret_cast<const uint4*>(p.codons + static_cast<size_t>(leaf) * p.leaf_stride + block_site0 + piece * 16);
            auto clamp64 = [](uint32_t x) {  // per byte: min(x, 64)
                const uint32_t over = (((x & 0x7f7f7f7fu) + 0x3f3f3f3fu) | x) & 0x80808080u;  // bit 7 of each byte > 64
                const uint32_t mask = (over >> 7) * 0xffu;
                return (x & ~mask) | (0x40404040u & mask);
            };
            v.x = clamp64(v.x); v.y = clamp64(v.y); v.z = clamp64(v.z); v.w = clamp64(v.w);
            *reinterpret_cast<uint4*>(lcod + leaf * SITES + piece * 16) = v;
        }
    } else {
        for (int idx = tid; idx < p.n_leaves * SITES; idx += NW * 64) {
            const int leaf = idx / SITES, s = idx - leaf * SITES;
            const size_t site = block_site0 + s;
            unsigned c = 64u;
            if (site < site_end) {
                c = p.codons[static_cast<size_t>(leaf) * p.leaf_stride + site];
                c = c > 64u ? 64u : c;
            }
            lcod[idx] = static_cast<unsigned char>(c);
        }
    }
    for (int i = tid; i < p.n_ops; i += NW * 64) lprog[i] = p.program[i];
    const unsigned char* my_cod = lcod + wave * 16 + sl;  // + leaf*SITES
    __syncthreads();
    // PAIRS: the pair code 65 a + b of every cherry that is an operand, per site; behind them the ROW of every triple and site
    uint16_t* lpair = reinterpret_cast<uint16_t*>(lds_raw + p.pair_lds_offset);
    const uint16_t* my_pair = lpair + wave * 16 + sl;     // + pair*SITES
    uint32_t* ltrip = reinterpret_cast<uint32_t*>(lds_raw + p.pair_lds_offset + ((static_cast<uint32_t>(PAIRS ? p.n_free_pairs : 0) * SITES * 2 + 15) & ~15u));
    const uint32_t* my_trip = ltrip + wave * 16 + sl;     // + triple*SITES; the quads' rows follow the free triples' (operand QUAD_BASE + q = entry n_free_triples + q)
    (void)my_pair;
    (void)my_trip;
    if constexpr (PAIRS) {
        static_assert(!BLOCKS, "per-block table sets have no pair rows");
        // (SITES = the wave's 64 lanes: entry idx = tid + 256 k belongs to table wave + 4 k -- wave-uniform, so its leaves come by scalar loads)
        static_assert(SITES == 64, "one table operand per wave and pass");
        for (int pp = wave; pp < p.n_free_pairs; pp += NW) {
            const int pu = __builtin_amdgcn_readfirstlane(pp);
            const unsigned la = p.pair_leaves[2 * pu], lb = p.pair_leaves[2 * pu + 1];
            lpair[pu * SITES + lane] = static_cast<uint16_t>(lcod[la * SITES + lane] * 65u + lcod[lb * SITES + lane]);
        }
        for (int t = wave; t < p.n_free_triples; t += NW) {
            const int tu = __builtin_amdgcn_readfirstlane(t);
            const uint32_t ab = p.trip_info[3 * tu], lc = p.trip_info[3 * tu + 1], base = p.trip_info[3 * tu + 2];
            ltrip[tu * SITES + lane] = base + (lcod[(ab & 0xffffu) * SITES + lane] * 65u + lcod[(ab >> 16) * SITES + lane]) * 65u + lcod[lc * SITES + lane];
        }
        for (int q = wave; q < p.n_quads; q += NW) {
            const int qu = __builtin_amdgcn_readfirstlane(q);
            const uint32_t ab = p.quad_info[3 * qu], cd = p.quad_info[3 * qu + 1], base = p.quad_info[3 * qu + 2];
            const uint32_t code = ((lcod[(ab & 0xffffu) * SITES + lane] * 65u + lcod[(ab >> 16) * SITES + lane]) * 65u + lcod[(cd & 0xffffu) * SITES + lane]) * 65u + lcod[(cd >> 16) * SITES + lane];
            ltrip[(p.n_free_triples + qu) * SITES + lane] = base + code;
        }
        if constexpr (!SKIP) __syncthreads();   // (SKIP: its own barriers below come before the first use)
    }
    // a leaf operand of an op word -> what the row look-up takes: the codon, or with PAIRS the row itself
    auto operand_row = [&](uint32_t x) __attribute__((always_inline)) -> uint32_t {
        if constexpr (!PAIRS) {
            return my_cod[x * SITES];
        } else {
#ifdef PCSF_OPERAND_BRANCHFREE
            // all three look-ups issued, the operand's kind picks one: no scalar branch in the op loop (indices clamped into their arrays)
            const uint32_t kind = x >> 12, li = x < PAIR_OPERAND_BASE_DEV ? x : 0u, pi = kind == 2u ? x - PAIR_OPERAND_BASE_DEV : 0u,
                           ti = kind == 3u ? (x < QUAD_OPERAND_BASE_DEV ? x - TRIPLE_OPERAND_BASE_DEV : static_cast<uint32_t>(p.n_free_triples) + x - QUAD_OPERAND_BASE_DEV) : 0u;
            const uint32_t c8 = my_cod[li * SITES], c16 = my_pair[pi * SITES], c32 = my_trip[ti * SITES];
            const uint32_t r_leaf = li * 65u + c8, r_pair = static_cast<uint32_t>(p.n_leaves) * 65u + pi * PAIR_CODES_DEV + c16;
            return kind < 2u ? r_leaf : (kind == 2u ? r_pair : c32);
#else
            if (x < PAIR_OPERAND_BASE_DEV) return x * 65u + my_cod[x * SITES];
            if (x < TRIPLE_OPERAND_BASE_DEV) {
                const uint32_t pp = x - PAIR_OPERAND_BASE_DEV;
                return static_cast<uint32_t>(p.n_leaves) * 65u + pp * PAIR_CODES_DEV + my_pair[pp * SITES];
            }
            if (x < QUAD_OPERAND_BASE_DEV) return my_trip[(x - TRIPLE_OPERAND_BASE_DEV) * SITES];
            return my_trip[(static_cast<uint32_t>(p.n_free_triples) + x - QUAD_OPERAND_BASE_DEV) * SITES];
#endif
        }
    };
    // SKIP: the work-group's own program (see above), built by wave 0 from the leaves that have a codon in any of its columns
    __shared__ unsigned long long lhas[2];
    __shared__ uint32_t lprog2[SKIP ? 128 : 1];
    __shared__ uint16_t lslot2[SKIP ? 128 : 1], lnext2[SKIP ? 128 : 1];   // table slot of the op's product; of the NEXT product (bit 15: in the next pass)
    __shared__ uint32_t linfo2[SKIP ? 128 : 1];                         // lslot2 | lnext2 << 16: what the op loop fetches one op ahead, with the op word
    __shared__ int lcount2[2];                                           // ops, first product's table slot (-1: no product at all)
    if constexpr (SKIP) {
        static_assert(NW == 4, "the presence pass reads 64-byte codon rows");
        if (wave < 2) {
            const int leaf = wave * 64 + lane;
            bool has = false;
            if (leaf < p.n_leaves) {
                const uint4* row = reinterpret_cast<const uint4*>(lcod + leaf * SITES);
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const uint4 w = row[u];
                    has |= (w.x != 0x40404040u) | (w.y != 0x40404040u) | (w.z != 0x40404040u) | (w.w != 0x40404040u);
                }
            }
            const unsigned long long m = __ballot(has);
            if (lane == 0) lhas[wave] = m;
        }
        __syncthreads();
        if (wave == 0) {
            const unsigned long long has0 = lhas[0], has1 = lhas[1];
            const int n_ops = p.n_ops;
            uint32_t opw[2], mt[2];
            unsigned long long ones_after[2], active[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int j = 64 * k + lane;
                const bool valid = j < n_ops;
                opw[k] = valid ? lprog[j] : (F_ROOT_DEV << 28);
                mt[k] = valid ? p.op_meta[j] : 0u;
                const unsigned long long l0 = valid ? p.op_leaves[2 * j] : 0ull, l1 = valid ? p.op_leaves[2 * j + 1] : 0ull;
                ones_after[k] = __ballot(valid && ((l0 & has0) | (l1 & has1)) == 0ull);
            }
            uint32_t neww[2];
            bool prod[2], gapb[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int j = 64 * k + lane;
                const uint32_t code = opw[k] >> 28, args = opw[k] & 0x0fffffffu;
                const bool is_push = code >= F_MM_PUSH_CHERRY0_DEV && code < F_MM_POP0_DEV, is_pop = code >= F_MM_POP0_DEV && code < F_ROOT_DEV;
                const bool is_root = code >= F_ROOT_DEV;
                const bool after = (ones_after[k] >> lane) & 1ull;
                const bool before = lane > 0 ? ((ones_after[k] >> (lane - 1)) & 1ull) : (k == 1 && (ones_after[0] >> 63));   // op j - 1 left all ones
                // Only ops without stack traffic disappear: a push or pop above all-gap subtrees costs no product either way (it
                // takes the recorded message), and keeping them keeps the stack of parked tiles exactly as the full program has it.
                const bool trivial = j < n_ops && after && !is_root && !is_push && !is_pop;
                active[k] = __ballot(j < n_ops && !trivial);
                neww[k] = opw[k];
                (void)args;
                gapb[k] = before && code != F_CHERRY_DEV;
                prod[k] = !gapb[k] && code != F_CHERRY_DEV && !is_root;
            }
            const int n0 = __popcll(active[0]), n2 = n0 + __popcll(active[1]);
            const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
            for (int k = 0; k < 2; ++k)
                if ((active[k] >> lane) & 1ull) {
                    const int pos = (k ? n0 : 0) + __popcll(active[k] & below);
                    lprog2[pos] = neww[k];
                    lslot2[pos] = static_cast<uint16_t>(mt[k] & 0x3fffu) | (prod[k] ? 0x8000u : 0u) | (gapb[k] ? 0x4000u : 0u);
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // the product that follows every op (its matrix is staged while the op's own product runs)
            unsigned long long pc[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) pc[k] = __ballot(64 * k + lane < n2 && (lslot2[64 * k + lane < n2 ? 64 * k + lane : 0] & 0x8000u));
            const int first = pc[0] ? __builtin_ctzll(pc[0]) : (pc[1] ? 64 + __builtin_ctzll(pc[1]) : -1);
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int j = 64 * k + lane;
                if (j < n2) {
                    const unsigned long long above0 = k == 0 ? (lane < 63 ? pc[0] & ~((2ull << lane) - 1ull) : 0ull) : 0ull;
                    const unsigned long long above1 = k == 0 ? pc[1] : (lane < 63 ? pc[1] & ~((2ull << lane) - 1ull) : 0ull);
                    const int nx = above0 ? __builtin_ctzll(above0) : (above1 ? 64 + __builtin_ctzll(above1) : -1);
                    lnext2[j] = nx >= 0 ? static_cast<uint16_t>(lslot2[nx] & 0x3fffu) : static_cast<uint16_t>((first >= 0 ? (lslot2[first] & 0x3fffu) : 0u) | 0x8000u);
                    linfo2[j] = static_cast<uint32_t>(lslot2[j]) | (static_cast<uint32_t>(lnext2[j]) << 16);
                }
            }
            if (lane == 0) {
                lcount2[0] = n2;
                lcount2[1] = first >= 0 ? static_cast<int>(lslot2[first] & 0x3fffu) : -1;
            }
        }
        __syncthreads();
    }
    const uint32_t* const prog = SKIP ? lprog2 : lprog;
    const int n_run = SKIP ? lcount2[0] : p.n_ops;

    int slot = 0;   // products done so far: the LDS buffer a product reads is slot & 1 (SKIP: the table slot comes from lslot2)
    if constexpr (SKIP) {
        if (lcount2[1] >= 0 && total_slots > 0) {
            stage_piece(std::integral_constant<int, 0>{}, lcount2[1], 0);
            stage_piece(std::integral_constant<int, 1>{}, lcount2[1], 0);
            stage_piece(std::integral_constant<int, 2>{}, lcount2[1], 0);
            stage_piece(std::integral_constant<int, 3>{}, lcount2[1], 0);
            static_assert(!SKIP || CHUNKS_PER_WAVE == 8, "four waves move eight pieces each");
            stage_piece(std::integral_constant<int, 4>{}, lcount2[1], 0);
            stage_piece(std::integral_constant<int, 5>{}, lcount2[1], 0);
            stage_piece(std::integral_constant<int, 6>{}, lcount2[1], 0);
            stage_piece(std::integral_constant<int, 7>{}, lcount2[1], 0);
        }
    }
    // Two work-groups share a CU (one wave each per SIMD), run the same program and start together,
    // so they would reach their matrix phases and their between-product gaps at the same time and
    // leave the pipe idle in every gap.  The first generation of work-groups is staggered by
    // their hardware wave slot (HW_ID.WAVE_ID): the group in the odd slot starts half a product
    // late.  Later work-groups inherit the phase of the group they replace.  Speed only.
    if (p.stagger_sleep > 0 && blockIdx.x < p.stagger_first_blocks && wave == 0) {
        const unsigned wave_slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);  // HW_REG_HW_ID[3:0]
        if (wave_slot & 1)
            for (int i = 0; i < p.stagger_sleep; ++i) __builtin_amdgcn_s_sleep(16);  // 16 x 64 cycles
    }
    if (p.prio_mode != 0) {  // experiment: asymmetric instruction priority of the two waves that share a SIMD
        const unsigned wave_slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);  // HW_REG_HW_ID[3:0] = wave slot
        const bool odd = (wave_slot & 1) != 0;
        if (p.prio_mode == 1) {
            if (odd) __builtin_amdgcn_s_setprio(0);
            else __builtin_amdgcn_s_setprio(3);
        } else if (p.prio_mode == 2) {
            if (odd) __builtin_amdgcn_s_setprio(3);
            else __builtin_amdgcn_s_setprio(0);
        } else if (p.prio_mode == 3) {  // by work-group parity instead of the hardware slot
            if (blockIdx.x & 1) __builtin_amdgcn_s_setprio(0);
            else __builtin_amdgcn_s_setprio(3);
        }
    }
    double lnz0 = 0.0, lnz1 = 0.0, anc0 = 0.0, anc1 = 0.0;
    const bool stamping = STAMP && p.stamps != nullptr && blockIdx.x == p.stamp_block && wave == 0;
    (void)stamping;
    // diagnostic build: per-segment cycle sums, kept in scalar registers until the end of the kernel
    unsigned long long seg_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, stamp_prev = stamp_t0;
    unsigned seg_cnt[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    (void)seg_sum; (void)seg_cnt; (void)stamp_prev;

    if constexpr (STAMP) {  // segment 6: work-group prologue (codons + program into LDS, first slot requested)
        seg_sum[6] = stamp_now() - stamp_prev;
        seg_cnt[6] = 1;
        stamp_prev += seg_sum[6];
    }
    // recorded all-gap messages (SKIP): row `pass * (n_edges + 1) + slot`, slot n_edges = the root's partial likelihoods; [g][q] lane order
    auto gap_row_ptr = [&](int row) { return reinterpret_cast<const double2_t*>(p.gap_tab + static_cast<size_t>(row) * 64 + g4 * 16); };
    auto load_gap_row = [&](int row) {
        const double2_t* r2 = gap_row_ptr(row);
        LeafRow r;
#pragma unroll
        for (int i = 0; i < 8; ++i) r.v[i] = r2[i];
        return r;
    };
    (void)load_gap_row;
    // ... straight into the accumulator layout of a product's result: acc[t][r] = state 4 (4t + r) + g4
    auto gap_message = [&](int row, double4_t (&dst)[4]) {
        const double2_t* r2 = gap_row_ptr(row);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const double2_t lo = r2[2 * t], hi = r2[2 * t + 1];
            dst[t] = double4_t{lo.x, lo.y, hi.x, hi.y};
        }
    };
    (void)gap_message;
    for (int model = 0; model < p.n_passes; ++model) {
        const double* __restrict__ leaf_tab = p.leaf_tab + (PAIRS ? static_cast<size_t>(table0 + model) * p.leaf_rows * 64
                                                                  : static_cast<size_t>(table0 + model) * p.n_leaves * LEAF_TABLE_DOUBLES_DEV);
        const double* __restrict__ prior = p.prior + static_cast<size_t>(prior0 + model) * 64 + g4 * 16;

        double cur[16];
        double4_t acc[4];
        double4_t stk[MAX_STACK_DEPTH_DEV][4];  // parked messages, in accumulator layout: a push accumulates in place
        double4_t deep_stk[DEEP ? MAX_STACK_DEPTH_DEEP_DEV : 1][4];  // DEEP: dynamically indexed -> scratch memory
        int sp = 0;
        (void)deep_stk; (void)sp;

        // The op word and the codon bytes of op i+1 are fetched from LDS while op i runs, so that no LDS
        // round trip (op word -> leaf index -> codon byte) sits between two products.
#define PCSF_STAMP_RAW(K)                                                            \
    if constexpr (STAMP) {                                                           \
        __builtin_amdgcn_sched_barrier(0);                                           \
        const unsigned long long ts_ = stamp_now();                                  \
        __builtin_amdgcn_sched_barrier(0);                                           \
        seg_sum[K] += ts_ - stamp_prev;                                              \
        seg_cnt[K] += 1;                                                             \
        stamp_prev = ts_;                                                            \
    }
        uint32_t op = __builtin_amdgcn_readfirstlane(prog[0]);
        uint32_t word_ahead = prog[n_run > 1 ? 1 : 0];
        uint32_t info = SKIP ? __builtin_amdgcn_readfirstlane(linfo2[0]) : 0u, info_ahead = SKIP ? linfo2[n_run > 1 ? 1 : 0] : 0u;
        (void)info_ahead;
        uint32_t ca = operand_row((op >> 14) & 0x3fff), cb = operand_row(op & 0x3fff);
        for (int ip = 0; ip < n_run; ++ip) {
            PCSF_STAMP_RAW(7)
            const uint32_t code = op >> 28;
            const int a = (op >> 14) & 0x3fff, b = op & 0x3fff;
            const uint32_t cod_a = ca, cod_b = cb;
            (void)cod_b;
            // The next op's word and codon bytes are fetched from LDS while this op runs: inside its product (slot 3 of
            // k-step 13, PCSF_STAGE_SLOT) for the product arms, right here for the two arms without a product
#define PCSF_FETCH_NEXT_OP()                                                         \
    {                                                                                \
        op = __builtin_amdgcn_readfirstlane(word_ahead);                             \
        word_ahead = prog[ip + 2 < n_run ? ip + 2 : n_run - 1];                      \
        if constexpr (SKIP) {                                                        \
            info = __builtin_amdgcn_readfirstlane(info_ahead);                       \
            info_ahead = linfo2[ip + 2 < n_run ? ip + 2 : n_run - 1];                \
        }                                                                            \
        ca = operand_row((op >> 14) & 0x3fff);                                       \
        cb = operand_row(op & 0x3fff);                                               \
    }
            // SKIP: table slot of the op's edge | 0x8000: the op runs a product | 0x4000: the walk's message before the op is all gaps
            const uint32_t op_flags = info & 0xffffu, op_next = info >> 16;   // of THIS op: `info` moves on to the next op with `op`
            (void)op_next;
            const bool gap_before = SKIP && (op_flags & 0x4000u);
            if (SKIP ? !(op_flags & 0x8000u) : (code == F_CHERRY_DEV || code >= F_ROOT_DEV)) PCSF_FETCH_NEXT_OP()   // the arms without a product
            PCSF_STAMP_RAW(5)

            // Every op except the first cherry and the root starts with a product.  Before it: (a) this wave's
            // LDS-DMA of the current slot has landed (explicit vmcnt(0), see PCSF_PRODUCT_PROLOGUE), then the barrier
            // (b) publishes all waves' parts and (c) retires all reads of the other buffer, which the next slot's
            // DMA overwrites.
// segment K ends here: add (now - previous stamp) to its scalar sum; nothing is stored inside the loop
// (a store would put its own latency behind the next vmcnt wait and distort what is measured)
#define PCSF_STAMP(K)                                                                \
    if constexpr (STAMP) {                                                           \
        __builtin_amdgcn_sched_barrier(0);                                           \
        const unsigned long long ts_ = stamp_now();                                  \
        __builtin_amdgcn_sched_barrier(0);                                           \
        seg_sum[K] += ts_ - stamp_prev;                                              \
        seg_cnt[K] += 1;                                                             \
        stamp_prev = ts_;                                                            \
    }
#define PCSF_PRODUCT_PROLOGUE()                                                      \
    PCSF_STAMP(0)                                                                    \
    /* this wave's LDS-DMA pieces of the current slot must have LANDED before the barrier publishes them.  The barrier does \
       not do that: __syncthreads() is a work-group fence, which waits for LDS operations (lgkmcnt) but not for vector \
       memory (vmcnt).  Ops with leaf rows wait for them in their epilogue (in-order return covers the DMA issued earlier); \
       after a POP nothing did, and a product could read pieces of the matrix two slots back -- seen only under memory \
       pressure (fused front end: one work-group in ~30,000).  The pieces were requested a product ago: normally free. */ \
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                \
    __syncthreads();                                                                 \
    PCSF_STAMP(1)                                                                    \
    const double2_t* buf = lfrag + (slot & 1) * (FRAG_DOUBLES_DEV / 2);              \
    const int this_slot = slot;                                                      \
    /* branch-free: the last product re-stages its own slot into the idle buffer (never read) */ \
    int stage_next = this_slot + 1 < total_slots ? this_slot + 1 : this_slot;        \
    if constexpr (SKIP) {                                                            \
        const uint32_t nx_ = op_next;                                                \
        const int pass_ = (nx_ & 0x8000u) ? model + 1 : model;                       \
        stage_next = pass_ < p.n_passes ? pass_ * p.n_edges + static_cast<int>(nx_ & 0x3fffu) : model * p.n_edges + static_cast<int>(op_flags & 0x3fffu); \
    }                                                                                \
    ++slot;
// slot 1 of k-steps 0..CHUNKS_PER_WAVE-1: one LDS-DMA piece of the next slot each
#define PCSF_STAGE_SLOT(q, j)                                                        \
    if constexpr (j == 1 && q >= PCSF_STAGE_Q0 && q < PCSF_STAGE_Q0 + CHUNKS_PER_WAVE) { \
        stage_piece(std::integral_constant<int, (q >= PCSF_STAGE_Q0 && q < PCSF_STAGE_Q0 + CHUNKS_PER_WAVE ? q - PCSF_STAGE_Q0 : 0)>{}, stage_next, (this_slot + 1) & 1); \
    }                                                                                \
    if constexpr (j == 3 && q == 13) PCSF_FETCH_NEXT_OP()
// Cherry rows: piece i of row a in k-step i slot 2, of row b in k-step i slot 3 (i = 0..7); multiplied after the
// product, by which time the last piece has been on its way for eight k-steps.
/* diagnostic variant only: the message of every edge (and the root vector) of work-group 0's first column into p.gap_dump -- run once \
   per context on an all-gap column, this records what SKIP substitutes for the products above all-gap subtrees */ \
#define PCSF_DUMP(A)                                                                 \
    if constexpr (STAMP) {                                                           \
        if (p.gap_dump && blockIdx.x == 0 && wave == 0 && sl == 0) {                 \
            double* d_ = p.gap_dump + (static_cast<size_t>(model) * (p.n_edges + 1) + (this_slot - model * p.n_edges)) * 64 + g4 * 16; \
            _Pragma("unroll") for (int q_ = 0; q_ < 16; ++q_) d_[q_] = A[q_ >> 2][q_ & 3]; \
        }                                                                            \
    }
#define PCSF_PUSH_CHERRY(D)                                                          \
    {                                                                                \
        const double2_t* pa = operand_row_ptr<PAIRS>(leaf_tab, a, cod_a, g4);        \
        const double2_t* pb = operand_row_ptr<PAIRS>(leaf_tab, b, cod_b, g4);        \
        LeafRow ra, rb;                                                              \
        if (gap_before) {   /* SKIP: the recorded message of the all-gap subtree is parked instead of a product */ \
            gap_message(model * (p.n_edges + 1) + static_cast<int>(op_flags & 0x3fffu), stk[D]); \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) {                          \
                ra.v[i] = pa[i * ROW_PIECE_STRIDE];                                  \
                rb.v[i] = pb[i * ROW_PIECE_STRIDE];                                  \
            }                                                                        \
        } else {                                                                     \
        PCSF_PRODUCT_PROLOGUE()                                                      \
        product(buf, lane, cur, stk[D], [&](auto qc, auto jc) {                         \
            constexpr int q = decltype(qc)::value;                                   \
            constexpr int j = decltype(jc)::value;                                   \
            PCSF_STAGE_SLOT(q, j)                                                    \
            if constexpr (j == 2 && q < 8) ra.v[q] = pa[q * ROW_PIECE_STRIDE];       \
            if constexpr (j == 3 && q < 8) rb.v[q] = pb[q * ROW_PIECE_STRIDE];       \
        });                                                                          \
        PCSF_DUMP(stk[D])                                                            \
        }                                                                            \
        PCSF_STAMP(4)                                                                \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) {                              \
            cur[2 * i] = ra.v[i].x * rb.v[i].x;                                      \
            cur[2 * i + 1] = ra.v[i].y * rb.v[i].y;                                  \
        }                                                                            \
        PCSF_STAMP(3)                                                                \
    }
#define PCSF_POP(D)                                                                  \
    {                                                                                \
        if (gap_before) {                                                            \
            gap_message(model * (p.n_edges + 1) + static_cast<int>(op_flags & 0x3fffu), acc); \
        } else {                                                                     \
        PCSF_PRODUCT_PROLOGUE()                                                      \
        product(buf, lane, cur, acc, [&](auto qc, auto jc) {                         \
            constexpr int q = decltype(qc)::value;                                   \
            constexpr int j = decltype(jc)::value;                                   \
            PCSF_STAGE_SLOT(q, j)                                                    \
        });                                                                          \
        PCSF_DUMP(acc)                                                               \
        }                                                                            \
        PCSF_STAMP(2)                                                                \
        _Pragma("unroll") for (int q = 0; q < 16; ++q) cur[q] = stk[D][q >> 2][q & 3] * acc[q >> 2][q & 3]; \
        PCSF_STAMP(3)                                                                \
    }
            switch (code) {
                case F_CHERRY_DEV: {
                    const double2_t* pa = operand_row_ptr<PAIRS>(leaf_tab, a, cod_a, g4);
                    const double2_t* pb = operand_row_ptr<PAIRS>(leaf_tab, b, cod_b, g4);
                    LeafRow ra, rb;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        ra.v[i] = pa[i * ROW_PIECE_STRIDE];
                        rb.v[i] = pb[i * ROW_PIECE_STRIDE];
                    }
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        cur[2 * i] = ra.v[i].x * rb.v[i].x;
                        cur[2 * i + 1] = ra.v[i].y * rb.v[i].y;
                    }
                    break;
                }
                case F_MM_LEAF_DEV: {
                    const double2_t* pa = operand_row_ptr<PAIRS>(leaf_tab, a, cod_a, g4);
                    LeafRow ra;
                    if (gap_before) {   // SKIP: the subtree below is all gaps: its recorded message, then the leaf as usual
                        gap_message(model * (p.n_edges + 1) + static_cast<int>(op_flags & 0x3fffu), acc);
#pragma unroll
                        for (int i = 0; i < 8; ++i) ra.v[i] = pa[i * ROW_PIECE_STRIDE];
                    } else {
                    PCSF_PRODUCT_PROLOGUE()
                    product(buf, lane, cur, acc, [&](auto qc, auto jc) {
                        constexpr int q = decltype(qc)::value;
                        constexpr int j = decltype(jc)::value;
                        PCSF_STAGE_SLOT(q, j)
                        if constexpr (j == 2 && q >= PCSF_LEAF_ROW_Q0 && q < PCSF_LEAF_ROW_Q0 + 8) ra.v[q - PCSF_LEAF_ROW_Q0] = pa[(q - PCSF_LEAF_ROW_Q0) * ROW_PIECE_STRIDE];  // one 16-byte piece per k-step
                    });
                    PCSF_DUMP(acc)
                    }
                    PCSF_STAMP(2)
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        cur[2 * i] = acc[(2 * i) >> 2][(2 * i) & 3] * ra.v[i].x;
                        cur[2 * i + 1] = acc[(2 * i + 1) >> 2][(2 * i + 1) & 3] * ra.v[i].y;
                    }
                    PCSF_STAMP(3)
                    break;
                }
                case F_MM_PUSH_CHERRY0_DEV + 0:
                    PCSF_PUSH_CHERRY(0)
                    if constexpr (DEEP) {
                        _Pragma("unroll") for (int t = 0; t < 4; ++t) deep_stk[sp][t] = stk[0][t];
                        ++sp;
                    }
                    break;
                case F_MM_PUSH_CHERRY0_DEV + 1: PCSF_PUSH_CHERRY(1) break;
                case F_MM_PUSH_CHERRY0_DEV + 2: PCSF_PUSH_CHERRY(2) break;
                case F_MM_POP0_DEV + 0:
                    if constexpr (DEEP) {
                        --sp;
                        _Pragma("unroll") for (int t = 0; t < 4; ++t) stk[0][t] = deep_stk[sp][t];
                    }
                    PCSF_POP(0)
                    break;
                case F_MM_POP0_DEV + 1: PCSF_POP(1) break;
                case F_MM_POP0_DEV + 2: PCSF_POP(2) break;
                default: {  // F_ROOT
                    if (gap_before) {   // SKIP: every species a gap in all of the work-group's columns: the recorded root vector
                        const LeafRow rr = load_gap_row(model * (p.n_edges + 1) + p.n_edges);
#pragma unroll
                        for (int i = 0; i < 8; ++i) {
                            cur[2 * i] = rr.v[i].x;
                            cur[2 * i + 1] = rr.v[i].y;
                        }
                    }
                    if constexpr (STAMP) {
                        if (p.gap_dump && blockIdx.x == 0 && wave == 0 && sl == 0) {
                            double* d_ = p.gap_dump + (static_cast<size_t>(model) * (p.n_edges + 1) + p.n_edges) * 64 + g4 * 16;
#pragma unroll
                            for (int q_ = 0; q_ < 16; ++q_) d_[q_] = cur[q_];
                        }
                    }
                    double z = 0.0;
#pragma unroll
                    for (int q = 0; q < 16; ++q) z += prior[q] * cur[q];
                    z += __shfl_xor(z, 16);
                    z += __shfl_xor(z, 32);
                    const double lz = log(z);
                    if (model == 0) lnz0 = lz;
                    else lnz1 = lz;
                    if (p.out_anc) {  // sum_x ln(prior_x) * prior_x * alpha_root[x] / z  (root posterior term)
                        const double* __restrict__ w = p.anc_weight + static_cast<size_t>(prior0 + model) * 64 + g4 * 16;
                        double t = 0.0;
#pragma unroll
                        for (int q = 0; q < 16; ++q) t += w[q] * cur[q];
                        t += __shfl_xor(t, 16);
                        t += __shfl_xor(t, 32);
                        const double ratio = t / ((z == 0.0) ? 1.0 : z);  // select, not a branch: keeps the op loop uniform
                        t = (z == 0.0) ? 0.0 : ratio;
                        if (model == 0) anc0 = t;
                        else anc1 = t;
                    }
                    break;
                }
            }
#undef PCSF_PRODUCT_PROLOGUE
#undef PCSF_STAMP
#undef PCSF_STAMP_RAW
#undef PCSF_STAGE_SLOT
#undef PCSF_FETCH_NEXT_OP
#undef PCSF_PUSH_CHERRY
#undef PCSF_DUMP
#undef PCSF_POP
        }
    }

    // The last product of a work-group still stages a slot (into the idle buffer, branch-free): that LDS-DMA must have
    // landed before the wave ends -- the work-group's LDS is handed to the next work-group on this CU the moment its waves
    // are gone, and a piece arriving later would overwrite that group's first branch matrix (seen as one work-group in
    // ~30,000 with 64 slightly wrong scores).  The pieces were requested a whole product ago: the wait is normally free.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if constexpr (STAMP) {
        if (stamping && lane == 0)
            for (int k = 0; k < 8; ++k) {
                p.stamps[k] = seg_sum[k];
                p.stamps[8 + k] = seg_cnt[k];
            }
    }
    if (g4 == 0 && valid) {
        const size_t out_at = (BLOCKS && p.wg_out) ? p.wg_out[blockIdx.x] + (my_site - block_site0) : my_site;
        if (p.out_score) p.out_score[out_at] = 10.0 * (lnz0 - lnz1) / log(10.0);
        if (p.out_lnz) {
            p.out_lnz[out_at] = lnz0;
            if (p.n_passes > 1) p.out_lnz[p.n_sites + out_at] = lnz1;
        }
        if (p.out_anc) {
            p.out_anc[out_at] = anc0;
            if (p.n_passes > 1) p.out_anc[p.n_sites + out_at] = anc1;
        }
    }
}

template <int NW, bool STAMP, bool BLOCKS, bool FUSED = false, bool DEEP = false, bool SKIP = false, bool PAIRS = false>
static hipError_t launch_impl(const PruneParams& p_in, hipStream_t stream, size_t extra_lds)
{
    PruneParams p = p_in;
    if (SKIP && (!p.op_leaves || !p.op_meta || p.n_ops > 128 || p.n_leaves > 128)) return hipErrorInvalidValue;
    if (PAIRS && (!p.pair_leaves || p.n_pairs <= 0 || p.n_triples < 0 || (p.n_free_triples > 0 && !p.trip_info) || p.n_free_pairs < 0 || p.n_free_pairs > p.n_pairs ||
                  p.n_free_triples < 0 || p.n_free_triples > p.n_triples || p.n_quads < 0 || (p.n_quads > 0 && !p.quad_info) ||
                  p.leaf_rows != table_set_rows(p.n_leaves, p.n_pairs, p.n_triples, p.n_quads)))
        return hipErrorInvalidValue;
    const size_t sites_per_block = static_cast<size_t>(NW) * 16;
    const size_t blocks = BLOCKS ? p.n_wgs : (p.n_sites + sites_per_block - 1) / sites_per_block;
    if (blocks == 0) return hipSuccess;
    if (blocks > 0x7fffffffULL) return hipErrorInvalidValue;
    size_t lds = 2 * FRAG_DOUBLES_DEV * sizeof(double) + ((static_cast<size_t>(p.n_ops) + 3) & ~size_t(3)) * 4 +
                 static_cast<size_t>(p.n_leaves) * sites_per_block;
    if (FUSED) {
        if (!p.nucl || !p.segments || !p.wg_first_seg) return hipErrorInvalidValue;
        lds = ((lds + 15) & ~size_t(15)) + 256 + sites_per_block * 9 + 16;  // character classes, codon starts, strands
    }
    p.pair_lds_offset = 0;
    if (PAIRS) {   // the work-group's pair codes: two bytes per cherry and site
        lds = (lds + 15) & ~size_t(15);
        p.pair_lds_offset = static_cast<uint32_t>(lds);
        lds += ((static_cast<size_t>(p.n_free_pairs) * sites_per_block * 2 + 15) & ~size_t(15)) + static_cast<size_t>(p.n_free_triples + p.n_quads) * sites_per_block * 4;
    }
    lds += extra_lds;
    if (lds > 160 * 1024 - (SKIP ? 2048 : 0)) return hipErrorInvalidValue;   // SKIP: 1 KiB of static LDS beside the dynamic array
    // the attribute belongs to the (function, device) pair: one flag per template instantiation AND device.  Setting it
    // twice is harmless, so concurrent first launches from two host threads need no lock, only an atomic bit mask.
    static std::atomic<uint64_t> attr_set{0};
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev); e != hipSuccess) return e;
    if (!(attr_set.load(std::memory_order_acquire) >> (dev & 63) & 1)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&prune_fixed<NW, STAMP, BLOCKS, FUSED, DEEP, SKIP, PAIRS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - (SKIP ? 2048 : 0));
        if (e != hipSuccess) return e;
        attr_set.fetch_or(uint64_t(1) << (dev & 63), std::memory_order_release);
    }
    hipLaunchKernelGGL((prune_fixed<NW, STAMP, BLOCKS, FUSED, DEEP, SKIP, PAIRS>), dim3(static_cast<unsigned>(blocks)), dim3(NW * 64), lds, stream, p);
    return hipGetLastError();
}

hipError_t launch_prune_fixed(const PruneParams& p, int variant, hipStream_t stream, size_t extra_lds)
{
    switch (variant) {
        case 0: return launch_impl<4, false, false>(p, stream, extra_lds);
        case 1: return launch_impl<8, false, false>(p, stream, extra_lds);
        case 2: return launch_impl<4, true, false>(p, stream, extra_lds);  // diagnostic: in-kernel stamps
        case 16: return launch_impl<4, false, true>(p, stream, extra_lds); // per-work-group windows/tables
        case 32: return launch_impl<4, false, false, true>(p, stream, extra_lds);  // tracks mode, codons formed from the nucleotide rows
        case 64: return launch_impl<4, false, false, false, true>(p, stream, extra_lds);       // + 64: parked tiles in scratch (deep trees)
        case 64 + 2: return launch_impl<4, true, false, false, true>(p, stream, extra_lds);
        case 64 + 16: return launch_impl<4, false, true, false, true>(p, stream, extra_lds);
        case 64 + 32: return launch_impl<4, false, false, true, true>(p, stream, extra_lds);
        // + 128 (tracks mode, trees of at most 128 leaves): all-gap subtrees of a work-group's columns are skipped
        case 128: return launch_impl<4, false, false, false, false, true>(p, stream, extra_lds);
        case 128 + 32: return launch_impl<4, false, false, true, false, true>(p, stream, extra_lds);
        case 128 + 64: return launch_impl<4, false, false, false, true, true>(p, stream, extra_lds);
        case 128 + 64 + 32: return launch_impl<4, false, false, true, true, true>(p, stream, extra_lds);
        // + 256: cherry tables (the reduced tree's program; the context's own branch lengths)
        case 256: return launch_impl<4, false, false, false, false, false, true>(p, stream, extra_lds);
        case 256 + 2: return launch_impl<4, true, false, false, false, false, true>(p, stream, extra_lds);
        case 256 + 32: return launch_impl<4, false, false, true, false, false, true>(p, stream, extra_lds);
        case 256 + 64: return launch_impl<4, false, false, false, true, false, true>(p, stream, extra_lds);
        case 256 + 64 + 2: return launch_impl<4, true, false, false, true, false, true>(p, stream, extra_lds);
        case 256 + 64 + 32: return launch_impl<4, false, false, true, true, false, true>(p, stream, extra_lds);
        case 256 + 128: return launch_impl<4, false, false, false, false, true, true>(p, stream, extra_lds);
        case 256 + 128 + 32: return launch_impl<4, false, false, true, false, true, true>(p, stream, extra_lds);
        case 256 + 128 + 64: return launch_impl<4, false, false, false, true, true, true>(p, stream, extra_lds);
        case 256 + 128 + 64 + 32: return launch_impl<4, false, false, true, true, true, true>(p, stream, extra_lds);
        default: return hipErrorInvalidValue;
    }
}

// The rows of a subtree table (PAIRS): one work-group = 64 codes of one table and model; a lane multiplies the two operand rows of its
// code like the CHERRY arm above does and runs product() -- the instruction sequence the full program runs on that edge, so the row
// holds the bits the full program would compute -- and stores the result as a row of the table set.
__global__ __launch_bounds__(256) void combine_table_kernel(const CombineTable* __restrict__ tables, const double* __restrict__ frag, int n_frag, double* __restrict__ leaf_tab_all,
                                                            uint32_t leaf_rows)
{
    __shared__ __attribute__((aligned(16))) double2_t lfrag[FRAG_DOUBLES_DEV / 2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, sl = lane & 15, g4 = lane >> 4;
    const int model = blockIdx.z;
    const CombineTable t = tables[blockIdx.y];
    const uint32_t n_codes = t.x_size * t.y_size;
    if (blockIdx.x * 64u >= n_codes) return;   // (the launch is sized for its largest table)
    const double2_t* src = reinterpret_cast<const double2_t*>(frag + (static_cast<size_t>(model) * n_frag + t.frag) * FRAG_DOUBLES_DEV);
    for (int i = tid; i < FRAG_DOUBLES_DEV / 2; i += 256) lfrag[i] = src[i];
    __syncthreads();
    const uint32_t code = blockIdx.x * 64u + wave * 16u + sl;
    const bool valid = code < n_codes;
    const uint32_t cc = valid ? code : 0u, cx = cc / t.y_size, cy = cc % t.y_size;
    double* leaf_tab = leaf_tab_all + static_cast<size_t>(model) * leaf_rows * 64;
    const double2_t* px = operand_row_ptr<true>(leaf_tab, 0, t.x_base + cx, g4);
    const double2_t* py = operand_row_ptr<true>(leaf_tab, 0, t.y_base + cy, g4);
    double cur[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const double2_t rx = px[i * ROW_PIECE_STRIDE], ry = py[i * ROW_PIECE_STRIDE];
        cur[2 * i] = rx.x * ry.x;
        cur[2 * i + 1] = rx.y * ry.y;
    }
    double4_t acc[4];
    product(lfrag, lane, cur, acc, [](auto, auto) {});
    if (valid) {
        double2_t* dst = reinterpret_cast<double2_t*>(leaf_tab + (static_cast<size_t>(t.out_base) + code) * 64) + g4;
#pragma unroll
        for (int i = 0; i < 8; ++i) dst[i * ROW_PIECE_STRIDE] = double2_t{acc[(2 * i) >> 2][(2 * i) & 3], acc[(2 * i + 1) >> 2][(2 * i + 1) & 3]};
    }
}

hipError_t launch_combine_tables(const CombineTable* d_tables, int n_tables, uint32_t max_rows, const double* frag, int n_frag, double* leaf_tab, uint32_t leaf_rows, hipStream_t stream)
{
    if (n_tables <= 0 || max_rows == 0) return hipSuccess;
    hipLaunchKernelGGL(combine_table_kernel, dim3((max_rows + 63) / 64, static_cast<unsigned>(n_tables), 2), dim3(256), 0, stream, d_tables, frag, n_frag, leaf_tab, leaf_rows);
    return hipGetLastError();
}

int prune_variant_count() { return 3; }   // variants selectable with PCSF_VARIANT (0, 1, 2); block / fused / deep-tree forms are chosen by the entry points

}  // namespace pcsf
