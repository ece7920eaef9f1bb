// frontend.hip -- six-frame codon extraction on the device.
// Restates what the reference does on the host for every alignment and frame: alignment_t::update_seqs /
// get_amino_acid_id (src/parallel_file_reader.hpp:61-113, src/translation.hpp:29-88) and the reverse
// complement between the strands (src/phylocsf++build_tracks.hpp:219-226).  The nucleotide rows are uploaded
// once; each (alignment, frame) segment is translated into the leaf-major codon matrix the pruning kernel
// reads.  '-' strand codons are written mirrored so that the scores come out in forward genome order (the
// reference reverses the score vector afterwards, src/phylocsf++build_tracks.hpp:175-178).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

namespace pcsf {

// 0..3 = ACGT (either case), 4 = N n - ., 5 = a character the reference exits on (src/translation.hpp:29-88)
__device__ __forceinline__ unsigned nucleotide_id(unsigned char c)
{
    switch (c) {
        case 'A': case 'a': return 0u;
        case 'C': case 'c': return 1u;
        case 'G': case 'g': return 2u;
        case 'T': case 't': return 3u;
        case 'N': case 'n': case '-': case '.': return 4u;
        default: return 5u;
    }
}

__global__ __launch_bounds__(256) void extract_frames_kernel(const FrameSegment* __restrict__ seg, const unsigned char* __restrict__ nucl,
                                                             size_t row_stride, int n_leaves, unsigned char* __restrict__ codons,
                                                             size_t leaf_stride, unsigned long long* __restrict__ first_bad)
{
    // 64 codons x 4 groups of species per work-group: a wave writes 64 consecutive codons of one species row, and the
    // four waves walk different species (shorter dependent chains than one thread per codon over all species)
    const FrameSegment s = seg[blockIdx.y];
    const uint32_t c = blockIdx.x * 64u + (threadIdx.x & 63u);
    if (c >= s.n_codons) return;
    const bool minus = s.minus != 0;
    // '+': columns col0+skip+3c .. +2 ; '-': the reverse-complemented row starts at the last column
    const size_t first = minus ? s.col0 + s.len - 1 - s.skip - 3 * static_cast<size_t>(c) : s.col0 + s.skip + 3 * static_cast<size_t>(c);
    const size_t dst = s.dst0 + (minus ? (s.n_codons - 1 - c) : c);
#pragma unroll 4
    for (int l = static_cast<int>(threadIdx.x >> 6); l < n_leaves; l += 4) {
        const unsigned char* row = nucl + static_cast<size_t>(l) * row_stride;
        unsigned a, b, d;
        const size_t second = minus ? first - 1 : first + 1, third = minus ? first - 2 : first + 2;
        a = nucleotide_id(row[first]);
        b = nucleotide_id(row[second]);
        d = nucleotide_id(row[third]);
        if ((a == 5u || b == 5u || d == 5u) && first_bad) {  // a character outside ACGTNacgtn-.: the call fails with its position
            const size_t where = a == 5u ? first : (b == 5u ? second : third);
            atomicMin(first_bad, static_cast<unsigned long long>(static_cast<size_t>(l) * row_stride + where));
        }
        if (minus) {  // complement: A<->T, C<->G
            a = a < 4u ? 3u - a : a;
            b = b < 4u ? 3u - b : b;
            d = d < 4u ? 3u - d : d;
        }
        const unsigned cod = (a >= 4u || b >= 4u || d >= 4u) ? 64u : 16u * a + 4u * b + d;
        codons[static_cast<size_t>(l) * leaf_stride + dst] = static_cast<unsigned char>(cod);
    }
}

hipError_t launch_extract_frames(const FrameSegment* d_seg, uint32_t n_segments, uint32_t max_codons, const unsigned char* d_nucl,
                                 size_t row_stride, int n_leaves, unsigned char* d_codons, size_t leaf_stride, unsigned long long* d_first_bad,
                                 hipStream_t stream)
{
    if (n_segments == 0 || max_codons == 0) return hipSuccess;
    if (n_segments > 65535u) return hipErrorInvalidValue;
    hipLaunchKernelGGL(extract_frames_kernel, dim3((max_codons + 63u) / 64u, n_segments), dim3(256), 0, stream, d_seg, d_nucl, row_stride,
                       n_leaves, d_codons, leaf_stride, d_first_bad);
    return hipGetLastError();
}

// ---- branch-length score ------------------------------------------------------------------------------------
// compute_bls_score / newick_sum_branch_lengths (src/additional_scores.hpp:5-84) per alignment column: S = the
// species with one of ACGTacgt in the column; fewer than two -> 0; otherwise the branch lengths of the subtree
// spanned by S -- every node strictly below S's last common ancestor whose leaves meet S -- over the same sum
// for the full species set.  The reference recursion adds, per node, (own branch, left subtree, right subtree) in
// that order; any evaluation order that has both children's values ready performs the same additions on the same
// operands, so the double results, and the float threshold decisions made from them, are identical.
// One thread per column: S is a 128-bit mask built from the column's characters (coalesced across the work-group:
// consecutive threads read consecutive columns of a species row); overlaps are popcounts against the per-node leaf
// masks held in LDS.  The internal nodes run as a stack program (api.cpp `bls_program`: the deeper subtree first, a leaf
// child's value is its branch length and never touches the stack), the waiting values in a column of LDS per thread
// (conflict-free: thread t uses bank pair t) -- the first version kept one value per NODE in a per-thread array,
// i.e. 2 KB of scratch memory per thread and a dependent scratch round trip per node: 0.12 TB/s on a byte kernel.
__global__ __launch_bounds__(256) void bls_kernel(BlsTree t, const BlsSpan* __restrict__ spans, const unsigned char* __restrict__ nucl,
                                                  size_t row_stride, double* __restrict__ out, unsigned long long* __restrict__ first_bad)
{
    __shared__ uint64_t s_range[BLS_MAX_NODES][2];
    __shared__ double s_len[BLS_MAX_NODES];
    __shared__ uint32_t s_prog[128];
    __shared__ double s_stack[BLS_STACK_DEPTH][256];
    for (int i = threadIdx.x; i < t.n_nodes; i += 256) {
        s_range[i][0] = t.range[2 * i];
        s_range[i][1] = t.range[2 * i + 1];
        s_len[i] = t.len[i];
    }
    for (int i = threadIdx.x; i < t.n_prog; i += 256) s_prog[i] = t.prog[i];
    __syncthreads();
    const BlsSpan span = spans[blockIdx.y];
    const uint64_t x = static_cast<uint64_t>(blockIdx.x) * 256u + threadIdx.x;
    if (x >= span.len) return;
    const size_t col = span.col0 + x;
    uint64_t lo = 0, hi = 0;
    for (int l = 0; l < t.n_leaves; ++l) {
        const unsigned char ch = nucl[static_cast<size_t>(l) * row_stride + col];
        const unsigned char up = ch & 0xDFu;  // letters to upper case
        const bool acgt = up == 'A' || up == 'C' || up == 'G' || up == 'T';
        if (acgt) {
            if (l < 64) lo |= uint64_t(1) << l;
            else hi |= uint64_t(1) << (l - 64);
        } else if (!(up == 'N' || ch == '-' || ch == '.')) {
            atomicMin(first_bad, static_cast<unsigned long long>(static_cast<size_t>(l) * row_stride + col));
        }
    }
    const int in_set = __popcll(lo) + __popcll(hi);
    if (in_set < 2) {
        out[col] = 0.0;
        return;
    }
    double* stk = &s_stack[0][threadIdx.x];   // + 256 per level
    int sp = 0;
    double bl = 0.0;   // a tree of one leaf has no internal node (and never gets here: in_set < 2)
    for (int k = 0; k < t.n_prog; ++k) {
        const uint32_t w = s_prog[k];
        const int i = w & 0xffu, l = (w >> 8) & 0xffu, r = (w >> 16) & 0xffu;
        const uint32_t flags = w >> 24;
        const int ol = __popcll(lo & s_range[l][0]) + __popcll(hi & s_range[l][1]);
        const int orr = __popcll(lo & s_range[r][0]) + __popcll(hi & s_range[r][1]);
        double vl = s_len[l], vr = s_len[r];
        if (!(flags & (BLS_LEFT_LEAF | BLS_RIGHT_LEAF))) {   // two values waiting (wave-uniform branches: the program is the same for every column)
            const double top = stk[(sp - 1) * 256], below = stk[(sp - 2) * 256];
            sp -= 2;
            vl = (flags & BLS_RIGHT_ON_TOP) ? below : top;
            vr = (flags & BLS_RIGHT_ON_TOP) ? top : below;
        } else if (!(flags & BLS_LEFT_LEAF)) {
            vl = stk[--sp * 256];
        } else if (!(flags & BLS_RIGHT_LEAF)) {
            vr = stk[--sp * 256];
        }
        bl = ol + orr < in_set ? s_len[i] : 0.0;  // strictly below the last common ancestor of S (a node outside S's subtree gets a value nobody adds)
        if (ol > 0) bl += vl;
        if (orr > 0) bl += vr;
        stk[sp++ * 256] = bl;
    }
    out[col] = bl / t.total;
}

hipError_t launch_bls(const BlsTree& tree, const BlsSpan* d_spans, uint32_t n_spans, uint64_t max_len, const unsigned char* d_nucl,
                      size_t row_stride, double* d_out, unsigned long long* d_first_bad, hipStream_t stream)
{
    if (n_spans == 0 || max_len == 0) return hipSuccess;
    if (n_spans > 65535u || tree.n_nodes > BLS_MAX_NODES || tree.n_prog > 128 || !tree.prog) return hipErrorInvalidValue;
    hipLaunchKernelGGL(bls_kernel, dim3(static_cast<unsigned>((max_len + 255u) / 256u), n_spans), dim3(256), 0, stream, tree, d_spans, d_nucl,
                       row_stride, d_out, d_first_bad);
    return hipGetLastError();
}

}  // namespace pcsf
