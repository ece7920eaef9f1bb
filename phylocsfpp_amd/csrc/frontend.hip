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
// that order; evaluating the nodes bottom-up (children have smaller indices than their parent) performs the same
// additions on the same operands, so the double results, and the float threshold decisions made from them, are
// identical.  One thread per column: S is a 128-bit mask built from the column's characters (coalesced across the
// work-group: consecutive threads read consecutive columns of a species row), overlaps are popcounts against the
// per-node leaf masks held in LDS, and the per-node partial sums live in the thread's scratch.
__global__ __launch_bounds__(256) void bls_kernel(BlsTree t, const BlsSpan* __restrict__ spans, const unsigned char* __restrict__ nucl,
                                                  size_t row_stride, double* __restrict__ out, unsigned long long* __restrict__ first_bad)
{
    __shared__ uint64_t s_range[BLS_MAX_NODES][2];
    __shared__ double s_len[BLS_MAX_NODES];
    __shared__ int16_t s_c1[BLS_MAX_NODES], s_c2[BLS_MAX_NODES];
    for (int i = threadIdx.x; i < t.n_nodes; i += 256) {
        s_range[i][0] = t.range[2 * i];
        s_range[i][1] = t.range[2 * i + 1];
        s_len[i] = t.len[i];
        s_c1[i] = t.child1[i];
        s_c2[i] = t.child2[i];
    }
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
    double val[BLS_MAX_NODES];
    for (int i = 0; i < t.n_nodes; ++i) {
        const int ov = __popcll(lo & s_range[i][0]) + __popcll(hi & s_range[i][1]);
        if (ov == 0) continue;  // never visited
        if (i < t.n_leaves) {
            val[i] = s_len[i];
            continue;
        }
        const int l = s_c1[i], r = s_c2[i];
        const int ol = __popcll(lo & s_range[l][0]) + __popcll(hi & s_range[l][1]);
        double bl = ov < in_set ? s_len[i] : 0.0;  // strictly below the last common ancestor of S
        if (ol > 0) bl += val[l];
        if (ov - ol > 0) bl += val[r];
        val[i] = bl;
    }
    out[col] = val[t.n_nodes - 1] / t.total;
}

hipError_t launch_bls(const BlsTree& tree, const BlsSpan* d_spans, uint32_t n_spans, uint64_t max_len, const unsigned char* d_nucl,
                      size_t row_stride, double* d_out, unsigned long long* d_first_bad, hipStream_t stream)
{
    if (n_spans == 0 || max_len == 0) return hipSuccess;
    if (n_spans > 65535u || tree.n_nodes > BLS_MAX_NODES) return hipErrorInvalidValue;
    hipLaunchKernelGGL(bls_kernel, dim3(static_cast<unsigned>((max_len + 255u) / 256u), n_spans), dim3(256), 0, stream, tree, d_spans, d_nucl,
                       row_stride, d_out, d_first_bad);
    return hipGetLastError();
}

}  // namespace pcsf
