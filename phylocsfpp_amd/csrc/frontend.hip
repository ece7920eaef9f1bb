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

__device__ __forceinline__ unsigned nucleotide_id(unsigned char c)
{
    switch (c) {
        case 'A': case 'a': return 0u;
        case 'C': case 'c': return 1u;
        case 'G': case 'g': return 2u;
        case 'T': case 't': return 3u;
        default: return 4u;  // N n - . (the host rejects every other character before the upload)
    }
}

__global__ __launch_bounds__(256) void extract_frames_kernel(const FrameSegment* __restrict__ seg, const unsigned char* __restrict__ nucl,
                                                             size_t row_stride, int n_leaves, unsigned char* __restrict__ codons,
                                                             size_t leaf_stride)
{
    const FrameSegment s = seg[blockIdx.y];
    const uint32_t c = blockIdx.x * 256u + threadIdx.x;
    if (c >= s.n_codons) return;
    const bool minus = s.minus != 0;
    // '+': columns col0+skip+3c .. +2 ; '-': the reverse-complemented row starts at the last column
    const size_t first = minus ? s.col0 + s.len - 1 - s.skip - 3 * static_cast<size_t>(c) : s.col0 + s.skip + 3 * static_cast<size_t>(c);
    const size_t dst = s.dst0 + (minus ? (s.n_codons - 1 - c) : c);
    for (int l = 0; l < n_leaves; ++l) {
        const unsigned char* row = nucl + static_cast<size_t>(l) * row_stride;
        unsigned a, b, d;
        if (minus) {
            a = nucleotide_id(row[first]);
            b = nucleotide_id(row[first - 1]);
            d = nucleotide_id(row[first - 2]);
            a = a < 4u ? 3u - a : a;  // complement: A<->T, C<->G
            b = b < 4u ? 3u - b : b;
            d = d < 4u ? 3u - d : d;
        } else {
            a = nucleotide_id(row[first]);
            b = nucleotide_id(row[first + 1]);
            d = nucleotide_id(row[first + 2]);
        }
        const unsigned cod = (a == 4u || b == 4u || d == 4u) ? 64u : 16u * a + 4u * b + d;
        codons[static_cast<size_t>(l) * leaf_stride + dst] = static_cast<unsigned char>(cod);
    }
}

hipError_t launch_extract_frames(const FrameSegment* d_seg, uint32_t n_segments, uint32_t max_codons, const unsigned char* d_nucl,
                                 size_t row_stride, int n_leaves, unsigned char* d_codons, size_t leaf_stride, hipStream_t stream)
{
    if (n_segments == 0 || max_codons == 0) return hipSuccess;
    if (n_segments > 65535u) return hipErrorInvalidValue;
    hipLaunchKernelGGL(extract_frames_kernel, dim3((max_codons + 255u) / 256u, n_segments), dim3(256), 0, stream, d_seg, d_nucl, row_stride,
                       n_leaves, d_codons, leaf_stride);
    return hipGetLastError();
}

}  // namespace pcsf
