// frames.hpp -- six-frame arithmetic shared by the host driver and the device front end.
// Restates alignment_t::update_seqs (reference src/parallel_file_reader.hpp:61-113) and the score
// placement of the build-tracks loop (src/phylocsf++build_tracks.hpp:162-227).
#pragma once

#include <cstddef>
#include <cstdint>

namespace pcsf {

struct FrameView {
    unsigned skip = 0;       // leading columns (of the strand's own row orientation) before the first codon
    size_t n_codons = 0;
    uint64_t start_pos = 0;  // genome position (1-based) of the first emitted score
    size_t bls_offset = 0;   // index into the forward per-base BLS vector of the first score
};

// orig_start: 1-based start of the alignment; len: columns after reference-gap removal
inline FrameView frame_view(uint64_t orig_start, uint64_t chrom_len, size_t len, char strand, unsigned frame)
{
    FrameView v;
    int64_t skip;
    if (strand == '+') skip = (static_cast<int64_t>(frame) - static_cast<int64_t>(orig_start)) % 3;
    else skip = (static_cast<int64_t>(frame) - static_cast<int64_t>(chrom_len - (orig_start + len) + 2)) % 3;
    if (skip < 0) skip += 3;
    if (static_cast<uint64_t>(skip) > len) skip = static_cast<int64_t>(len);
    v.skip = static_cast<unsigned>(skip);
    v.n_codons = (len - v.skip) / 3;
    if (strand == '+') {
        v.start_pos = orig_start + v.skip;
        v.bls_offset = v.skip;
    } else {
        // codons are read right-to-left; the 0-2 left-over bases sit at the left end of the forward strand
        v.start_pos = orig_start + (len - v.skip) % 3;
        v.bls_offset = (len - v.skip) % 3;
    }
    return v;
}

}  // namespace pcsf
