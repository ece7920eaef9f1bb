// wig.hip -- the text of build-tracks' seven wig files, formatted on the device (byte work; reference: the output loop of
// src/phylocsf++build_tracks.hpp:139-156 (power track) and :183-217 (six raw tracks), my_fprintf / print_trimmed src/common.hpp:48-68).
//
// Per batch the scores (d_out, fp64 decibans per codon in forward genome order) and the per-base branch-length scores (d_bls) are
// already resident; the host used to download both (16 + 8 bytes per codon-frame and column), narrow, filter, format and append --
// 0.5-0.6 s of sixteen host threads per 100 Mb, the next ceiling of an eight-GPU node after the parser.  Here:
//   wig_lengths_kernel  one thread per LINE CANDIDATE (a codon of a frame, or a codon of the power track): is it written
//                       (bls_codon_sum >= 3 x threshold, build_tracks.hpp:201-205), does a header line come before it (the codon
//                       before it was not written, or it is its alignment's first: :207-210), how long is its text
//   exclusive scan      (rocPRIM) of the lengths over all seven files at once: every line's byte offset; a file's text is the
//                       stretch between the offsets of its first candidates
//   wig_write_kernel    every thread formats its line again and stores it at its offset
// All three are enqueued behind the scoring kernels without the host in between: the text buffer is sized from an estimate (12 bytes per
// candidate: a line is ~8.4), lines that would not fit are left out, and the caller -- who reads the eight file boundaries and so the
// total afterwards -- grows the buffer and runs the write pass again if that ever happens.  (With the sizes fetched first, the write pass
// was launched into a device already full of the other context's pruning work-groups and the call waited ~2 ms for it.)
// The host copies the text down and write()s it.  The value text is format_trimmed's (host/tracks.hpp): the float times 10^D is
// exact in double, rounded to nearest-even like printf, trailing zeros removed but one decimal kept; NaN and infinities print as
// glibc prints them; a finite |value| >= 9e14 (never a score) raises a flag and the caller formats that batch on the host.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include <cstring>

#include <rocprim/device/device_scan.hpp>

#include "kernels.h"

namespace pcsf {

namespace {

// digits of v (< 10^19) into dst, most significant first; returns their number
__device__ __forceinline__ int put_uint(char* dst, unsigned long long v)
{
    char tmp[20];
    int n = 0;
    do {
        tmp[n++] = static_cast<char>('0' + v % 10ull);
        v /= 10ull;
    } while (v);
    for (int i = 0; i < n; ++i) dst[i] = tmp[n - 1 - i];
    return n;
}
__device__ __forceinline__ int count_digits(unsigned long long v)
{
    int n = 1;
    while (v >= 10ull) {
        v /= 10ull;
        ++n;
    }
    return n;
}

// format_trimmed<D> (host/tracks.hpp) without the terminator; dst == nullptr: the length only.  *big: the value needs the host path.
template <int D>
__device__ __forceinline__ int format_value(float v, char* dst, bool* big)
{
    const double x = static_cast<double>(v);
    const bool neg = __builtin_signbit(x);
    if (x != x) {   // glibc: "nan" / "-nan"
        if (dst) {
            int n = 0;
            if (neg) dst[n++] = '-';
            dst[n++] = 'n';
            dst[n++] = 'a';
            dst[n++] = 'n';
        }
        return neg ? 4 : 3;
    }
    if (__builtin_isinf(x)) {
        if (dst) {
            int n = 0;
            if (neg) dst[n++] = '-';
            dst[n++] = 'i';
            dst[n++] = 'n';
            dst[n++] = 'f';
        }
        return neg ? 4 : 3;
    }
    if (!(fabs(x) < 9e14)) {
        *big = true;
        return 0;
    }
    constexpr unsigned long long POW = D == 3 ? 1000ull : 10000ull;
    const unsigned long long scaled = static_cast<unsigned long long>(rint(fabs(x) * static_cast<double>(POW)));   // nearest, ties to even, on the exact product
    const unsigned long long ip = scaled / POW;
    unsigned frac = static_cast<unsigned>(scaled % POW);
    int last = D - 1;   // index of the last decimal kept: trailing zeros go, one decimal stays
    {
        unsigned f = frac;
        while (last > 0 && f % 10u == 0u) {
            f /= 10u;
            --last;
        }
    }
    const int len = (neg ? 1 : 0) + count_digits(ip) + 1 + (last + 1);
    if (dst) {
        char* p = dst;
        if (neg) *p++ = '-';
        p += put_uint(p, ip);
        *p++ = '.';
        char fr[D];
        for (int i = D - 1; i >= 0; --i) {
            fr[i] = static_cast<char>('0' + frac % 10u);
            frac /= 10u;
        }
        for (int i = 0; i <= last; ++i) *p++ = fr[i];
    }
    return len;
}

// "fixedStep chrom=<chrom> start=<pos> step=3 span=3\n"
__device__ __forceinline__ int header_text(const char* __restrict__ chrom, uint32_t chrom_len, long long pos, char* dst)
{
    const bool neg = pos < 0;
    const unsigned long long mag = neg ? static_cast<unsigned long long>(-pos) : static_cast<unsigned long long>(pos);
    const int len = 16 + static_cast<int>(chrom_len) + 7 + (neg ? 1 : 0) + count_digits(mag) + 15;
    if (dst) {
        const char a[] = "fixedStep chrom=", b[] = " start=", c[] = " step=3 span=3\n";
        char* p = dst;
        for (int i = 0; i < 16; ++i) *p++ = a[i];
        for (uint32_t i = 0; i < chrom_len; ++i) *p++ = chrom[i];
        for (int i = 0; i < 7; ++i) *p++ = b[i];
        if (neg) *p++ = '-';
        p += put_uint(p, mag);
        for (int i = 0; i < 15; ++i) *p++ = c[i];
    }
    return len;
}

// line candidate e of the call -> its file, its frame record and its index within the record
struct Candidate {
    int file;
    const WigFrame* fr;
    uint32_t xx;
};
__device__ __forceinline__ Candidate locate(const WigParams& p, uint64_t e)
{
    int file = 0;
    while (file < 6 && e >= p.file_elem0[file + 1]) ++file;
    // the file's records in alignment order: binary search for the last one whose first candidate is <= e
    const WigFrame* recs = p.frames + p.file_frame0[file];
    uint32_t lo = 0, hi = p.file_frame0[file + 1] - p.file_frame0[file];   // [lo, hi)
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (recs[mid].elem0 <= e) lo = mid;
        else hi = mid;
    }
    return Candidate{file, recs + lo, static_cast<uint32_t>(e - recs[lo].elem0)};
}

// WRITE = false: the candidate's text length into p.lens; true: its text at p.offsets
template <bool WRITE>
__global__ __launch_bounds__(256) void wig_kernel(const WigParams p)
{
    const uint64_t e = static_cast<uint64_t>(blockIdx.x) * 256 + threadIdx.x;
    if (e >= p.n_elems) return;
    const Candidate c = locate(p, e);
    const WigFrame& f = *c.fr;
    const double* __restrict__ b = p.bls + f.bls0 + 3ull * c.xx;
    uint32_t len = 0;
    char* dst = nullptr;
    if (WRITE) {
        if (e < 8) p.bounds[e] = p.offsets[p.file_elem0[e]];   // the eight file boundaries (n_elems >= 8 or not: see launch_wig_write)
        // a line beyond the buffer (the caller sized it from an estimate) is left out; the total tells the caller, who writes again
        const uint32_t at = p.offsets[e];
        if (static_cast<uint64_t>(at) + p.lens[e] > p.text_capacity || p.lens[e] == 0) return;
        dst = p.text + at;
    }
    bool big = false;
    if (c.file == 0) {
        // power track (build_tracks.hpp:139-156): one header per alignment, every codon of frame 3 of the + strand
        if (c.xx == 0) {
            const int h = header_text(p.chroms + f.chrom0, f.chrom_len, f.start_pos, dst);
            len += h;
            if (WRITE) dst += h;
        }
        const float avg = static_cast<float>((b[0] + b[1] + b[2]) / 3.0);
        const int n = format_value<4>(avg, dst, &big);
        if (WRITE) dst[n] = '\n';
        len += n + 1;
    } else {
        const float sum = static_cast<float>(b[0] + b[1] + b[2]);
        if (!(sum < p.threshold3)) {
            bool header = c.xx == 0;
            if (!header) {
                const double* __restrict__ bp = b - 3;
                header = static_cast<float>(bp[0] + bp[1] + bp[2]) < p.threshold3;   // the codon before was not written: a new run
            }
            if (header) {
                const int h = header_text(p.chroms + f.chrom0, f.chrom_len, f.start_pos + 3ll * c.xx, dst);
                len += h;
                if (WRITE) dst += h;
            }
            const int n = format_value<3>(static_cast<float>(p.scores[f.score0 + c.xx]), dst, &big);
            if (WRITE) dst[n] = '\n';
            len += n + 1;
        }
    }
    if (!WRITE) {
        p.lens[e] = len;
        if (big) *p.needs_host = 1u;
    }
}

__global__ void wig_bounds_kernel(const WigParams p)
{
    p.bounds[threadIdx.x] = p.offsets[p.file_elem0[threadIdx.x]];
}

// tests: one value per thread into a 32-byte slot, its length beside it (0xffffffff: the host formats this one)
__global__ __launch_bounds__(256) void wig_format_debug_kernel(const float* __restrict__ v, uint32_t n, int decimals, char* __restrict__ text, uint32_t* __restrict__ lens)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    bool big = false;
    const int len = decimals == 4 ? format_value<4>(v[i], text + 32ull * i, &big) : format_value<3>(v[i], text + 32ull * i, &big);
    lens[i] = big ? 0xffffffffu : static_cast<uint32_t>(len);
}

}  // namespace

hipError_t launch_wig_format_debug(const float* v, uint32_t n, int decimals, char* text, uint32_t* lens, hipStream_t stream)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(wig_format_debug_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, v, n, decimals, text, lens);
    return hipGetLastError();
}

size_t wig_scan_temp_bytes(uint64_t n_elems)
{
    size_t bytes = 0;
    (void)rocprim::exclusive_scan(nullptr, bytes, static_cast<const uint32_t*>(nullptr), static_cast<uint32_t*>(nullptr), 0u, n_elems + 1, rocprim::plus<uint32_t>(), nullptr);
    return bytes;
}

// lens[n_elems] is followed by one zero: offsets[n_elems] is then the total number of bytes
hipError_t launch_wig_lengths(const WigParams& p, void* d_temp, size_t temp_bytes, hipStream_t stream)
{
    if (p.n_elems == 0) return hipSuccess;
    hipLaunchKernelGGL(wig_kernel<false>, dim3(static_cast<unsigned>((p.n_elems + 255) / 256)), dim3(256), 0, stream, p);
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    if (hipError_t e = hipMemsetAsync(p.lens + p.n_elems, 0, sizeof(uint32_t), stream); e != hipSuccess) return e;
    return rocprim::exclusive_scan(d_temp, temp_bytes, p.lens, p.offsets, 0u, p.n_elems + 1, rocprim::plus<uint32_t>(), stream);
}

// also fills p.bounds[0..7]; a call with fewer than eight candidates gets them from a second, eight-thread kernel
hipError_t launch_wig_write(const WigParams& p, hipStream_t stream)
{
    if (p.n_elems == 0) return hipSuccess;
    hipLaunchKernelGGL(wig_kernel<true>, dim3(static_cast<unsigned>((p.n_elems + 255) / 256)), dim3(256), 0, stream, p);
    if (hipError_t e = hipGetLastError(); e != hipSuccess) return e;
    if (p.n_elems < 8) {
        hipLaunchKernelGGL(wig_bounds_kernel, dim3(1), dim3(8), 0, stream, p);
        return hipGetLastError();
    }
    return hipSuccess;
}

}  // namespace pcsf
