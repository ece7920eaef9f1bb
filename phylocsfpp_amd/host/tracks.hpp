// tracks.hpp -- everything between an Alignment and the text the reference writes:
// branch-length score, six-frame codon extraction, wig / .scores formatting.
// Restates (from scratch) the behaviour of the reference's
//   compute_bls_score / newick_sum_branch_lengths   src/additional_scores.hpp:5-84
//   alignment_t::update_seqs / translate            src/parallel_file_reader.hpp:46-113
//   get_amino_acid_id / complement                  src/translation.hpp:29-88
//   the per-frame output loop                       src/phylocsf++build_tracks.hpp:139-227
//   my_fprintf                                      src/common.hpp:48-68
#pragma once

#include <algorithm>
#include <cinttypes>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../csrc/frames.hpp"
#include "maf_reader.hpp"
#include "model_io.hpp"

namespace pcsf_host {

// ---- nucleotides and codons ------------------------------------------------------------------
// 0..3 = A C G T (either case), 4 = N n - . ; anything else is an input error (the reference exits)
inline int nucleotide_id(char c)
{
    switch (c) {
        case 'A': case 'a': return 0;
        case 'C': case 'c': return 1;
        case 'G': case 'g': return 2;
        case 'T': case 't': return 3;
        case 'N': case 'n': case '-': case '.': return 4;
        default: throw std::runtime_error(std::string("unexpected character '") + c + "' in alignment");
    }
}

inline uint8_t codon_id(char a, char b, char c)
{
    const int x = nucleotide_id(a), y = nucleotide_id(b), z = nucleotide_id(c);
    if (x == 4 || y == 4 || z == 4) return 64;  // marginalise
    return static_cast<uint8_t>(16 * x + 4 * y + z);
}

inline char complement(char c)
{
    switch (c) {
        case 'A': return 'T'; case 'a': return 't';
        case 'C': return 'G'; case 'c': return 'g';
        case 'G': return 'C'; case 'g': return 'c';
        case 'T': return 'A'; case 't': return 'a';
        default: return c;
    }
}

inline void reverse_complement(std::vector<std::string>& rows)
{
    for (auto& r : rows) {
        std::string out(r.size(), ' ');
        for (size_t i = 0; i < r.size(); ++i) out[r.size() - 1 - i] = complement(r[i]);
        r.swap(out);
    }
}

// Frame arithmetic of update_seqs: shared with the device front end (csrc/frames.hpp)
using pcsf::FrameView;
using pcsf::frame_view;

// appends the codons of one frame, leaf-major, into dst[leaf*stride + offset ...]
inline void extract_codons(const std::vector<std::string>& rows, unsigned skip, size_t n_codons, uint8_t* dst, size_t stride, size_t offset)
{
    for (size_t leaf = 0; leaf < rows.size(); ++leaf) {
        const char* s = rows[leaf].data() + skip;
        uint8_t* d = dst + leaf * stride + offset;
        for (size_t c = 0; c < n_codons; ++c) d[c] = codon_id(s[3 * c], s[3 * c + 1], s[3 * c + 2]);
    }
}

// ---- branch-length score ---------------------------------------------------------------------
// For a column, S = leaves showing A/C/G/T.  bl(S) = sum of the branch lengths below the last
// common ancestor of S that lead to at least one member of S, accumulated in the order
// (own branch, left subtree, right subtree) in double precision -- the order matters because the
// result feeds a float threshold.  Leaves are numbered left to right, so a subtree is a range.
class BranchLengthScore {
public:
    explicit BranchLengthScore(const Model& m) : m_(m), lo_(m.n_nodes), hi_(m.n_nodes), prefix_(m.n_leaves + 1)
    {
        for (int i = 0; i < m.n_nodes; ++i) {
            if (i < m.n_leaves) { lo_[i] = i; hi_[i] = i + 1; }
            else { lo_[i] = lo_[m.child1[i]]; hi_[i] = hi_[m.child2[i]]; }
        }
        words_ = (static_cast<size_t>(m.n_leaves) + 63) / 64;
        range_.assign(static_cast<size_t>(m.n_nodes) * words_, 0);
        for (int i = 0; i < m.n_nodes; ++i)
            for (int l = lo_[i]; l < hi_[i]; ++l) range_[i * words_ + (l >> 6)] |= uint64_t(1) << (l & 63);
        std::vector<char> all(m.n_leaves, 1);
        total_ = subtree_sum(all);
    }
    double total() const { return total_; }

    // tests: the vector and the scalar character classifiers agree on n random rows (returns the number of mismatches)
    static size_t selftest_presence_bits(size_t n_rows)
    {
        size_t bad = 0;
        uint64_t state = 0x2545f4914f6cdd1dull;
        const char alphabet[] = "ACGTacgtNn-.ACGTacgt";
        for (size_t r = 0; r < n_rows; ++r) {
            state ^= state << 13; state ^= state >> 7; state ^= state << 17;
            const size_t len = 1 + state % 300;
            std::vector<unsigned char> row(len);
            for (size_t i = 0; i < len; ++i) {
                state ^= state << 13; state ^= state >> 7; state ^= state << 17;
                row[i] = static_cast<unsigned char>(alphabet[state % 20]);
            }
            if (r % 7 == 0) row[state % len] = static_cast<unsigned char>("XRY*@m "[state % 7]);  // not allowed
            std::vector<uint64_t> a((len + 63) / 64, 0), b((len + 63) / 64, 0);
            const bool oka = presence_bits(row.data(), len, a.data()), okb = presence_bits_scalar(row.data(), len, b.data());
            if (oka != okb || a != b) ++bad;
        }
        return bad;
    }

    // present[leaf] != 0 for members of S; requires |S| >= 1
    double subtree_sum(const std::vector<char>& present)
    {
        prefix_[0] = 0;
        for (int l = 0; l < m_.n_leaves; ++l) prefix_[l + 1] = prefix_[l] + (present[l] ? 1 : 0);
        return walk(m_.n_nodes - 1, false);
    }

    // per-base scores bl(S)/bl(all) (0 when fewer than two species have a base); returns the mean.
    // The species set of a column is a bit mask built row by row (sequential reads of each species' row, one table
    // look-up per character); the tree is walked only when the mask differs from the previous column's.
    double per_base(const Alignment& aln, std::vector<double>* out)
    {
        const size_t L = aln.length();
        const int nl = m_.n_leaves;
        const size_t words = (static_cast<size_t>(nl) + 63) / 64;
        masks_.assign(L * words, 0);
        static const CharClass cls;
        // (1) per species, one presence bit per column (64 columns per word): 32 characters per step with AVX2
        const size_t col_words = (L + 63) / 64;
        bits_.assign(static_cast<size_t>(nl) * col_words, 0);
        for (int l = 0; l < nl; ++l) {
            const unsigned char* row = reinterpret_cast<const unsigned char*>(aln.seqs[l].data());
            if (!presence_bits(row, L, bits_.data() + static_cast<size_t>(l) * col_words))
                for (size_t i = 0; i < L; ++i) nucleotide_id(static_cast<char>(row[i]));  // throws at the first offending character
        }
        // (2) 64 x 64 bit transposes turn 64 species words into 64 column masks
        for (size_t cw = 0; cw < col_words; ++cw) {
            for (size_t w = 0; w < words; ++w) {
                uint64_t blk[64];
                for (int r = 0; r < 64; ++r) {
                    const size_t l = w * 64 + static_cast<size_t>(r);
                    blk[r] = l < static_cast<size_t>(nl) ? bits_[l * col_words + cw] : 0;
                }
                transpose64(blk);
                const size_t n_cols = std::min<size_t>(64, L - cw * 64);
                for (size_t c = 0; c < n_cols; ++c) masks_[(cw * 64 + c) * words + w] = blk[c];
            }
        }
        std::vector<uint64_t> last(words, ~uint64_t(0));
        bool have_last = false;
        double last_val = 0.0, sum = 0.0;
        if (out) {
            out->clear();
            out->reserve(L);
        }
        for (size_t i = 0; i < L; ++i) {
            const uint64_t* mk = masks_.data() + i * words;
            int cnt = 0;
            for (size_t w = 0; w < words; ++w) cnt += __builtin_popcountll(mk[w]);
            double v = 0.0;
            if (cnt >= 2) {
                bool same = have_last;
                for (size_t w = 0; w < words && same; ++w) same = mk[w] == last[w];
                if (!same) {
                    last_val = sum_for_mask(mk);
                    for (size_t w = 0; w < words; ++w) last[w] = mk[w];
                    have_last = true;
                }
                v = last_val;
                sum += v;
            }
            if (out) out->push_back(cnt >= 2 ? v / total_ : 0.0);
        }
        return sum / (total_ * static_cast<double>(L));
    }

private:
    // bl(S) for a species set given as a bit mask: the same walk (own branch, left, right) with the overlaps taken
    // from popcounts, and a memo of the sets seen so far (alignments repeat a small number of presence patterns)
    int overlap_mask(int node, const uint64_t* mk) const
    {
        int c = 0;
        for (size_t w = 0; w < words_; ++w) c += __builtin_popcountll(mk[w] & range_[static_cast<size_t>(node) * words_ + w]);
        return c;
    }
    double walk_mask(int node, bool below_lca, const uint64_t* mk, int here) const
    {
        if (node < m_.n_leaves) return m_.branch_len_d[node];
        const int l = m_.child1[node], r = m_.child2[node];
        const int ol = overlap_mask(l, mk), orr = here - ol;
        double bl = below_lca ? m_.branch_len_d[node] : 0.0;
        if (ol > 0 && orr > 0) below_lca = true;
        if (ol > 0) bl += walk_mask(l, below_lca, mk, ol);
        if (orr > 0) bl += walk_mask(r, below_lca, mk, orr);
        return bl;
    }
    double sum_for_mask(const uint64_t* mk)
    {
        if (words_ > 2) return walk_mask(m_.n_nodes - 1, false, mk, overlap_mask(m_.n_nodes - 1, mk));
        const uint64_t lo = mk[0], hi = words_ > 1 ? mk[1] : 0;
        uint64_t x = lo * 0x9e3779b97f4a7c15ull ^ (hi + 0x7f4a7c159e3779b9ull) * 0xc2b2ae3d27d4eb4full;
        x ^= x >> 29;
        if (memo_.empty()) memo_.resize(MEMO_SLOTS);
        MemoSlot& slot = memo_[x & (MEMO_SLOTS - 1)];  // direct-mapped: a set seen recently is found with one probe
        if (slot.used && slot.lo == lo && slot.hi == hi) return slot.value;
        const double v = walk_mask(m_.n_nodes - 1, false, mk, overlap_mask(m_.n_nodes - 1, mk));
        slot = MemoSlot{lo, hi, v, true};
        return v;
    }
    struct MemoSlot {
        uint64_t lo = 0, hi = 0;
        double value = 0.0;
        bool used = false;
    };
    static constexpr size_t MEMO_SLOTS = 1 << 13;
    std::vector<MemoSlot> memo_;
    size_t words_ = 1;
    std::vector<uint64_t> range_;  // [node][words_] leaves below the node
    struct CharClass {  // 0..3 = A C G T (either case), 4 = N n - . , 8 = not allowed in an alignment
        unsigned char id[256];
        CharClass()
        {
            for (int c = 0; c < 256; ++c) id[c] = 8;
            id['A'] = id['a'] = 0;
            id['C'] = id['c'] = 1;
            id['G'] = id['g'] = 2;
            id['T'] = id['t'] = 3;
            id['N'] = id['n'] = id['-'] = id['.'] = 4;
        }
    };
    std::vector<uint64_t> masks_, bits_;

    // out bit i = row[i] is one of ACGTacgt; returns false if a character outside ACGTNacgtn-. occurs
    static bool presence_bits_scalar(const unsigned char* row, size_t n, uint64_t* out)
    {
        static const CharClass cls;
        unsigned bad = 0;
        for (size_t i = 0; i < n; ++i) {
            const unsigned c = cls.id[row[i]];
            bad |= c;
            if (c <= 3) out[i >> 6] |= uint64_t(1) << (i & 63);
        }
        return !(bad & 8);
    }
#if defined(__x86_64__)
    __attribute__((target("avx2"))) static bool presence_bits_avx2(const unsigned char* row, size_t n, uint64_t* out)
    {
        const __m256i up = _mm256_set1_epi8(static_cast<char>(0xDF));
        const __m256i cA = _mm256_set1_epi8('A'), cC = _mm256_set1_epi8('C'), cG = _mm256_set1_epi8('G'), cT = _mm256_set1_epi8('T');
        const __m256i cN = _mm256_set1_epi8('N'), cDash = _mm256_set1_epi8('-'), cDot = _mm256_set1_epi8('.');
        uint32_t all_ok = 0xffffffffu;
        size_t i = 0;
        for (; i + 32 <= n; i += 32) {
            const __m256i x = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(row + i));
            const __m256i u = _mm256_and_si256(x, up);  // letters to upper case (other characters change, they are tested on x)
            const __m256i acgt = _mm256_or_si256(_mm256_or_si256(_mm256_cmpeq_epi8(u, cA), _mm256_cmpeq_epi8(u, cC)),
                                                 _mm256_or_si256(_mm256_cmpeq_epi8(u, cG), _mm256_cmpeq_epi8(u, cT)));
            const __m256i other = _mm256_or_si256(_mm256_cmpeq_epi8(u, cN), _mm256_or_si256(_mm256_cmpeq_epi8(x, cDash), _mm256_cmpeq_epi8(x, cDot)));
            const uint32_t m = static_cast<uint32_t>(_mm256_movemask_epi8(acgt));
            all_ok &= m | static_cast<uint32_t>(_mm256_movemask_epi8(other));
            out[i >> 6] |= static_cast<uint64_t>(m) << (i & 63);
        }
        bool ok = all_ok == 0xffffffffu;
        if (i < n) {  // tail; bit positions continue at i
            static const CharClass cls;
            unsigned bad = 0;
            for (; i < n; ++i) {
                const unsigned c = cls.id[row[i]];
                bad |= c;
                if (c <= 3) out[i >> 6] |= uint64_t(1) << (i & 63);
            }
            ok = ok && !(bad & 8);
        }
        return ok;
    }
#endif
    static bool presence_bits(const unsigned char* row, size_t n, uint64_t* out)
    {
#if defined(__x86_64__)
        static const bool avx2 = __builtin_cpu_supports("avx2");
        if (avx2) return presence_bits_avx2(row, n, out);
#endif
        return presence_bits_scalar(row, n, out);
    }
    // in-place transpose of a 64 x 64 bit matrix (row r = a[r], column c = bit c): six butterfly stages
    static void transpose64(uint64_t a[64])
    {
        uint64_t m = 0x00000000ffffffffull;
        for (int j = 32; j != 0; j >>= 1, m ^= m << j)
            for (int k = 0; k < 64; k = (k + j + 1) & ~j) {
                const uint64_t t = ((a[k] >> j) ^ a[k + j]) & m;
                a[k] ^= t << j;
                a[k + j] ^= t;
            }
    }
    int overlap(int node) const { return prefix_[hi_[node]] - prefix_[lo_[node]]; }
    double walk(int node, bool below_lca) const
    {
        if (node < m_.n_leaves) return m_.branch_len_d[node];
        const int l = m_.child1[node], r = m_.child2[node];
        const int ol = overlap(l), orr = overlap(node) - ol;
        double bl = below_lca ? m_.branch_len_d[node] : 0.0;
        if (ol > 0 && orr > 0) below_lca = true;
        if (ol > 0) bl += walk(l, below_lca);
        if (orr > 0) bl += walk(r, below_lca);
        return bl;
    }
    const Model& m_;
    std::vector<int> lo_, hi_, prefix_;
    double total_ = 0.0;
};

// ---- text output -----------------------------------------------------------------------------
// format_trimmed<D>(dst, v): what print_trimmed writes for "%.<D>f" (D = 3 or 4), without the newline, ~15x
// faster than snprintf.  Exact: a float times 10^D (D <= 4) is exact in double (24 + 10 significant bits), and
// rounding to nearest-even on that exact value is what glibc's printf does.  Values below 1000 in magnitude (all scores
// of a track, in practice) take a branch-free path: three-digit groups from a table, lengths by comparison -- the sign,
// the number of integer digits and the trailing zeros are all data-dependent, and as branches each cost a
// misprediction every few values.  Falls back to snprintf for NaN, infinities and |v| >= 9e14.  Returns the number of
// characters written; dst needs room for 64.
struct Digits3 {
    char d[3];
    uint8_t trimmed;   // digits left when trailing zeros go, at least one
};
struct Digits3Table {
    Digits3 e[1000];
    Digits3Table()
    {
        for (int i = 0; i < 1000; ++i) {
            e[i].d[0] = static_cast<char>('0' + i / 100);
            e[i].d[1] = static_cast<char>('0' + i / 10 % 10);
            e[i].d[2] = static_cast<char>('0' + i % 10);
            e[i].trimmed = static_cast<uint8_t>(i % 10 ? 3 : (i / 10 % 10 ? 2 : 1));
        }
    }
};
inline const Digits3Table DIGITS3;
static_assert(__BYTE_ORDER__ == __ORDER_LITTLE_ENDIAN__, "format_trimmed shifts digit bytes inside a word");

// the text of sign and |value| * 10^D = scaled (< 1000 * 10^D) with trailing zeros removed but one decimal kept: format_trimmed's
// branch-free path, also used where the scaled integer is what a caller holds (the smoothing pass's printed-value bins)
template <int D>
inline size_t format_scaled(char* dst, bool negative, uint64_t scaled)
{
    constexpr uint64_t POW = D == 3 ? 1000 : 10000;
    {
        const unsigned ip = static_cast<unsigned>(scaled / POW), frac = static_cast<unsigned>(scaled % POW);
        char* p = dst;
        *p = '-';
        p += negative ? 1 : 0;
        const unsigned nd = 1u + (ip >= 10u) + (ip >= 100u);
        uint32_t w;
        memcpy(&w, &DIGITS3.e[ip], 4);                                    // digits in bytes 0..2 (x86 is little-endian: byte 0 is written first)
        w = ((w & 0x00ffffffu) | (uint32_t('.') << 24)) >> (8 * (3 - nd));   // the last nd digits, then the point
        memcpy(p, &w, 4);                                                 // (up to three bytes beyond them are rewritten below)
        p += nd + 1;
        if (D == 3) {
            const Digits3& F = DIGITS3.e[frac];
            memcpy(p, &F, 4);
            p += F.trimmed;
        } else {
            const Digits3& F = DIGITS3.e[frac / 10];
            const unsigned last = frac % 10;
            memcpy(p, &F, 4);
            p[3] = static_cast<char>('0' + last);
            p += last ? 4u : F.trimmed;
        }
        *p = 0;
        return static_cast<size_t>(p - dst);
    }
}

template <int D>
inline size_t format_trimmed(char* dst, float v)
{
    static_assert(D == 3 || D == 4, "scale must stay exact in double");
    const double x = static_cast<double>(v);
    if (!(std::fabs(x) < 9e14)) {  // also NaN (9e14 * 10^4 still fits the signed 64-bit conversion below)
        char fmt[8];
        snprintf(fmt, sizeof fmt, "%%.%df", D);
        int n = snprintf(dst, 64, fmt, x);
        for (int i = n; i >= 0; --i) {  // same trimming as print_trimmed
            if (dst[i] == '.') {
                dst[i + 1] = '0';
                break;
            }
            if (dst[i] >= '0' && dst[i] <= '9') {
                if (dst[i] != '0') break;
                dst[i] = 0;
            }
        }
        return strlen(dst);
    }
    constexpr uint64_t POW = D == 3 ? 1000 : 10000;
    // round to nearest, ties to even, on the exact product: cvtsd2si under the default MXCSR does what nearbyint() does, in one
    // instruction instead of a libm call (stage 3 formats ~2.3 values per reference base)
#if defined(__x86_64__)
    const uint64_t scaled = static_cast<uint64_t>(_mm_cvtsd_si64(_mm_set_sd(std::fabs(x) * static_cast<double>(POW))));
#else
    const uint64_t scaled = static_cast<uint64_t>(std::nearbyint(std::fabs(x) * static_cast<double>(POW)));
#endif
    if (scaled < 1000 * POW) return format_scaled<D>(dst, std::signbit(x), scaled);
    uint64_t ip = scaled / POW;
    unsigned frac = static_cast<unsigned>(scaled % POW);
    char* p = dst;
    if (std::signbit(x)) *p++ = '-';
    char tmp[24];
    int n = 0;
    do {
        tmp[n++] = static_cast<char>('0' + ip % 10);
        ip /= 10;
    } while (ip);
    while (n) *p++ = tmp[--n];
    *p++ = '.';
    char fr[D];
    for (int i = D - 1; i >= 0; --i) {
        fr[i] = static_cast<char>('0' + frac % 10);
        frac /= 10;
    }
    int last = D - 1;
    while (last > 0 && fr[last] == '0') --last;  // keep at least one decimal
    for (int i = 0; i <= last; ++i) *p++ = fr[i];
    *p = 0;
    return static_cast<size_t>(p - dst);
}


// "%.3f"-style formatting of a float with trailing zeros removed but one decimal kept
inline void print_trimmed(FILE* f, const char* fmt, float v)
{
    char buf[64];
    snprintf(buf, sizeof buf, fmt, static_cast<double>(v));
    for (int i = static_cast<int>(strlen(buf)); i >= 0; --i) {
        if (buf[i] == '.') {
            buf[i + 1] = '0';
            break;
        }
        if (buf[i] >= '0' && buf[i] <= '9') {
            if (buf[i] != '0') break;
            buf[i] = 0;
        }
    }
    fputs(buf, f);
    fputc('\n', f);
}

}  // namespace pcsf_host
