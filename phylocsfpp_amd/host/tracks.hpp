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
        constexpr size_t CHUNK = 2048;  // columns per pass: the chunk's masks stay in L1 while every species' row streams by
        for (size_t i0 = 0; i0 < L; i0 += CHUNK) {
            const size_t i1 = std::min(L, i0 + CHUNK);
            for (int l = 0; l < nl; ++l) {
                const unsigned char* row = reinterpret_cast<const unsigned char*>(aln.seqs[l].data());
                uint64_t* mw = masks_.data() + (l >> 6);
                const uint64_t bit = uint64_t(1) << (l & 63);
                unsigned bad = 0;
                for (size_t i = i0; i < i1; ++i) {
                    const unsigned c = cls.id[row[i]];
                    bad |= c;
                    mw[i * words] |= c <= 3 ? bit : 0;
                }
                if (bad & 8) {  // report the first offending character like nucleotide_id() does
                    for (size_t i = i0; i < i1; ++i) nucleotide_id(static_cast<char>(row[i]));
                }
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
        if (words_ <= 2) {
            const Key k{mk[0], words_ > 1 ? mk[1] : 0};
            auto it = memo_.find(k);
            if (it != memo_.end()) return it->second;
            const double v = walk_mask(m_.n_nodes - 1, false, mk, overlap_mask(m_.n_nodes - 1, mk));
            if (memo_.size() < (size_t(1) << 20)) memo_.emplace(k, v);
            return v;
        }
        return walk_mask(m_.n_nodes - 1, false, mk, overlap_mask(m_.n_nodes - 1, mk));
    }
    struct Key {
        uint64_t lo, hi;
        bool operator==(const Key& o) const { return lo == o.lo && hi == o.hi; }
    };
    struct KeyHash {
        size_t operator()(const Key& k) const
        {
            uint64_t x = k.lo * 0x9e3779b97f4a7c15ull ^ (k.hi + 0x7f4a7c159e3779b9ull) * 0xc2b2ae3d27d4eb4full;
            return static_cast<size_t>(x ^ (x >> 29));
        }
    };
    std::unordered_map<Key, double, KeyHash> memo_;
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
    std::vector<uint64_t> masks_;
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
// format_trimmed<D>(dst, v): what print_trimmed writes for "%.<D>f" (D = 3 or 4), without the newline, 10x
// faster than snprintf.  Exact: a float times 10^D (D <= 4) is exact in double (24 + 10 significant bits), and
// nearbyint() in the default rounding mode rounds ties to even on that exact value, as glibc's printf does.
// Falls back to snprintf for NaN, infinities and |v| >= 1e15.  Returns the number of characters written.
template <int D>
inline size_t format_trimmed(char* dst, float v)
{
    static_assert(D == 3 || D == 4, "scale must stay exact in double");
    const double x = static_cast<double>(v);
    if (!(std::fabs(x) < 1e15)) {  // also NaN
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
    const uint64_t scaled = static_cast<uint64_t>(std::nearbyint(std::fabs(x) * static_cast<double>(POW)));
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
