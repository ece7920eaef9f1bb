// maf_reader.hpp -- sequential MAF reader with the column-indexing semantics of the reference's
// parallel_maf_reader::get_next_alignment (src/parallel_file_reader.hpp:430-685), restated:
//   * only "a" and "s" lines matter; species = text before the first '.', lower-cased, looked up
//     in the model's name map; unknown rows are skipped with a one-time warning
//   * the first mapped row of a block is the reference row
//   * build-tracks mode (concatenate): the next block is appended while its reference row
//     continues exactly where the running alignment ends (same chromosome, start0 == start-1 +
//     sum of reference sizes); species missing from a block are padded with 'N'
//   * a block whose reference interval crosses a multiple of 1,000,000 marks a break point:
//     reading continues until >= 2 more reference bases are available, the alignment is cut to
//     (length at the end of the break-point block + 2) and the NEXT alignment restarts at the
//     block after the break-point block
//   * afterwards every column where the reference row has '-' is removed from all rows
//   * score-msa mode: every block alone, any strand
// Byte-range jobs (parallel ingest), as in the reference (src/parallel_file_reader.hpp:283-350,396-425): a
// job owns the blocks whose "\na " begins inside its range; a job other than the first steps back to the
// block before its first one, reads (and discards) the alignment that starts there -- it may swallow some of
// the job's leading blocks, which the previous job delivers -- and continues after it.  An alignment that
// starts inside a range is read to its end even when that lies beyond the range.  The concatenation of all
// jobs' alignments does not depend on the split (tests/test_host_layer.py).
#pragma once

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "model_io.hpp"

#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace pcsf_host {

constexpr uint64_t BREAKPOINT_INTERVAL = 1000000;

// ---- byte scanners of the text path (round 5: stage 1 of build-tracks was ~400 ns of thread time per ~240-byte line -- two libc calls and
// two byte-wise walks per line; reference: the tokenizer of src/parallel_file_reader.hpp:180-245 is a malloc + strtok_r per row) ----
namespace scan {
// first '\n' or ' ' in [p, e), or e: ONE pass over a row's sequence finds its end (and a stray space, which ends the token)
#if defined(__x86_64__)
__attribute__((target("avx2"))) inline const char* newline_or_space_avx2(const char* p, const char* e)
{
    const __m256i nl = _mm256_set1_epi8('\n'), sp = _mm256_set1_epi8(' ');
    while (p + 32 <= e) {
        const __m256i v = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(p));
        const unsigned m = static_cast<unsigned>(_mm256_movemask_epi8(_mm256_or_si256(_mm256_cmpeq_epi8(v, nl), _mm256_cmpeq_epi8(v, sp))));
        if (m) return p + __builtin_ctz(m);
        p += 32;
    }
    while (p < e && *p != '\n' && *p != ' ') ++p;
    return p;
}
// first position q in [p, e) with q[0] == '\n' and q[1] == 'a' (the line AFTER q starts a block), or e
__attribute__((target("avx2"))) inline const char* newline_a_avx2(const char* p, const char* e)
{
    const __m256i nl = _mm256_set1_epi8('\n'), a = _mm256_set1_epi8('a');
    while (p + 33 <= e) {
        const __m256i v0 = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(p)), v1 = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(p + 1));
        const unsigned m = static_cast<unsigned>(_mm256_movemask_epi8(_mm256_and_si256(_mm256_cmpeq_epi8(v0, nl), _mm256_cmpeq_epi8(v1, a))));
        if (m) return p + __builtin_ctz(m);
        p += 32;
    }
    while (p + 1 < e && !(p[0] == '\n' && p[1] == 'a')) ++p;
    return p + 1 < e ? p : e;
}
inline bool have_avx2()
{
    static const bool yes = __builtin_cpu_supports("avx2");
    return yes;
}
#endif
inline const char* newline_or_space(const char* p, const char* e)
{
#if defined(__x86_64__)
    if (have_avx2()) return newline_or_space_avx2(p, e);
#endif
    while (p < e && *p != '\n' && *p != ' ') ++p;
    return p;
}
inline const char* newline_a(const char* p, const char* e)
{
#if defined(__x86_64__)
    if (have_avx2()) return newline_a_avx2(p, e);
#endif
    while (p + 1 < e && !(p[0] == '\n' && p[1] == 'a')) ++p;
    return p + 1 < e ? p : e;
}
}  // namespace scan

struct Alignment {
    std::string chrom;
    uint64_t start_pos = 0;  // 1-based
    uint64_t chrom_len = 0;
    char strand = '+';
    std::vector<std::string> seqs;  // one row per leaf, equal lengths, reference gaps removed
    size_t length() const { return seqs.empty() ? 0 : seqs[0].size(); }
};

// An alignment located and measured but not read (MafReader::measure), so that its rows can be parsed straight into their place in a
// batch matrix (MafReader::fill) once the alignments before it -- other jobs' too -- have been given their columns.
struct AlignmentSpan {
    size_t begin = 0;        // byte offset of the "a" line of its first block
    uint32_t n_blocks = 0;   // blocks whose rows it takes (the block that ends it is not one of them)
    uint32_t len = 0;        // columns: reference gaps removed, cut at a break point
    uint32_t kept = 0;       // columns before that cut (len > kept only for a MAF whose size fields contradict its rows)
    uint64_t start_pos = 0;  // 1-based
    uint64_t chrom_len = 0;
    std::string chrom;
    char strand = '+';
};

// read-only mapping of a MAF file, shared by the readers of all byte-range jobs
class MafFile {
public:
    explicit MafFile(const std::string& path)
    {
        fd_ = ::open(path.c_str(), O_RDONLY);
        if (fd_ < 0) throw std::runtime_error("cannot open " + path);
        struct stat st;
        if (fstat(fd_, &st) != 0) throw std::runtime_error("cannot stat " + path);
        size_ = static_cast<size_t>(st.st_size);
        if (size_ > 0) {
            void* p = mmap(nullptr, size_, PROT_READ, MAP_PRIVATE, fd_, 0);
            if (p == MAP_FAILED) throw std::runtime_error("cannot mmap " + path);
            data_ = static_cast<const char*>(p);
        }
    }
    ~MafFile()
    {
        if (data_) munmap(const_cast<char*>(data_), size_);
        if (fd_ >= 0) ::close(fd_);
    }
    MafFile(const MafFile&) = delete;
    MafFile& operator=(const MafFile&) = delete;
    const char* data() const { return data_; }
    size_t size() const { return size_; }
    int fd() const { return fd_; }   // for positional reads that copy from the page cache without mapping it

private:
    int fd_ = -1;
    const char* data_ = nullptr;
    size_t size_ = 0;
};

class MafReader {
public:
    // whole file
    MafReader(const std::string& path, const std::unordered_map<std::string, int>& name_to_leaf, int n_leaves, bool concatenate,
              bool verbose = true)
        : names_(name_to_leaf), n_leaves_(n_leaves), concatenate_(concatenate), verbose_(verbose), own_(new MafFile(path))
    {
        init(*own_, 0, own_->size());
    }
    // byte-range job [range_begin, range_end) of an already mapped file
    MafReader(const MafFile& file, size_t range_begin, size_t range_end, const std::unordered_map<std::string, int>& name_to_leaf,
              int n_leaves, bool concatenate, bool verbose = true)
        : names_(name_to_leaf), n_leaves_(n_leaves), concatenate_(concatenate), verbose_(verbose)
    {
        init(file, range_begin, range_end);
    }
    MafReader(const MafReader&) = delete;
    MafReader& operator=(const MafReader&) = delete;

    size_t bytes_total() const { return size_; }
    size_t bytes_done() const { return pos_; }

    bool next(Alignment& aln)
    {
        // a block belongs to this job when the newline before its "a " lies inside the range
        if (pos_ >= size_ || (pos_ > 0 && pos_ - 1 >= end_)) return false;
        if (!scan_only_) aln.seqs.assign(n_leaves_, std::string());
        bool first_block = true, stop = !concatenate_, reached_bp = false, freeze_restart = false;
        size_t restart_pos = pos_;
        int ref_leaf = -1;
        uint64_t ref_bases = 0, ref_bases_at_bp = 0;
        size_t block_base = 0;   // length of every row before the current block

        while ((!stop || first_block) && pos_ < size_) {
            if (!freeze_restart) {
                if (reached_bp) freeze_restart = true;
                restart_pos = pos_;
            }
            skip_line();  // the "a ..." line
            int block_ref = -1;
            if (reached_bp && ref_bases >= ref_bases_at_bp + 2) stop = true;

            while ((!stop || first_block) && pos_ < size_ && data_[pos_] != 'a') {
                if (data_[pos_] != 's') {
                    skip_line();
                    continue;
                }
                SLine s = parse_s_line();
                if (!first_block && block_ref == -1 && !((aln.start_pos - 1) + ref_bases == s.start() && s.chrom_is(aln.chrom))) {
                    stop = true;  // not contiguous: this block starts the next alignment
                    break;
                }
                const int leaf = leaf_of(s);
                if (leaf < 0) continue;  // not part of the model
                if (first_block && ref_leaf == -1) {
                    aln.start_pos = s.start() + 1;
                    aln.chrom = s.chrom_str();
                    aln.chrom_len = s.src_size();
                    aln.strand = s.strand;
                    ref_leaf = leaf;
                    const uint64_t ref_size = s.size();
                    ref_bases = ref_size;
                    check_strand(s);
                    if (!reached_bp && aln.start_pos / BREAKPOINT_INTERVAL < (aln.start_pos + ref_size) / BREAKPOINT_INTERVAL) {
                        reached_bp = true;
                        ref_bases_at_bp = ref_bases;
                    }
                } else if (!first_block && block_ref == -1) {
                    block_ref = leaf;
                    const uint64_t prev_end = aln.start_pos + ref_bases;
                    const uint64_t block_size = s.size();
                    ref_bases += block_size;
                    if (!reached_bp && prev_end / BREAKPOINT_INTERVAL < (prev_end + block_size) / BREAKPOINT_INTERVAL) {
                        reached_bp = true;
                        ref_bases_at_bp = ref_bases;
                    }
                    if (block_ref != ref_leaf)
                        throw std::runtime_error("alignment block does not start with the reference species: " + s.species_str() + "." + s.chrom_str());
                    check_strand(s);
                }
                if (scan_only_) {  // the block's reference row has been seen: nothing else in it decides where the alignment ends
                    if (measuring_) {   // ... and its non-gap characters are the columns the block adds
                        size_t gaps = 0;
                        for (size_t i = 0; i < s.text_len; ++i) gaps += s.text[i] == '-';
                        measuring_->kept += static_cast<uint32_t>(s.text_len - gaps);
                        ++measuring_->n_blocks;
                    }
                    skip_to_next_block();
                    break;
                }
                // malformed blocks are refused, by both forms of the reader alike (fill() below): a second row of a species, a row longer
                // than its block's reference row.  A SHORTER row is padded with 'N' like a missing one (both forms).
                if (aln.seqs[leaf].size() > block_base) throw std::runtime_error("species " + s.species_str() + " has two rows in one alignment block");
                aln.seqs[leaf].append(s.text, s.text_len);
                if (aln.seqs[leaf].size() > aln.seqs[ref_leaf].size())
                    throw std::runtime_error("the row of species " + s.species_str() + " is longer than its block's reference row");
            }
            if (ref_leaf < 0) throw std::runtime_error("alignment block without any species of the model");
            first_block = false;
            if (scan_only_) continue;
            const size_t L = aln.seqs[ref_leaf].size();
            for (auto& row : aln.seqs)
                if (row.size() != L) row.append(L - row.size(), 'N');
            block_base = L;
        }
        if (reached_bp && ref_bases >= ref_bases_at_bp + 2) stop = true;
        if (stop && concatenate_) pos_ = restart_pos;
        if (scan_only_) {
            if (measuring_) {
                measuring_->start_pos = aln.start_pos;
                measuring_->chrom_len = aln.chrom_len;
                measuring_->strand = aln.strand;
                measuring_->chrom = aln.chrom;
                measuring_->len = (reached_bp && ref_bases > ref_bases_at_bp + 2) ? static_cast<uint32_t>(ref_bases_at_bp + 2) : measuring_->kept;
            }
            return true;
        }

        // remove the columns where the reference row is a gap
        // (the kept columns as runs between the gaps: every row is compacted with one memmove per run)
        const std::string& ref = aln.seqs[ref_leaf];
        runs_.clear();
        size_t kept = 0;
        for (size_t i = 0, n = ref.size(); i < n;) {
            const void* g = memchr(ref.data() + i, '-', n - i);
            const size_t j = g ? static_cast<size_t>(static_cast<const char*>(g) - ref.data()) : n;
            if (j > i) runs_.push_back({i, j - i});
            kept += j - i;
            i = j;
            while (i < n && ref[i] == '-') ++i;
        }
        if (kept != ref.size()) {
            for (auto& row : aln.seqs) {
                size_t w = 0;
                for (const auto& r : runs_) {
                    if (r.first != w) memmove(&row[w], &row[r.first], r.second);
                    w += r.second;
                }
                row.resize(kept);
            }
        }
        if (reached_bp && ref_bases > ref_bases_at_bp + 2)
            for (auto& row : aln.seqs) row.resize(ref_bases_at_bp + 2);
        return true;
    }

    // The next alignment of this job as next() would deliver it, measured instead of read: where it starts, how many blocks it takes
    // and how many columns it will have.  Only the blocks' reference rows are looked at.
    bool measure(AlignmentSpan& sp)
    {
        if (pos_ >= size_ || (pos_ > 0 && pos_ - 1 >= end_)) return false;
        sp = AlignmentSpan();
        sp.begin = pos_;
        scratch_.seqs.clear();
        measuring_ = &sp;
        scan_only_ = true;
        bool ok = false;
        try {
            ok = next(scratch_);
        } catch (...) {
            scan_only_ = false;
            measuring_ = nullptr;
            throw;
        }
        scan_only_ = false;
        measuring_ = nullptr;
        return ok;
    }

    // The rows of a measured alignment, written where they belong: row `leaf` at base + leaf * stride, sp.len characters each.  What
    // next() builds in strings -- rows appended block by block, species missing from a block padded with 'N', the reference's gap columns
    // removed, the cut at a break point -- is done per block on the way: the block's first mapped row is its reference row (checked by
    // measure()), so the columns to keep are known before any other row of the block is copied.  Same bytes as next() (tests/test_host_layer.py).
    void fill(const AlignmentSpan& sp, char* base, size_t stride)
    {
        const size_t keep_pos = pos_;
        pos_ = sp.begin;
        const size_t cap = std::min<size_t>(sp.len, sp.kept);   // columns that come from the rows
        filled_.assign(static_cast<size_t>(n_leaves_), 0);
        size_t written = 0;   // columns of the blocks before the current one
        for (uint32_t b = 0; b < sp.n_blocks && pos_ < size_; ++b) {
            skip_line();  // the "a ..." line
            bool have_ref = false;
            int ref_row_leaf = -1;
            const char* ref_text = nullptr;
            size_t ref_text_len = 0;
            size_t block_kept = 0;
            while (pos_ < size_ && data_[pos_] != 'a') {
                if (data_[pos_] != 's') {
                    skip_line();
                    continue;
                }
                const SLine s = parse_s_line();
                const int leaf = leaf_of(s);
                if (leaf < 0) continue;
                if (!have_ref) {   // the reference row: the runs of columns that stay
                    have_ref = true;
                    ref_row_leaf = leaf;
                    ref_text = s.text;
                    ref_text_len = s.text_len;
                    runs_.clear();
                    for (size_t i = 0; i < s.text_len;) {
                        const void* g = memchr(s.text + i, '-', s.text_len - i);
                        const size_t j = g ? static_cast<size_t>(static_cast<const char*>(g) - s.text) : s.text_len;
                        if (j > i) runs_.push_back({i, j - i});
                        block_kept += j - i;
                        i = j;
                        while (i < s.text_len && s.text[i] == '-') ++i;
                    }
                }
                if (filled_[static_cast<size_t>(leaf)] > written || (leaf == ref_row_leaf && &s.text[0] != ref_text))
                    throw std::runtime_error("species " + s.species_str() + " has two rows in one alignment block");
                if (s.text_len > ref_text_len) throw std::runtime_error("the row of species " + s.species_str() + " is longer than its block's reference row");
                char* row = base + static_cast<size_t>(leaf) * stride;
                size_t at = written;
                for (const auto& r : runs_) {
                    if (at >= cap || r.first >= s.text_len) break;
                    const size_t n = std::min(std::min(r.second, s.text_len - r.first), cap - at);
                    memcpy(row + at, s.text + r.first, n);
                    at += n;
                    if (n < r.second) break;   // a row shorter than its block's reference row: padded below like a missing one
                }
                filled_[static_cast<size_t>(leaf)] = static_cast<uint32_t>(at);
            }
            written += block_kept;
            const size_t upto = std::min(written, cap);
            for (int l = 0; l < n_leaves_; ++l) {   // species missing from the block (or short of it)
                uint32_t& f = filled_[static_cast<size_t>(l)];
                if (f < upto) memset(base + static_cast<size_t>(l) * stride + f, 'N', upto - f);
                f = static_cast<uint32_t>(upto);
            }
        }
        if (sp.len > cap)   // (std::string::resize past the end: zero bytes, which no later stage accepts as a nucleotide)
            for (int l = 0; l < n_leaves_; ++l) memset(base + static_cast<size_t>(l) * stride + cap, 0, sp.len - cap);
        pos_ = keep_pos;
    }

private:
    void init(const MafFile& file, size_t range_begin, size_t range_end)
    {
        data_ = file.data();
        size_ = file.size();
        end_ = range_end < size_ ? range_end : size_;
        pos_ = range_begin < size_ ? range_begin : size_;
        // first alignment block: the file starts with "a " or we look for "\na "
        if (!(pos_ == 0 && size_ >= 2 && data_[0] == 'a' && data_[1] == ' ')) {
            const char* hit = pos_ < size_ ? static_cast<const char*>(memmem(data_ + pos_, size_ - pos_, "\na ", 3)) : nullptr;
            pos_ = hit ? static_cast<size_t>(hit - data_) + 1 : size_;
        }
        if (range_begin == 0 || pos_ >= size_ || pos_ < 2) return;
        // not the first job: the alignment that starts at the previous block is the previous job's
        size_t prev = pos_ - 1;  // the '\n' before this block's "a "
        bool found = false;
        while (prev > 0) {
            const void* nl = memrchr(data_, '\n', prev);
            const size_t line = nl ? static_cast<size_t>(static_cast<const char*>(nl) - data_) + 1 : 0;
            if (line + 1 < size_ && data_[line] == 'a' && data_[line + 1] == ' ') {
                prev = line;
                found = true;
                break;
            }
            if (line == 0) break;
            prev = line - 1;
        }
        if (!found) return;
        const size_t keep_end = end_;
        end_ = size_;
        pos_ = prev;
        // only where that alignment ends matters: walk its blocks looking at the reference rows alone
        Alignment scratch;
        const bool verbose = verbose_;
        verbose_ = false;
        scan_only_ = true;
        next(scratch);
        scan_only_ = false;
        verbose_ = verbose;
        end_ = keep_end;
    }

    struct SLine {  // views into the mapped file, except `species` (lower-cased copy in a buffer that is reused)
        const char* species = nullptr;  // as written in the file (the model's names are lower case)
        size_t species_len = 0;
        const char* chrom = nullptr;
        size_t chrom_len = 0;
        std::string species_str() const
        {
            std::string k(species, species_len);
            for (char& c : k) c = static_cast<char>(std::tolower(static_cast<unsigned char>(c)));
            return k;
        }
        std::string chrom_str() const { return std::string(chrom, chrom_len); }
        bool chrom_is(const std::string& c) const { return c.size() == chrom_len && memcmp(c.data(), chrom, chrom_len) == 0; }
        // start, size, srcSize where they stand in the line: converted when asked for -- only a block's reference row is (three numbers of
        // ~20 digits were a third of the time spent on every other row)
        const char* num[3] = {nullptr, nullptr, nullptr};
        size_t num_len[3] = {0, 0, 0};
        static uint64_t number(const char* p, size_t n)   // atoi semantics for the plain non-negative integers of a MAF
        {
            uint64_t v = 0;
            size_t i = 0;
            bool neg = false;
            if (i < n && (p[i] == '-' || p[i] == '+')) neg = p[i++] == '-';
            for (; i < n && p[i] >= '0' && p[i] <= '9'; ++i) v = v * 10 + static_cast<uint64_t>(p[i] - '0');
            return neg ? static_cast<uint64_t>(-static_cast<int64_t>(v)) : v;
        }
        uint64_t start() const { return number(num[0], num_len[0]); }
        uint64_t size() const { return number(num[1], num_len[1]); }
        uint64_t src_size() const { return number(num[2], num_len[2]); }
        char strand = '.';
        const char* text = nullptr;
        size_t text_len = 0;
    };

    // to the next line that starts a block ('a'), from the start of a line: one streaming scan for "\na" (line by line with memchr it
    // was ~40 ns per line of a walk that only wants the block's end; glibc's memmem with that needle runs at a fifth of memchr's speed)
    void skip_to_next_block()
    {
        if (pos_ >= size_ || data_[pos_] == 'a') return;
        const char* const end = data_ + size_;
        const char* q = scan::newline_a(data_ + pos_, end);
        pos_ = q < end ? static_cast<size_t>(q - data_) + 1 : size_;
    }

    void skip_line()
    {
        const void* nl = memchr(data_ + pos_, '\n', size_ - pos_);
        pos_ = nl ? static_cast<size_t>(static_cast<const char*>(nl) - data_) + 1 : size_;
    }

    // "s <src> <start> <size> <strand> <srcSize> <text>", fields separated by runs of spaces.  The six short fields are walked byte by
    // byte (~40 bytes), the sequence -- almost the whole line -- in ONE vector scan for its end (a newline, or a space: a stray field
    // behind the sequence is ignored as before).
    SLine parse_s_line()
    {
        const char* const e = data_ + size_;   // (a line never runs past its newline: every walk below stops there)
        const char* p = data_ + pos_;
        const char* tok[7] = {nullptr};
        size_t len[7] = {0};
        int k = 0;
        while (k < 6) {
            while (p < e && *p == ' ') ++p;
            if (p >= e || *p == '\n') break;
            const char* b = p;
            while (p < e && *p != ' ' && *p != '\n') ++p;
            tok[k] = b;
            len[k] = static_cast<size_t>(p - b);
            ++k;
        }
        const char* nl = nullptr;
        if (k == 6) {
            while (p < e && *p == ' ') ++p;
            if (p < e && *p != '\n') {
                const char* b = p;
                p = scan::newline_or_space(p, e);
                tok[6] = b;
                len[6] = static_cast<size_t>(p - b);
                k = 7;
                nl = (p < e && *p == '\n') ? p : static_cast<const char*>(memchr(p, '\n', static_cast<size_t>(e - p)));
            }
        }
        if (k < 7) throw std::runtime_error("malformed 's' line in MAF at byte " + std::to_string(pos_));
        SLine s;
        const char* dot = static_cast<const char*>(memchr(tok[1], '.', len[1]));
        if (!dot) throw std::runtime_error("MAF source '" + std::string(tok[1], len[1]) + "' is not of the form species.chrom");
        s.species = tok[1];
        s.species_len = static_cast<size_t>(dot - tok[1]);
        s.chrom = dot + 1;
        s.chrom_len = len[1] - static_cast<size_t>(dot + 1 - tok[1]);
        s.num[0] = tok[2]; s.num_len[0] = len[2];
        s.num[1] = tok[3]; s.num_len[1] = len[3];
        s.strand = tok[4][0];
        s.num[2] = tok[5]; s.num_len[2] = len[5];
        s.text = tok[6];
        s.text_len = len[6];
        if (s.text_len && s.text[s.text_len - 1] == '\r') --s.text_len;
        pos_ = nl ? static_cast<size_t>(nl - data_) + 1 : size_;
        return s;
    }

    // leaf of the line's species, -1 if the model does not have it.  A small direct-mapped table keyed by the name as
    // written in the file answers the repeats (a MAF names the same few species in every block) with one memcmp; the
    // lower-casing and the hash look-ups happen only for a name not seen recently.
    int leaf_of(const SLine& s)
    {
        // key of the slot: the name's first and last eight bytes and its length, mixed (a byte-wise hash cost as much as the rest of the
        // look-up; the slot's name is compared in full below, so a collision only costs a miss)
        uint64_t h = 0;
        if (s.species_len >= 8) {
            uint64_t tail;
            memcpy(&h, s.species, 8);
            memcpy(&tail, s.species + s.species_len - 8, 8);
            h ^= tail * 0x9e3779b97f4a7c15ull;
        } else {
            memcpy(&h, s.species, s.species_len);
        }
        h = (h ^ (static_cast<uint64_t>(s.species_len) << 56)) * 0xff51afd7ed558ccdull;
        if (recent_.empty()) recent_.resize(RECENT_SLOTS);
        Recent& r = recent_[(h ^ (h >> 29)) & (RECENT_SLOTS - 1)];
        if (r.used && r.name.size() == s.species_len && memcmp(r.name.data(), s.species, s.species_len) == 0) return r.leaf;
        key_.assign(s.species, s.species_len);  // no allocation once the buffer has grown
        for (char& c : key_) c = static_cast<char>(std::tolower(static_cast<unsigned char>(c)));
        const auto it = names_.find(key_);
        const int leaf = it == names_.end() ? -1 : it->second;
        if (leaf < 0 && unknown_.insert(key_).second && verbose_)
            fprintf(stderr, "WARNING: species %s in the alignment file is not part of the model\n", key_.c_str());
        r.name.assign(s.species, s.species_len);
        r.leaf = leaf;
        r.used = true;
        return leaf;
    }

    void check_strand(const SLine& s) const
    {
        if (concatenate_ && s.strand != '+')
            throw std::runtime_error("reference sequence is not on the + strand (" + s.species_str() + "." + s.chrom_str() + " at " + std::to_string(s.start()) + ")");
    }

    const std::unordered_map<std::string, int>& names_;
    int n_leaves_;
    bool concatenate_, verbose_;
    bool scan_only_ = false;  // next() walks the blocks of an alignment without collecting its rows (init(), measure())
    AlignmentSpan* measuring_ = nullptr;   // measure(): what the walk records
    Alignment scratch_;
    std::vector<uint32_t> filled_;         // fill(): columns written so far, per leaf
    std::unique_ptr<MafFile> own_;
    const char* data_ = nullptr;
    size_t size_ = 0, pos_ = 0, end_ = 0;
    std::unordered_set<std::string> unknown_;
    std::string key_;  // lower-cased species name of the line being looked up
    struct Recent {
        std::string name;  // as written in the file
        int leaf = -1;
        bool used = false;
    };
    static constexpr size_t RECENT_SLOTS = 1024;
    std::vector<Recent> recent_;
    std::vector<std::pair<size_t, size_t>> runs_;  // (first column, length) of the runs of non-gap reference columns
};

}  // namespace pcsf_host
