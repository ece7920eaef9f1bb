// packed_reader.hpp -- binary form of what the MAF reader yields in build-tracks mode (".pcsfbin").
//
// build-tracks spends its host time turning ~90 bytes of MAF text per reference base into the leaf-major
// nucleotide rows the scoring core takes (58 bytes per base for 58mammals).  At 8 GPUs that text parse (~25 GB/s
// on the 16 host threads of a box) is the ceiling SURVEY.md 8d.4 names.  `phylocsf-amd pack-maf` runs the reader
// ONCE -- concatenation, 'N' padding, reference-gap removal, the 1 Mb break points, i.e. every column-indexing
// rule of src/parallel_file_reader.hpp:430-685 -- and stores its alignments; build-tracks on the packed file then
// only copies rows.  The alignments, and therefore every output byte, are those of the text path
// (tests/test_host_layer.py).
//
// Layout (little-endian):
//   header   "PCSFBIN1" | u32 n_leaves | u32 row-map hash: the ordered leaf labels and the species-name -> row map of the model the file
//            was packed with, after --species / --mapping (a file packed for another tree of the same size is refused, not mis-scored)
//   record   u32 'PCSR' | u32 chrom_bytes | u64 start_pos (1-based) | u64 chrom_len | u32 len | u32 n_leaves
//            | chrom, padded with zeros to a multiple of 8 | n_leaves rows of len bytes, padded to a multiple of 8
//   index    u64 record offsets [n_records]
//   trailer  u64 n_records | u64 index offset | "PCSFIDX1"
// A byte-range job owns the records that START inside its range (binary search in the index): any split of the
// file yields every record exactly once, in order.
#pragma once

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "maf_reader.hpp"

namespace pcsf_host {

constexpr char PACKED_MAGIC[9] = "PCSFBIN1";
constexpr char PACKED_INDEX_MAGIC[9] = "PCSFIDX1";
constexpr uint32_t PACKED_RECORD_MAGIC = 0x52534350u;  // "PCSR"

inline bool is_packed_file(const MafFile& f) { return f.size() >= 16 + 24 && memcmp(f.data(), PACKED_MAGIC, 8) == 0; }

// what decides which row of an alignment a MAF line lands in: leaf labels in row order, then every accepted species name with its row
// (FNV-1a, 32 bits, never 0)
template <class ModelT>
inline uint32_t row_map_hash(const ModelT& model)
{
    uint32_t h = 2166136261u;
    auto eat = [&](const void* p, size_t n) {
        const unsigned char* b = static_cast<const unsigned char*>(p);
        for (size_t i = 0; i < n; ++i) h = (h ^ b[i]) * 16777619u;
    };
    for (int i = 0; i < model.n_leaves; ++i) eat(model.label[i].c_str(), model.label[i].size() + 1);
    std::vector<std::pair<std::string, int>> names(model.name_to_leaf.begin(), model.name_to_leaf.end());
    std::sort(names.begin(), names.end());
    for (const auto& kv : names) {
        eat(kv.first.c_str(), kv.first.size() + 1);
        const int32_t row = kv.second;
        eat(&row, sizeof row);
    }
    return h ? h : 1u;
}

class PackedWriter {
public:
    PackedWriter(const std::string& path, int n_leaves, uint32_t row_hash) : n_leaves_(static_cast<uint32_t>(n_leaves))
    {
        f_ = fopen(path.c_str(), "wb");
        if (!f_) throw std::runtime_error("cannot create file '" + path + "'");
        put(PACKED_MAGIC, 8);
        const uint32_t hdr[2] = {n_leaves_, row_hash};
        put(hdr, sizeof hdr);
    }
    PackedWriter(const PackedWriter&) = delete;
    PackedWriter& operator=(const PackedWriter&) = delete;
    ~PackedWriter()
    {
        if (f_) fclose(f_);
    }
    void add(const Alignment& aln)
    {
        if (aln.seqs.size() != n_leaves_) throw std::runtime_error("PackedWriter: alignment with a different number of rows");
        offsets_.push_back(at_);
        const uint32_t len = static_cast<uint32_t>(aln.length());
        const uint32_t head[2] = {PACKED_RECORD_MAGIC, static_cast<uint32_t>(aln.chrom.size())};
        put(head, sizeof head);
        const uint64_t pos[2] = {aln.start_pos, aln.chrom_len};
        put(pos, sizeof pos);
        const uint32_t dims[2] = {len, n_leaves_};
        put(dims, sizeof dims);
        put(aln.chrom.data(), aln.chrom.size());
        pad();
        for (const std::string& row : aln.seqs) {
            if (row.size() != len) throw std::runtime_error("PackedWriter: ragged alignment");
            put(row.data(), len);
        }
        pad();
    }
    void finish()
    {
        const uint64_t index_at = at_;
        if (!offsets_.empty()) put(offsets_.data(), offsets_.size() * sizeof(uint64_t));
        const uint64_t tail[2] = {offsets_.size(), index_at};
        put(tail, sizeof tail);
        put(PACKED_INDEX_MAGIC, 8);
        const int rc = fclose(f_);
        f_ = nullptr;
        if (rc != 0) throw std::runtime_error("write failed");
    }
    size_t records() const { return offsets_.size(); }
    uint64_t bytes() const { return at_; }

private:
    void put(const void* p, size_t n)
    {
        if (n && fwrite(p, 1, n, f_) != n) throw std::runtime_error("write failed");
        at_ += n;
    }
    void pad()
    {
        static const char zeros[8] = {0};
        if (at_ % 8) put(zeros, 8 - at_ % 8);
    }
    FILE* f_ = nullptr;
    uint32_t n_leaves_;
    uint64_t at_ = 0;
    std::vector<uint64_t> offsets_;
};

// the records that start in [range_begin, range_end) of a mapped packed file, in order
class PackedReader {
public:
    PackedReader(const MafFile& file, size_t range_begin, size_t range_end, int n_leaves, uint32_t row_hash)
        : data_(file.data()), size_(file.size()), n_leaves_(static_cast<uint32_t>(n_leaves))
    {
        if (!is_packed_file(file)) throw std::runtime_error("not a packed alignment file");
        uint32_t nl, hash;
        memcpy(&nl, data_ + 8, 4);
        memcpy(&hash, data_ + 12, 4);
        if (static_cast<int>(nl) != n_leaves)
            throw std::runtime_error("packed alignment file was written for a model with " + std::to_string(nl) + " species, this model has " + std::to_string(n_leaves));
        if (hash != row_hash)
            throw std::runtime_error("packed alignment file was written with another model, --species selection or --mapping (its species rows differ from this run's): pack the alignment again");
        if (memcmp(data_ + size_ - 8, PACKED_INDEX_MAGIC, 8) != 0) throw std::runtime_error("packed alignment file is truncated (no index)");
        uint64_t tail[2];
        memcpy(tail, data_ + size_ - 24, 16);
        n_records_ = tail[0];
        if (tail[1] > size_ - 24 || n_records_ > (size_ - 24 - tail[1]) / 8) throw std::runtime_error("packed alignment file: bad index");
        if (tail[1] % 8 || tail[1] < 16) throw std::runtime_error("packed alignment file: bad index");
        index_ = reinterpret_cast<const uint64_t*>(data_ + tail[1]);  // offsets are multiples of 8: aligned
        data_end_ = tail[1];
        // first record at or after range_begin ... first record at or after range_end
        next_ = lower_bound(range_begin);
        stop_ = lower_bound(range_end);
    }
    static uint64_t data_bytes(const MafFile& file)  // where the records end: what byte-range jobs are cut from
    {
        if (memcmp(file.data() + file.size() - 8, PACKED_INDEX_MAGIC, 8) != 0) throw std::runtime_error("packed alignment file is truncated (no index)");
        uint64_t tail[2];
        memcpy(tail, file.data() + file.size() - 24, 16);
        if (tail[1] > file.size() - 24 || tail[0] > (file.size() - 24 - tail[1]) / 8 || tail[1] % 8 || tail[1] < 16)
            throw std::runtime_error("packed alignment file: bad index");
        // the index once, before the byte-range jobs search it: offsets ascending, aligned and inside the data section (a record whose
        // offset lies outside would belong to no job and vanish without a word)
        const uint64_t* idx = reinterpret_cast<const uint64_t*>(file.data() + tail[1]);
        uint64_t prev = 0;
        for (uint64_t i = 0; i < tail[0]; ++i) {
            const uint64_t at = idx[i];
            if (at < 16 || at % 8 || at > tail[1] || tail[1] - at < 32 || (i > 0 && at <= prev))
                throw std::runtime_error("packed alignment file: bad record offset in the index");
            prev = at;
        }
        return tail[1];
    }
    // A record where it lies in the mapped file: nothing is copied.  rows + l * len = species row l (len characters).
    struct Record {
        const char* chrom = nullptr;
        uint32_t chrom_bytes = 0;
        uint64_t start_pos = 0, chrom_len = 0;
        uint32_t len = 0, n_rows = 0;
        const char* rows = nullptr;
    };
    bool next_record(Record& r)
    {
        if (next_ >= stop_) return false;
        // nothing of the file is trusted: the record must lie inside the data section before any of it is read
        const uint64_t at = index_[next_++];
        if (at < 16 || at % 8 || at > data_end_ || data_end_ - at < 32) throw std::runtime_error("packed alignment file: bad record offset in the index");
        const char* p = data_ + at;
        uint32_t head[2], dims[2];
        uint64_t pos[2];
        memcpy(head, p, 8);
        memcpy(pos, p + 8, 16);
        memcpy(dims, p + 24, 8);
        if (head[0] != PACKED_RECORD_MAGIC) throw std::runtime_error("packed alignment file: bad record");
        const uint64_t room = data_end_ - at - 32, chrom_room = (static_cast<uint64_t>(head[1]) + 7) & ~uint64_t(7);
        if (head[1] > room || chrom_room > room) throw std::runtime_error("packed alignment file: bad record (chromosome name exceeds the file)");
        if (dims[1] != n_leaves_) throw std::runtime_error("packed alignment file: bad record (row count differs from the header's)");
        p += 32;
        r.chrom = p;
        r.chrom_bytes = head[1];
        p += chrom_room;
        r.start_pos = pos[0];
        r.chrom_len = pos[1];
        r.len = dims[0];
        r.n_rows = dims[1];
        if (static_cast<uint64_t>(r.len) * r.n_rows > room - chrom_room) throw std::runtime_error("packed alignment file: record exceeds the file");
        r.rows = p;
        return true;
    }
    bool next(Alignment& aln)
    {
        Record r;
        if (!next_record(r)) return false;
        aln.chrom.assign(r.chrom, r.chrom_bytes);
        aln.start_pos = r.start_pos;
        aln.chrom_len = r.chrom_len;
        aln.strand = '+';
        aln.seqs.resize(r.n_rows);
        for (uint32_t l = 0; l < r.n_rows; ++l) aln.seqs[l].assign(r.rows + static_cast<size_t>(l) * r.len, r.len);
        return true;
    }

private:
    size_t lower_bound(uint64_t byte) const
    {
        size_t lo = 0, hi = n_records_;
        while (lo < hi) {
            const size_t mid = (lo + hi) / 2;
            if (index_[mid] < byte) lo = mid + 1;
            else hi = mid;
        }
        return lo;
    }
    const char* data_;
    size_t size_;
    uint32_t n_leaves_;
    const uint64_t* index_ = nullptr;
    uint64_t n_records_ = 0, data_end_ = 0;
    size_t next_ = 0, stop_ = 0;
};

}  // namespace pcsf_host
