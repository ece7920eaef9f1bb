// TEST INFRASTRUCTURE.  A command-line harness around the parts of the upstream reference that compile without GSL /
// libBigWig: it includes the reference's OWN headers where they lie (-I/root/reference/src; nothing is copied into this
// repository) and prints what they compute, so that tests/golden/make_ref_vectors.py can turn the reference's behaviour
// into committed fixtures.  The likelihood core (instance.hpp, fixed_lik.hpp, omega.hpp, run.hpp) needs GSL and is NOT part
// of this build; see DESIGN.md, "Oracle".  Built only where /root/reference exists (oracle/Makefile target `ref`), output
// oracle/_ref/ref_harness (git-ignored).  Nothing under phylocsfpp_amd/ may use it.
//
//   ref_harness model <name|prefix> [species] [mapping]         newick_parse/newick_flatten/newick_reduce, name map
//   ref_harness alignments <name|prefix> <maf> <jobs> <concat>  parallel_maf_reader::get_next_alignment
//   ref_harness frames <name|prefix> <maf>                      alignment_t::update_seqs + translate, six frames
//   ref_harness power <name|prefix> <maf> <jobs> <out.wig>      compute_bls_score<true> + the power-track loop
//   ref_harness smooth <p0,..,p7> <dir_in> <dir_out> <bed>      wig_reader + process_scores (posterior + Viterbi regions)
//   ref_harness estimate <exons> <genome_length>                estimate_hmm_params_for_genome
//   ref_harness hmm <p0,..,p7>                                  get_coding_hmm
#include <cinttypes>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <unordered_set>
#include <vector>

#include "common.hpp"
#include "models.hpp"
#include "parallel_file_reader.hpp"
#include "additional_scores.hpp"
#include "wig_file_reader.hpp"

static hmm_parameter parse_params(const char* s) {
    double v[8];
    if (sscanf(s, "%lf,%lf,%lf,%lf,%lf,%lf,%lf,%lf", v, v + 1, v + 2, v + 3, v + 4, v + 5, v + 6, v + 7) != 8) {
        fprintf(stderr, "eight comma-separated values expected\n");
        exit(2);
    }
    return hmm_parameter(v[0], v[1], v + 2, v + 5);
}

static void load(Model& m, const char* name, const char* species = "", const char* mapping = "") {
    if (mapping[0]) update_sequence_name_mapping(mapping);
    load_model(m, name, species, false, 0, "");
}

int main(int argc, char** argv) {
    if (argc < 2) return 2;
    const std::string cmd = argv[1];
    // the reference prints its warnings to stdout, so the harness writes what it dumps to a file of its own
    FILE* out = getenv("REF_HARNESS_OUT") ? fopen(getenv("REF_HARNESS_OUT"), "w") : stdout;
    if (!out) return 1;
    struct Closer { FILE* f; ~Closer() { if (f != stdout) fclose(f); else fflush(f); } } closer{out};
    if (cmd == "model" && argc >= 3) {
        Model m;
        load(m, argv[2], argc > 3 ? argv[3] : "", argc > 4 ? argv[4] : "");
        for (const newick_elem& e : m.phylo_array)
            fprintf(out, "%d\t%d\t%d\t%d\t%.9g\t%s\n", e.id, e.child1_id, e.child2_id, e.parent_id, static_cast<double>(e.branch_length), e.label.c_str());
        // the name map, sorted, so that two runs compare
        std::vector<std::pair<std::string, int>> names(m.seqid_to_phyloid.begin(), m.seqid_to_phyloid.end());
        std::sort(names.begin(), names.end());
        for (const auto& kv : names) fprintf(out, "name\t%s\t%d\n", kv.first.c_str(), kv.second);
        return 0;
    }
    if (cmd == "alignments" && argc >= 6) {
        Model m;
        load(m, argv[2]);
        const bool concat = atoi(argv[5]) != 0;
        parallel_maf_reader rd(argv[3], static_cast<unsigned>(atoi(argv[4])), &m.seqid_to_phyloid, concat);
        alignment_t aln((m.phylo_array.size() + 1) / 2);
        for (unsigned job = 0; job < rd.get_jobs(); ++job) {
            rd.skip_partial_alignment(aln, job);
            while (rd.get_next_alignment(aln, job, NULL, !concat)) {
                fprintf(out, "%s\t%" PRIu64 "\t%zu\t%c\t%" PRIu64, aln.chrom.c_str(), aln.start_pos, aln.length(), concat ? '+' : aln.strand, aln.chrom_len);
                for (const std::string& s : aln.seqs) fprintf(out, "\t%s", s.c_str());
                fprintf(out, "\n");
                for (std::string& s : aln.seqs) s = "";
            }
        }
        fprintf(stderr, "jobs %u\n", rd.get_jobs());
        return 0;
    }
    if (cmd == "frames" && argc >= 4) {
        Model m;
        load(m, argv[2]);
        parallel_maf_reader rd(argv[3], 1, &m.seqid_to_phyloid, true);
        alignment_t aln((m.phylo_array.size() + 1) / 2);
        while (rd.get_next_alignment(aln, 0)) {
            const uint64_t orig = aln.start_pos;
            for (char strand = '+'; strand <= '-'; strand += 2) {
                for (unsigned frame = 1; frame <= 3; ++frame) {
                    aln.update_seqs(orig, strand, frame);
                    uint64_t first = aln.start_pos;        // position of the first codon as the raw-track loop prints it
                    if (strand == '-') first += aln.length() % 3;
                    fprintf(out, "%s\t%" PRIu64 "\t%c\t%u\t%u\t%" PRIu64 "\t%zu", aln.chrom.c_str(), orig, strand, frame, aln.skip_bases, first, aln.peptides[0].size());
                    for (const auto& pep : aln.peptides) {
                        fprintf(out, "\t");
                        for (uint8_t c : pep) fprintf(out, "%02x", c);
                    }
                    fprintf(out, "\n");
                }
                for (std::string& s : aln.seqs) {
                    std::reverse(s.begin(), s.end());
                    for (char& c : s) c = complement(c);
                }
            }
            for (std::string& s : aln.seqs) s = "";
        }
        return 0;
    }
    if (cmd == "power" && argc >= 6) {
        Model m;
        load(m, argv[2]);
        parallel_maf_reader rd(argv[3], static_cast<unsigned>(atoi(argv[4])), &m.seqid_to_phyloid, true);
        alignment_t aln((m.phylo_array.size() + 1) / 2);
        FILE* pw = fopen(argv[5], "w");
        if (!pw) return 1;
        std::vector<double> bls;
        for (unsigned job = 0; job < rd.get_jobs(); ++job) {
            rd.skip_partial_alignment(aln, job);
            while (rd.get_next_alignment(aln, job)) {
                bls.clear();
                compute_bls_score<true>(m.phylo_tree, aln, m, bls);
                int64_t skip = static_cast<int64_t>(3 - aln.start_pos) % 3;
                if (skip < 0) skip += 3;
                if (static_cast<uint64_t>(skip) + 2 < bls.size())
                    fprintf(pw, "fixedStep chrom=%s start=%" PRId64 " step=3 span=3\n", aln.chrom.c_str(), static_cast<int64_t>(aln.start_pos) + skip);
                for (uint32_t pos = skip; pos + 2 < bls.size(); pos += 3) {
                    const float avg = (bls[pos] + bls[pos + 1] + bls[pos + 2]) / 3.0;
                    my_fprintf(pw, "%.4f", avg);
                }
                for (std::string& s : aln.seqs) s = "";
            }
        }
        fclose(pw);
        return 0;
    }
    if (cmd == "hmm" && argc >= 3) {
        const hmm h = get_coding_hmm(parse_params(argv[2]));
        for (int i = 0; i < 4; ++i) fprintf(out, "%.17g%c", h.init_probs[i], i == 3 ? '\n' : '\t');
        for (int i = 0; i < 4; ++i)
            for (int j = 0; j < 4; ++j) fprintf(out, "%.17g%c", h.trans_probs[i][j], j == 3 ? '\n' : '\t');
        return 0;
    }
    if (cmd == "estimate" && argc >= 4) {
        const hmm_parameter p = estimate_hmm_params_for_genome(argv[2], static_cast<uint32_t>(strtoull(argv[3], NULL, 10)));
        fprintf(out, "%.17g,%.17g,%.17g,%.17g,%.17g,%.17g,%.17g,%.17g\n", p.coding_prior, p.coding_length_in_codons, p.non_coding_weights[0], p.non_coding_weights[1],
               p.non_coding_weights[2], p.non_coding_lengths_in_codons[0], p.non_coding_lengths_in_codons[1], p.non_coding_lengths_in_codons[2]);
        return 0;
    }
    if (cmd == "smooth" && argc >= 6) {
        const hmm h = get_coding_hmm(parse_params(argv[2]));
        const bool bed = atoi(argv[5]) != 0;
        for (int i = 0; i < 6; ++i) {
            const char strand = i < 3 ? '+' : '-';
            const std::string tag = std::string(1, strand) + std::to_string(i % 3 + 1);
            const std::string in = std::string(argv[3]) + "/PhyloCSFRaw" + tag + ".wig";
            wig_reader rd(in.c_str());
            FILE* fw = fopen((std::string(argv[4]) + "/PhyloCSF" + tag + ".wig").c_str(), "w");
            FILE* fb = bed ? fopen((std::string(argv[4]) + "/PhyloCSF" + tag + "Regions.bed").c_str(), "w") : NULL;
            if (!fw || (bed && !fb)) return 1;
            std::vector<double> scores;
            std::vector<scored_region> region;
            std::vector<scored_bed_region> beds;
            std::string chr;
            uint64_t start = 0;
            while (rd.get_next_scores(scores, chr, start)) {
                process_scores(h, scores, start, region, beds);
                fprintf(fw, "fixedStep chrom=%s start=%" PRId64 " step=3 span=3\n", chr.c_str(), static_cast<int64_t>(start));
                for (const scored_region& r : region) my_fprintf(fw, "%.3f", r.log_odds_prob);
                if (fb)
                    for (const scored_bed_region& b : beds)
                        fprintf(fb, "%s\t%" PRIu32 "\t%" PRIu32 "\t%s:%" PRIu32 "-%" PRIu32 "\t0\t%c\t%" PRIu32 "\t%" PRIu32 "\t%" PRIu32 ",%" PRIu32 ",%" PRIu32 "\n",
                                chr.c_str(), b.region_start, b.region_end, chr.c_str(), b.region_start + 1, b.region_end, strand, b.region_start, b.region_end,
                                b.color, b.color, b.color);
                scores.clear();
                region.clear();
                beds.clear();
            }
            fclose(fw);
            if (fb) fclose(fb);
        }
        return 0;
    }
    fprintf(stderr, "unknown command\n");
    return 2;
}
