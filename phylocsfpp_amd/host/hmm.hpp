// hmm.hpp -- the smoothing pass of build-tracks (--output-phylo / --output-regions): a 4-state HMM
// (coding + three non-coding length classes) over the raw per-codon scores of one frame.
// Restates, from scratch, the observable behaviour of the reference's
//   hmm::state_posterior_probabilities / get_best_path_by_viterbi   src/create_tracks.hpp:25-159
//   get_coding_hmm                                                   src/create_tracks.hpp:163-200
//   compute_log_odds / process_scores                                src/create_tracks.hpp:226-314
//   estimate_hmm_params_for_genome (+ mixture EM, 1-D Nelder-Mead)   src/estimate_hmm_parameter.hpp:38-340
//   wig_reader::get_next_scores (run merging)                        src/wig_file_reader.hpp:91-140
// The arithmetic keeps the reference's operation order (sums over states in index order, scaling by the
// per-position maximum), so that with identical parameters the printed "%.3f" values are identical.
// This runs on the host: it is a strictly sequential scan over <= a few thousand values per run, 100 flops
// per value; the six frames (and the runs inside a frame) are independent and go to separate host threads.
#pragma once

#include <algorithm>
#include <cfloat>
#include <cinttypes>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <list>
#include <map>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

namespace pcsf_host {

struct HmmParams {  // hmm_parameter
    double coding_prior = 0.0, coding_length_in_codons = 0.0;
    double non_coding_weights[3] = {0, 0, 0}, non_coding_lengths_in_codons[3] = {0, 0, 0};
};

struct Hmm {
    double init[4] = {0, 0, 0, 0};
    double trans[4][4] = {{0}};
};

// state 0 = coding; states 1..3 = non-coding classes that can only be left towards coding
inline Hmm make_coding_hmm(const HmmParams& p)
{
    Hmm h{};   // every transition not set below is impossible: a non-coding class is only left towards coding
    const double* weight = p.non_coding_weights;
    const double* length = p.non_coding_lengths_in_codons;
    // a class's share of the non-coding positions is proportional to weight x mean length
    const double mass[3] = {weight[0] * length[0], weight[1] * length[1], weight[2] * length[2]};
    const double mass_sum = mass[0] + mass[1] + mass[2];
    // a coding stretch ends with probability 1 / coding length, into class k with its weight
    const double leave[3] = {weight[0] / p.coding_length_in_codons, weight[1] / p.coding_length_in_codons, weight[2] / p.coding_length_in_codons};
    h.init[0] = p.coding_prior;
    h.trans[0][0] = 1.0 - (leave[0] + leave[1] + leave[2]);
    for (int k = 0; k < 3; ++k) {
        const int state = k + 1;
        const double enter_coding = 1.0 / length[k];
        h.init[state] = (1 - p.coding_prior) * mass[k] / mass_sum;
        h.trans[0][state] = leave[k];
        h.trans[state][0] = enter_coding;
        h.trans[state][state] = 1.0 - enter_coding;
    }
    return h;
}

inline double hmm_emit(int state, double score) { return state == 0 ? std::pow(10.0, score / 10.0) : 1.0; }

// posterior probability of the coding state at every position (scaled forward-backward); e0[pos] = the coding state's emission
// hmm_emit(0, obs[pos]) (the other states emit 1)
inline void hmm_coding_posterior_emissions(const Hmm& h, const double* e0s, size_t n, double* post, std::vector<double>& work)
{
    if (n == 0) return;
    work.resize(n * 8);
    double* fwd = work.data();
    double* bwd = work.data() + n * 4;
    for (int s = 0; s < 4; ++s) fwd[s] = h.init[s] * (s == 0 ? e0s[0] : 1.0);
    for (size_t pos = 1; pos < n; ++pos) {
        const double e0 = e0s[pos];
        double maxf = 0.0;
        for (int s = 0; s < 4; ++s) {
            double sum = 0.0;
            for (int q = 0; q < 4; ++q) sum += fwd[(pos - 1) * 4 + q] * h.trans[q][s];
            const double v = sum * (s == 0 ? e0 : 1.0);
            fwd[pos * 4 + s] = v;
            maxf = std::max(maxf, v);
        }
        for (int s = 0; s < 4; ++s) fwd[pos * 4 + s] /= maxf;
    }
    for (int s = 0; s < 4; ++s) bwd[(n - 1) * 4 + s] = 1.0;
    for (size_t pos = n - 1; pos-- > 0;) {
        const double e0 = e0s[pos + 1];
        double maxb = 0.0;
        for (int s = 0; s < 4; ++s) {
            double sum = 0.0;
            for (int q = 0; q < 4; ++q) sum += (h.trans[s][q] * (q == 0 ? e0 : 1.0) * bwd[(pos + 1) * 4 + q]);
            bwd[pos * 4 + s] = sum;
            maxb = std::max(maxb, sum);
        }
        for (int s = 0; s < 4; ++s) bwd[pos * 4 + s] /= maxb;
    }
    for (size_t pos = 0; pos < n; ++pos) {
        double total = 0.0;
        for (int s = 0; s < 4; ++s) total += (fwd[pos * 4 + s] * bwd[pos * 4 + s]);
        post[pos] = (fwd[pos * 4] * bwd[pos * 4]) / total;
    }
}

inline void hmm_coding_posterior(const Hmm& h, const std::vector<double>& obs, std::vector<double>& post, std::vector<double>& work)
{
    const size_t n = obs.size();
    post.resize(n);
    std::vector<double> e0s(n);
    for (size_t pos = 0; pos < n; ++pos) e0s[pos] = hmm_emit(0, obs[pos]);
    hmm_coding_posterior_emissions(h, e0s.data(), n, post.data(), work);
}

inline double hmm_log_odds(double prob)
{
    const double MAX_LOG_ODDS = 15.0;
    if (prob < std::pow(10.0, -MAX_LOG_ODDS)) return -MAX_LOG_ODDS;
    if (prob > 1 - std::pow(10.0, -MAX_LOG_ODDS)) return MAX_LOG_ODDS;
    return std::log10(prob / (1 - prob));
}

// most probable state path (Viterbi with per-position rescaling; ties keep the lowest previous state); e0s as above
inline std::vector<uint32_t> hmm_viterbi_emissions(const Hmm& h, const double* e0s, size_t n)
{
    std::vector<uint32_t> path(n, 0);
    if (n == 0) return path;
    std::vector<uint8_t> back((n - 1) * 4);
    double prev[4];
    for (int s = 0; s < 4; ++s) prev[s] = h.init[s] * (s == 0 ? e0s[0] : 1.0);
    for (size_t pos = 1; pos < n; ++pos) {
        double cur[4], max_prob = 0.0;
        for (int s = 0; s < 4; ++s) {
            double best = 0.0;
            uint8_t arg = 0;
            for (int q = 0; q < 4; ++q) {
                const double v = prev[q] * h.trans[q][s];
                if (v > best) {
                    arg = static_cast<uint8_t>(q);
                    best = v;
                }
            }
            best *= (s == 0 ? e0s[pos] : 1.0);
            back[(pos - 1) * 4 + s] = arg;
            cur[s] = best;
            max_prob = std::max(max_prob, best);
        }
        for (int s = 0; s < 4; ++s) prev[s] = cur[s] / max_prob;
    }
    uint32_t best = 0;  // first maximum, starting from "no element yet"
    for (uint32_t i = 1; i < 4; ++i)
        if (prev[i] > prev[best]) best = i;
    if (!(prev[best] > 0.0)) {  // all zero: the reference's scan keeps the last index whose value is > 0, else index 0
        best = 0;
    }
    path[n - 1] = best;
    for (size_t pos = n - 1; pos-- > 0;) path[pos] = back[pos * 4 + path[pos + 1]];
    return path;
}

inline std::vector<uint32_t> hmm_viterbi(const Hmm& h, const std::vector<double>& obs)
{
    std::vector<double> e0s(obs.size());
    for (size_t pos = 0; pos < obs.size(); ++pos) e0s[pos] = hmm_emit(0, obs[pos]);
    return hmm_viterbi_emissions(h, e0s.data(), obs.size());
}

struct BedRegion {
    uint32_t start = 0, end = 0;
    double prob = 0.0;
    uint32_t color = 0;  // the reference computes a colour code and discards it: always 0
};

// coding regions of one run along the Viterbi path, with the reference's boundary rules
// (src/create_tracks.hpp:264-311): `prob` is the largest coding posterior inside the region
inline void hmm_regions(const std::vector<uint32_t>& path, const std::vector<double>& post, uint32_t block_start, std::vector<BedRegion>& out)
{
    if (path.size() < 2) return;
    uint32_t starting_position = 0, starting_count = 0;
    const size_t last = path.size() - 2;
    for (size_t i = 0; i + 1 < path.size(); ++i) {
        double prob = 0;
        if (i == 0 && path[i] == 0) {
            starting_position = block_start - 1;
            starting_count = 0;
            if (path[i + 1] != 0) out.push_back({starting_position, starting_position + 3, post[starting_count], 0});
        } else if (path[i + 1] == 0 && path[i] != 0) {
            if (i != last) {
                starting_position = block_start + 3 * static_cast<uint32_t>(i) + 2;
                starting_count = static_cast<uint32_t>(i) + 1;
            } else {
                const uint32_t end_position = block_start + 3 * static_cast<uint32_t>(i) + 5;
                out.push_back({end_position - 3, end_position, post[i + 1], 0});
            }
        } else if (path[i + 1] != 0 && path[i] == 0) {
            const uint32_t end_position = block_start + 3 * static_cast<uint32_t>(i) + 2;
            for (uint32_t c = starting_count; c <= i; ++c) prob = std::max(prob, post[c]);
            out.push_back({starting_position, end_position, prob, 0});
        } else if (i == last && path[i + 1] == 0 && path[i] == 0) {
            const uint32_t end_position = block_start + 3 * static_cast<uint32_t>(i) + 5;
            for (uint32_t c = starting_count; c <= i + 1; ++c) prob = std::max(prob, post[c]);
            out.push_back({starting_position, end_position, prob, 0});
        }
    }
}

// ---- parameter estimation from a list of coding exons ---------------------------------------------
namespace hmm_detail {
// negative weighted log-likelihood of exponential gap lengths with mean 10^x
inline double neg_loglik(const std::vector<uint32_t>& points, const double* weight, double x)
{
    double f = 0.0;
    for (size_t p = 0; p < points.size(); ++p) {
        const double tau = std::pow(10.0, x);
        f -= weight[p] * (-static_cast<double>(points[p]) / tau - std::log(tau));
    }
    return f;
}

// one-dimensional Nelder-Mead on a two-vertex simplex; returns the best vertex
inline double minimise_1d(const std::vector<uint32_t>& points, const double* weight, double guess, double scale, double rel_tol)
{
    using V = std::pair<double, double>;  // x, f(x)
    auto by_value = [](const V& a, const V& b) { return a.second < b.second; };
    std::vector<V> sx{{guess, neg_loglik(points, weight, guess)}, {guess + scale, neg_loglik(points, weight, guess + scale)}};
    std::sort(sx.begin(), sx.end(), by_value);
    const double xtol = rel_tol * scale;
    bool prev_expand_or_reduce = true;
    for (int step = 0; step <= 30; ++step) {
        if (!prev_expand_or_reduce) {
            double lo = DBL_MAX, hi = DBL_MIN;
            for (const V& v : sx) {
                lo = std::min(lo, v.first);
                hi = std::max(hi, v.first);
            }
            if (hi - lo < xtol) return sx[0].first;
        }
        const size_t n = sx.size() - 1;  // 1
        double centroid = 0.0;
        for (size_t i = 0; i < n; ++i) centroid += sx[i].first;
        centroid /= static_cast<double>(n);
        const double reflection = centroid + (centroid - sx[n].first);
        const double f_ref = neg_loglik(points, weight, reflection);
        bool expand_or_reduce = false;
        if (sx[0].second <= f_ref && f_ref < sx[n - 1].second) {
            sx[n] = {reflection, f_ref};
        } else if (f_ref < sx[0].second) {
            const double expansion = centroid + 2 * (centroid - sx[n].first);
            const double f_exp = neg_loglik(points, weight, expansion);
            if (f_exp < f_ref) {
                sx[n] = {expansion, f_exp};
                expand_or_reduce = true;
            } else {
                sx[n] = {reflection, f_ref};
            }
        } else {
            const double contraction = centroid - .5 * (centroid - sx[n].first);
            const double f_con = neg_loglik(points, weight, contraction);
            if (f_con < sx[n].second) {
                sx[n] = {contraction, f_con};
            } else {
                expand_or_reduce = true;
                for (size_t i = 1; i < n + 1; ++i) {
                    const double x = sx[0].first + 0.5 * (sx[i].first - sx[0].first);
                    sx[i] = {x, neg_loglik(points, weight, x)};
                }
            }
        }
        std::sort(sx.begin(), sx.end(), by_value);
        prev_expand_or_reduce = expand_or_reduce;
    }
    throw std::runtime_error("HMM parameter estimation: Nelder-Mead did not converge in 30 steps");
}
}  // namespace hmm_detail

// exon list: "<chrom> <strand> <frame> <start> <end>" per line (blank/tab separated).  Exons are grouped by
// chrom:strand:frame, sorted, nested/overlapping ones thinned (the longer of an overlapping pair stays), the
// gaps between survivors feed a 3-component exponential mixture (EM, 20 rounds; at most 20,000 gaps, chosen by
// std::shuffle with std::default_random_engine(0) like the reference).
inline HmmParams estimate_hmm_params(const std::string& exon_path, uint32_t genome_length)
{
    struct Range {
        uint32_t start, end;
    };
    FILE* f = fopen(exon_path.c_str(), "r");
    if (!f) throw std::runtime_error("cannot open coding-exons file '" + exon_path + "'");
    std::map<std::string, std::list<Range>> by_key;
    char line[1024];
    while (fgets(line, sizeof line, f)) {
        char* save = nullptr;
        const char* tok[5];
        int k = 0;
        for (char* t = strtok_r(line, " \t", &save); t && k < 5; t = strtok_r(nullptr, " \t", &save)) tok[k++] = t;
        if (k < 5) continue;
        by_key[std::string(tok[0]) + ":" + tok[1] + ":" + tok[2]].push_back({static_cast<uint32_t>(atoi(tok[3])), static_cast<uint32_t>(atoi(tok[4]))});
    }
    fclose(f);
    uint64_t num_exons = 0;
    size_t coding_nt = 0;
    std::vector<uint32_t> gaps;
    for (auto& kv : by_key) {
        std::list<Range>& l = kv.second;
        l.sort([](const Range& a, const Range& b) { return a.start != b.start ? a.start < b.start : a.end < b.end; });
        size_t i = 0;
        auto tail = l.begin(), head = l.begin();
        ++head;
        while (i < l.size() - 1) {
            if (head->start <= tail->end) {
                if (tail->end - tail->start >= head->end - head->start) {
                    head = l.erase(head);
                } else {
                    tail = l.erase(tail);
                    head = tail;
                    ++head;
                }
            } else {
                ++head;
                ++tail;
                ++i;
            }
        }
        for (auto it = l.begin(); it != l.end(); ++it) {  // note: steps two exons at a time, as the reference does
            const uint32_t end1 = it->end;
            ++it;
            if (it == l.end()) break;
            if (it->start > end1 + 1) gaps.push_back(it->start - end1 - 1);
        }
        num_exons += l.size();
        for (const Range& r : l) coding_nt += r.end - r.start + 1;
    }
    if (num_exons == 0 || gaps.empty()) throw std::runtime_error("coding-exons file '" + exon_path + "' holds no usable exons");
    if (gaps.size() > 20000) {
        std::default_random_engine rng(0);
        std::shuffle(gaps.begin(), gaps.end(), rng);
        gaps.resize(20000);
    }
    // mixture of three exponentials over gap lengths; parameters are log10 of the mean lengths
    double x[3] = {std::log10(3000.0), std::log10(80000.0), std::log10(100.0)};
    double prior[3] = {30.0 / 41.0, 10.0 / 41.0, 1.0 / 41.0};
    const size_t n = gaps.size();
    std::vector<double> resp[3] = {std::vector<double>(n), std::vector<double>(n), std::vector<double>(n)};
    for (int round = 0; round < 20; ++round) {
        for (size_t i = 0; i < n; ++i) {
            double lik[3];
            for (int j = 0; j < 3; ++j) {
                const double tau = std::pow(10.0, x[j]);
                lik[j] = prior[j] * std::exp(-static_cast<double>(gaps[i]) / tau - std::log(tau));
            }
            const double total = lik[0] + lik[1] + lik[2];
            for (int j = 0; j < 3; ++j) resp[j][i] = total != 0.0 ? lik[j] / total : 1.0 / 3;
        }
        for (int j = 0; j < 3; ++j) {
            double sum = 0.0;
            for (size_t i = 0; i < n; ++i) sum += resp[j][i];
            prior[j] = sum / static_cast<double>(n);
        }
        for (int j = 0; j < 3; ++j) {
            if (x[j] == 0) continue;
            x[j] = hmm_detail::minimise_1d(gaps, resp[j].data(), x[j], 0.1, 0.001);
        }
    }
    HmmParams p;
    p.coding_prior = static_cast<double>(coding_nt) / static_cast<double>(genome_length) / 6.0;  // per frame
    p.coding_length_in_codons = static_cast<double>(coding_nt) / static_cast<double>(num_exons) / 3.0;
    for (int j = 0; j < 3; ++j) {
        p.non_coding_weights[j] = prior[j];
        p.non_coding_lengths_in_codons[j] = std::pow(10.0, x[j]) / 3;
    }
    return p;
}

// ---- runs of a fixedStep wig file, merged like the reference's wig reader --------------------------
struct WigRun {
    std::string chrom;
    uint64_t start = 0;
    std::vector<double> scores;
};

// calls fn(run) for every maximal stretch of step-3 values: a "fixedStep" line that continues the previous
// stretch (same chromosome, start == previous start + 3 * values so far) does not start a new run
template <class Fn>
inline void for_each_wig_run(const std::string& path, Fn&& fn)
{
    FILE* f = fopen(path.c_str(), "r");
    if (!f) throw std::runtime_error("cannot open wig file '" + path + "'");
    WigRun run;
    char line[512];
    while (fgets(line, sizeof line, f)) {
        if (line[0] == 'f') {
            char chrom[256] = {0};
            uint64_t start = 0;
            if (sscanf(line, "fixedStep chrom=%255s start=%" SCNu64, chrom, &start) != 2) throw std::runtime_error("malformed fixedStep line in " + path);
            if (!run.scores.empty() && run.chrom == chrom && run.start + 3 * run.scores.size() == start) continue;
            if (!run.scores.empty()) fn(run);
            run.chrom = chrom;
            run.start = start;
            run.scores.clear();
        } else if (line[0] != '\n' && line[0] != 0) {
            run.scores.push_back(strtod(line, nullptr));
        }
    }
    fclose(f);
    if (!run.scores.empty()) fn(run);
}

}  // namespace pcsf_host
