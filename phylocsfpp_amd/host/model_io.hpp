// model_io.hpp -- host-side model input: ECM rate files, newick tree, species-name map.
// Mirrors what load_model() hands to the scoring core (reference src/models.hpp:1757-1857,
// src/ecm.hpp:20-67, src/newick.hpp:31-229), including the --species tree reduction
// (src/newick.hpp:286-363) and the --mapping name table (src/models.hpp:1709-1740); written from scratch.
#pragma once

#include <dlfcn.h>
#include <limits.h>
#include <sys/stat.h>

#include <algorithm>
#include <cctype>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <unordered_set>
#include <vector>

namespace pcsf_host {

struct Model {
    int n_nodes = 0, n_leaves = 0;
    // newick_flatten numbering: leaves 0..nl-1 in left-to-right order, internal nodes in post-order
    std::vector<int16_t> child1, child2, parent;
    std::vector<float> branch_len;     // what the likelihood uses (newick_elem::branch_length is float)
    std::vector<double> branch_len_d;  // what the branch-length score uses (newick_node, double)
    std::vector<std::string> label;    // lower-cased; empty for internal nodes
    std::vector<double> S_c, pi_c, S_nc, pi_nc;  // 64x64 row-major, 64
    std::unordered_map<std::string, int> name_to_leaf;  // labels + aliases, lower case
};

// ---- ECM file: 63 lower-triangle rows, a blank line, 64 frequencies (then codon names) --------
inline void read_ecm(const std::string& path, std::vector<double>& S, std::vector<double>& pi)
{
    std::ifstream in(path);
    if (!in) throw std::runtime_error("cannot open model file '" + path + "' (pass the prefix without file endings)");
    S.assign(64 * 64, 0.0);
    pi.assign(64, 0.0);
    std::string line;
    int row = 1;  // the first line of the file is row 1 of the triangle (row 0 would be empty)
    bool have_freq = false;
    while (std::getline(in, line)) {
        const char* p = line.c_str();
        char* end = nullptr;
        int col = 0;
        if (row <= 63) {
            for (double v = std::strtod(p, &end); p != end; v = std::strtod(p, &end)) {
                p = end;
                if (col < row) S[col * 64 + row] = S[row * 64 + col] = v;
                ++col;
            }
            if (col != row) throw std::runtime_error(path + ": triangle row " + std::to_string(row) + " has " + std::to_string(col) + " values");
        } else if (row == 65) {
            for (double v = std::strtod(p, &end); p != end; v = std::strtod(p, &end)) {
                p = end;
                if (col < 64) pi[col] = v;
                ++col;
            }
            if (col != 64) throw std::runtime_error(path + ": expected 64 codon frequencies");
            have_freq = true;
        }
        ++row;
    }
    if (!have_freq) throw std::runtime_error(path + ": codon frequency line missing");
}

// ---- newick ---------------------------------------------------------------------------------
namespace detail {
struct Node {
    int left = -1, right = -1, parent = -1;
    double len = 0.0;
    std::string label;
    int id = -1;
};

struct NewickParser {
    const std::string& s;
    size_t pos = 0;
    std::vector<Node> nodes;
    explicit NewickParser(const std::string& text) : s(text) {}

    int parse(int parent)
    {
        const int me = static_cast<int>(nodes.size());
        nodes.emplace_back();
        nodes[me].parent = parent;
        if (pos < s.size() && s[pos] == '(') {
            ++pos;
            const int l = parse(me);
            expect(',');
            const int r = parse(me);
            expect(')');  // strictly binary, like the reference
            nodes[me].left = l;
            nodes[me].right = r;
        } else {
            std::string lab;
            while (pos < s.size() && s[pos] != '(' && s[pos] != ',' && s[pos] != ')' && s[pos] != ':' && s[pos] != ';')
                lab += static_cast<char>(std::tolower(static_cast<unsigned char>(s[pos++])));
            if (lab.empty()) fail("empty leaf label");
            nodes[me].label = lab;
        }
        if (pos < s.size() && s[pos] == ':') {
            ++pos;
            std::string num;
            while (pos < s.size() && (std::isdigit(static_cast<unsigned char>(s[pos])) || s[pos] == '.')) num += s[pos++];
            if (num.empty()) fail("branch length expected");
            nodes[me].len = std::stod(num);
        }
        return me;
    }
    void expect(char c)
    {
        if (pos >= s.size() || s[pos] != c) fail(std::string("'") + c + "' expected");
        ++pos;
    }
    [[noreturn]] void fail(const std::string& what) { throw std::runtime_error("newick: " + what + " at offset " + std::to_string(pos)); }
};
}  // namespace detail

namespace detail {
inline int selected_below(const std::vector<Node>& nd, int v, const std::unordered_set<std::string>& keep)
{
    if (nd[v].left < 0) return keep.count(nd[v].label) ? 1 : 0;
    return selected_below(nd, nd[v].left, keep) + selected_below(nd, nd[v].right, keep);
}

// --species: drop every subtree without a selected leaf.  A node that loses a child is merged with the
// surviving child: it takes over that child's children (or its label, if the child is a leaf) and, unless it
// is the root, the child's branch length is ADDED to its own (double, before the float narrowing); the root
// never carries a length.  Same result as the reference's newick_reduce for any selection.
inline void reduce_to(std::vector<Node>& nd, int v, const std::unordered_set<std::string>& keep)
{
    while (nd[v].left >= 0) {
        const int l = nd[v].left, r = nd[v].right;
        const int nl = selected_below(nd, l, keep), nr = selected_below(nd, r, keep);
        if (nl > 0 && nr > 0) {
            reduce_to(nd, l, keep);
            reduce_to(nd, r, keep);
            return;
        }
        const int stay = nl == 0 ? r : l;
        nd[v].left = nd[stay].left;
        nd[v].right = nd[stay].right;
        if (nd[v].left >= 0) {
            nd[nd[v].left].parent = v;
            nd[nd[v].right].parent = v;
        } else {
            nd[v].label = nd[stay].label;
        }
        if (nd[v].parent >= 0) nd[v].len += nd[stay].len;
    }
}
}  // namespace detail

// keep: lower-cased tree labels to reduce the tree to (nullptr or empty = the whole tree)
inline void parse_newick(const std::string& text, Model& m, const std::unordered_set<std::string>* keep = nullptr)
{
    std::string s;
    for (char c : text)
        if (!std::isspace(static_cast<unsigned char>(c))) s += c;
    detail::NewickParser P(s);
    const int root = P.parse(-1);
    if (P.pos < s.size() && s[P.pos] != ';') P.fail("trailing characters");
    if (P.nodes.size() % 2 == 0) throw std::runtime_error("newick: not a binary tree");
    if (keep && !keep->empty()) {
        detail::reduce_to(P.nodes, root, *keep);
        std::vector<detail::Node> live;  // compact to the reachable nodes
        std::vector<int> remap(P.nodes.size(), -1), order{root};
        for (size_t i = 0; i < order.size(); ++i) {
            const detail::Node& x = P.nodes[order[i]];
            if (x.left >= 0) {
                order.push_back(x.left);
                order.push_back(x.right);
            }
        }
        for (size_t i = 0; i < order.size(); ++i) remap[order[i]] = static_cast<int>(i);
        for (int old : order) {
            detail::Node x = P.nodes[old];
            x.left = x.left >= 0 ? remap[x.left] : -1;
            x.right = x.right >= 0 ? remap[x.right] : -1;
            x.parent = x.parent >= 0 ? remap[x.parent] : -1;
            live.push_back(x);
        }
        live[0].parent = -1;
        live[0].len = 0.0;
        P.nodes.swap(live);
    }
    const int root_ix = (keep && !keep->empty()) ? 0 : root;
    auto& nd = P.nodes;
    int nl = 0;
    for (auto& x : nd) nl += x.left < 0;
    const int n = static_cast<int>(nd.size());
    if (n != 2 * nl - 1 || nl < 2) throw std::runtime_error("newick: not a binary tree with >= 2 leaves (after --species reduction)");
    // ids: leaves left-to-right, internal nodes in post-order starting at nl
    int next_leaf = 0, next_inner = nl;
    std::vector<std::pair<int, int>> stack{{root_ix, 0}};
    while (!stack.empty()) {
        auto& [v, stage] = stack.back();
        if (nd[v].left < 0) {
            nd[v].id = next_leaf++;
            stack.pop_back();
        } else if (stage == 0) {
            stage = 1;
            stack.push_back({nd[v].left, 0});
        } else if (stage == 1) {
            stage = 2;
            stack.push_back({nd[v].right, 0});
        } else {
            nd[v].id = next_inner++;
            stack.pop_back();
        }
    }
    m.n_nodes = n;
    m.n_leaves = nl;
    m.child1.assign(n, -1);
    m.child2.assign(n, -1);
    m.parent.assign(n, -1);
    m.branch_len.assign(n, 0.f);
    m.branch_len_d.assign(n, 0.0);
    m.label.assign(n, "");
    for (auto& x : nd) {
        const int i = x.id;
        if (x.left >= 0) {
            m.child1[i] = static_cast<int16_t>(nd[x.left].id);
            m.child2[i] = static_cast<int16_t>(nd[x.right].id);
        }
        m.parent[i] = x.parent >= 0 ? static_cast<int16_t>(nd[x.parent].id) : -1;
        m.branch_len[i] = static_cast<float>(x.len);
        m.branch_len_d[i] = x.len;
        m.label[i] = x.label;
    }
}

inline std::string slurp(const std::string& path)
{
    std::ifstream in(path, std::ios::binary);
    if (!in) throw std::runtime_error("cannot open '" + path + "'");
    std::ostringstream ss;
    ss << in.rdbuf();
    return ss.str();
}

inline std::string lower(std::string s)
{
    for (char& c : s) c = static_cast<char>(std::tolower(static_cast<unsigned char>(c)));
    return s;
}

// alias table: "<tree label>\t<alias> <alias> ...", one species per line (all matched lower-cased).  This file
// stands in for the name table compiled into the reference (src/models.hpp:1240-1706).
using AliasTable = std::unordered_map<std::string, std::vector<std::string>>;  // tree label -> other names

inline void read_alias_table(const std::string& path, AliasTable& t)
{
    std::ifstream in(path);
    if (!in) throw std::runtime_error("cannot open species alias file '" + path + "'");
    std::string line;
    while (std::getline(in, line)) {
        const size_t tab = line.find('\t');
        if (tab == std::string::npos) continue;
        auto& names = t[lower(line.substr(0, tab))];
        std::istringstream al(line.substr(tab + 1));
        std::string a;
        while (al >> a) {
            a = lower(a);
            if (std::find(names.begin(), names.end(), a) == names.end()) names.push_back(a);
        }
    }
}

// --mapping: two columns "<common name>\t<name used in the alignments>" per line (reference
// update_sequence_name_mapping).  Both columns are lower-cased here; the reference keeps the first column's
// case, which makes a capitalised common name unreachable from the (lower-cased) tree labels.
inline void read_mapping_file(const std::string& path, AliasTable& t)
{
    std::ifstream in(path);
    if (!in) throw std::runtime_error("cannot open mapping file '" + path + "'");
    std::string line;
    while (std::getline(in, line)) {
        std::istringstream ls(line);
        std::string common, other;
        if (!(ls >> common >> other)) continue;
        auto& names = t[lower(common)];
        other = lower(other);
        if (std::find(names.begin(), names.end(), other) == names.end()) names.push_back(other);
    }
}

// --species "a,b,c": each name is a tree label or one of its other names; returns tree labels
inline std::unordered_set<std::string> resolve_species(const std::string& csv, const std::vector<std::string>& tree_labels, const AliasTable& t)
{
    std::unordered_set<std::string> keep;
    std::vector<std::string> missing;
    std::istringstream ss(csv);
    std::string tok;
    while (std::getline(ss, tok, ',')) {
        tok = lower(tok);
        if (tok.empty()) continue;
        if (std::find(tree_labels.begin(), tree_labels.end(), tok) != tree_labels.end()) {
            keep.insert(tok);
            continue;
        }
        bool found = false;
        for (const auto& row : t)
            if (std::find(row.second.begin(), row.second.end(), tok) != row.second.end() &&
                std::find(tree_labels.begin(), tree_labels.end(), row.first) != tree_labels.end()) {
                keep.insert(row.first);
                found = true;
            }
        if (!found) missing.push_back(tok);
    }
    if (!missing.empty()) {
        std::string msg = "the following selected species are missing in the phylogenetic tree:";
        for (const auto& x : missing) msg += " " + x;
        throw std::runtime_error(msg);
    }
    return keep;
}

// <prefix>_coding.ECM, <prefix>_noncoding.ECM, <prefix>.nh  (reference src/models.hpp:1773-1775)
// Model names as the reference's command line takes them (src/models.hpp:1443-1455, load_model :1757-1783): a bare name such as
// "58mammals" selects a model that ships with the program, anything else is the prefix of <prefix>_coding.ECM, <prefix>_noncoding.ECM
// and <prefix>.nh.  The reference compiles its models in; here they are files in a `models` directory next to the library / the
// driver (phylocsfpp_amd/models in the source tree; $PCSF_MODEL_DIR first).  A prefix that names existing files always wins.
inline bool file_exists(const std::string& path)
{
    struct stat sb;
    return stat(path.c_str(), &sb) == 0 && S_ISREG(sb.st_mode);
}

inline std::vector<std::string> model_directories()
{
    std::vector<std::string> dirs;
    if (const char* e = getenv("PCSF_MODEL_DIR")) dirs.push_back(e);
    Dl_info info;
    if (dladdr(reinterpret_cast<const void*>(&file_exists), &info) && info.dli_fname) {   // the object this code was linked into
        std::string here = info.dli_fname;
        char resolved[PATH_MAX];
        if (realpath(here.c_str(), resolved)) here = resolved;
        const size_t slash = here.find_last_of('/');
        const std::string dir = slash == std::string::npos ? "." : here.substr(0, slash);
        for (const char* rel : {"/models", "/../models", "/../../phylocsfpp_amd/models"}) dirs.push_back(dir + rel);
    }
    return dirs;
}

inline std::string resolve_model_prefix(const std::string& name_or_prefix)
{
    if (file_exists(name_or_prefix + "_coding.ECM") || name_or_prefix.find('/') != std::string::npos) return name_or_prefix;
    std::string tried;
    for (const std::string& d : model_directories()) {
        const std::string cand = d + "/" + name_or_prefix;
        if (file_exists(cand + "_coding.ECM")) return cand;
        tried += (tried.empty() ? "" : ", ") + d;
    }
    throw std::runtime_error("model '" + name_or_prefix + "' is neither a shipped model nor the prefix of <prefix>_coding.ECM / _noncoding.ECM / .nh (looked in: " +
                             (tried.empty() ? std::string("no model directory") : tried) + ")");
}

inline Model load_model(const std::string& name_or_prefix, const std::string& alias_file_in = "", const std::string& species_csv = "",
                        const std::string& mapping_file = "")
{
    const std::string prefix = resolve_model_prefix(name_or_prefix);
    std::string alias_file = alias_file_in;
    if (alias_file.empty()) {  // convention: species_aliases.tsv next to the model files
        const size_t slash = prefix.find_last_of('/');
        const std::string cand = (slash == std::string::npos ? std::string(".") : prefix.substr(0, slash)) + "/species_aliases.tsv";
        if (file_exists(cand)) alias_file = cand;
    }
    Model m;
    read_ecm(prefix + "_coding.ECM", m.S_c, m.pi_c);
    read_ecm(prefix + "_noncoding.ECM", m.S_nc, m.pi_nc);
    AliasTable aliases;
    if (!alias_file.empty()) read_alias_table(alias_file, aliases);
    if (!mapping_file.empty()) read_mapping_file(mapping_file, aliases);
    const std::string text = slurp(prefix + ".nh");
    if (species_csv.empty()) {
        parse_newick(text, m);
    } else {
        Model full;
        parse_newick(text, full);
        const std::vector<std::string> leaves(full.label.begin(), full.label.begin() + full.n_leaves);
        const std::unordered_set<std::string> keep = resolve_species(species_csv, leaves, aliases);
        if (keep.size() < 2) throw std::runtime_error("--species must select at least two species of the tree");
        parse_newick(text, m, &keep);
    }
    for (int i = 0; i < m.n_leaves; ++i) {
        m.name_to_leaf.emplace(m.label[i], i);
        auto it = aliases.find(m.label[i]);
        if (it != aliases.end())
            for (const auto& a : it->second) m.name_to_leaf.emplace(a, i);
    }
    return m;
}

}  // namespace pcsf_host
