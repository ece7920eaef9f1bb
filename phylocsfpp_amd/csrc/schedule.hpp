// schedule.hpp -- turns the flattened tree into the straight-line "pruning program" the kernel
// interprets, and lays the branch matrices out the way the kernel reads them.
//
// The reference walks internal nodes in id order and keeps every node's alpha vector in a
// (n-nl) x 64 workspace (src/fixed_lik.hpp:125-164, 375-379).  On the GPU the per-node partial
// likelihoods of a 16-site tile live in registers, so the walk is re-ordered depth-first with
// the deeper subtree first (Sethi-Ullman): at any time one "current" tile, one accumulator tile
// and at most `stack_depth` parked tiles are live (<= 3 for every tree the reference ships).
#pragma once

#include <algorithm>
#include <array>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace pcsf {

// Primitive steps of the walk (used while building; see fuse_program for what the kernel runs)
enum OpCode : uint32_t {
    OP_CHERRY = 0,    // cur <- leafmsg(a) * leafmsg(b)                 (both children are leaves)
    OP_MATMUL = 1,    // acc <- P[next slot] x cur                      (message to the parent)
    OP_MUL_LEAF = 2,  // cur <- acc * leafmsg(a)
    OP_PUSH = 3,      // stack[a] <- acc
    OP_MUL_POP = 4,   // cur <- acc * stack[a]
    OP_ROOT = 5       // z <- prior . cur ; ln z
};

// Fused steps the kernel interprets.  Every product is fused with what follows it so that the
// leaf-message rows it needs can be fetched while the MFMAs run; the stack slot is part of the
// op code so that parked tiles are statically indexed registers.
enum FusedOp : uint32_t {
    F_CHERRY = 0,           // cur <- leafmsg(a) * leafmsg(b)                     (program start only)
    F_MM_LEAF = 1,          // acc <- P x cur ; cur <- acc * leafmsg(a)
    F_MM_PUSH_CHERRY0 = 2,  // acc <- P x cur ; stack[d] <- acc ; cur <- leafmsg(a) * leafmsg(b)   (d = code-2)
    F_MM_POP0 = 5,          // acc <- P x cur ; cur <- stack[d] * acc                                (d = code-5)
    F_ROOT = 8              // z <- prior . cur
};

inline uint32_t encode_op(OpCode c, int a = 0, int b = 0)
{
    return (static_cast<uint32_t>(c) << 28) | (static_cast<uint32_t>(a & 0x3fff) << 14) | static_cast<uint32_t>(b & 0x3fff);
}

constexpr int MAX_STACK_DEPTH = 3;        // parked tiles the kernels keep in registers (enough for every tree the reference ships)
// Deeper trees run the DEEP kernel variants, which park tiles in scratch memory (correct for any binary tree
// newick_parse accepts, src/newick.hpp:31-229; slower).  A tree of n leaves never needs more than log2(n) parked
// tiles and n_nodes < 16384, so 14 levels cover every tree that passes validate_tree.
constexpr int MAX_STACK_DEPTH_DEEP = 14;

struct Schedule {
    int n_nodes = 0, n_leaves = 0;
    std::vector<uint32_t> program;
    std::vector<int> slot_node;  // slot -> node id whose branch matrix the slot holds (internal children)
    int stack_depth = 0;
};

struct TreeView {
    int n_nodes;
    const int16_t* child1;
    const int16_t* child2;
    int n_leaves() const { return (n_nodes + 1) / 2; }
};

// throws std::invalid_argument on anything that is not a newick_flatten-ordered binary tree
inline void validate_tree(const TreeView& t)
{
    const int n = t.n_nodes, nl = t.n_leaves();
    if (n < 3 || (n % 2) == 0 || n > 0x3fff) throw std::invalid_argument("n_nodes must be odd, >= 3 and < 16384");
    std::vector<int> seen(n, 0);
    for (int i = 0; i < n; ++i) {
        const int a = t.child1[i], b = t.child2[i];
        if (i < nl) {
            if (a != -1 || b != -1) throw std::invalid_argument("node " + std::to_string(i) + " must be a leaf (children -1)");
        } else {
            if (a < 0 || b < 0 || a >= i || b >= i || a == b)
                throw std::invalid_argument("internal node " + std::to_string(i) + " must have two distinct children with smaller ids");
            if (seen[a]++ || seen[b]++) throw std::invalid_argument("a node has two parents");
        }
    }
    for (int i = 0; i < n - 1; ++i)
        if (!seen[i]) throw std::invalid_argument("node " + std::to_string(i) + " is not attached to the tree");
}

// peephole fusion of the primitive walk into the kernel's op set.  A subtree's walk always starts
// at a cherry (the deepest-first descent ends at a node with two leaf children), so a PUSH is
// always followed by a CHERRY, and a MATMUL by exactly one of MUL_LEAF / PUSH / MUL_POP.
inline std::vector<uint32_t> fuse_program(const std::vector<uint32_t>& prim)
{
    auto code = [](uint32_t op) { return op >> 28; };
    auto arg_a = [](uint32_t op) { return static_cast<int>((op >> 14) & 0x3fff); };
    auto arg_b = [](uint32_t op) { return static_cast<int>(op & 0x3fff); };
    std::vector<uint32_t> out;
    size_t i = 0;
    while (i < prim.size()) {
        const uint32_t op = prim[i];
        if (code(op) == OP_CHERRY && i == 0) {
            out.push_back((F_CHERRY << 28) | (op & 0x0fffffff));
            i += 1;
        } else if (code(op) == OP_MATMUL) {
            if (i + 1 >= prim.size()) throw std::logic_error("dangling MATMUL");
            const uint32_t nx = prim[i + 1];
            if (code(nx) == OP_MUL_LEAF) {
                out.push_back(encode_op(static_cast<OpCode>(F_MM_LEAF), arg_a(nx)));
                i += 2;
            } else if (code(nx) == OP_MUL_POP) {
                out.push_back(encode_op(static_cast<OpCode>(F_MM_POP0 + arg_a(nx))));
                i += 2;
            } else if (code(nx) == OP_PUSH) {
                if (i + 2 >= prim.size() || code(prim[i + 2]) != OP_CHERRY) throw std::logic_error("PUSH not followed by CHERRY");
                const uint32_t ch = prim[i + 2];
                out.push_back(encode_op(static_cast<OpCode>(F_MM_PUSH_CHERRY0 + arg_a(nx)), arg_a(ch), arg_b(ch)));
                i += 3;
            } else {
                throw std::logic_error("MATMUL followed by an unexpected step");
            }
        } else if (code(op) == OP_ROOT) {
            out.push_back(static_cast<uint32_t>(F_ROOT) << 28);
            i += 1;
        } else {
            throw std::logic_error("unexpected step in the pruning walk");
        }
    }
    return out;
}

inline Schedule build_schedule(const TreeView& t)
{
    validate_tree(t);
    Schedule s;
    s.n_nodes = t.n_nodes;
    s.n_leaves = t.n_leaves();
    const int nl = s.n_leaves;
    // parked tiles needed to evaluate each subtree (leaves cost nothing: they are table look-ups)
    std::vector<int> need(t.n_nodes, 0);
    for (int i = nl; i < t.n_nodes; ++i) {
        const int a = t.child1[i], b = t.child2[i];
        const bool la = a < nl, lb = b < nl;
        if (la && lb) need[i] = 0;
        else if (la) need[i] = need[b];
        else if (lb) need[i] = need[a];
        else {
            const int hi = std::max(need[a], need[b]), lo = std::min(need[a], need[b]);
            need[i] = std::max(hi, lo + 1);
        }
    }
    s.stack_depth = need[t.n_nodes - 1];
    if (s.stack_depth > MAX_STACK_DEPTH_DEEP)
        throw std::invalid_argument("tree needs " + std::to_string(s.stack_depth) + " parked tiles; this build supports " +
                                    std::to_string(MAX_STACK_DEPTH_DEEP));
    // register-resident stack: the slot is part of the op code.  Deep trees: the kernel keeps its own stack pointer
    // (pushes and pops nest), the op codes carry slot 0.
    const bool deep = s.stack_depth > MAX_STACK_DEPTH;
    // iterative post-order emission (explicit stack: caterpillar trees are as deep as they are large)
    struct Frame { int node, depth, stage; };
    std::vector<Frame> st;
    st.push_back({t.n_nodes - 1, 0, 0});
    auto emit_matmul = [&](int child) {
        s.program.push_back(encode_op(OP_MATMUL));
        s.slot_node.push_back(child);
    };
    while (!st.empty()) {
        Frame& f = st.back();
        const int a = t.child1[f.node], b = t.child2[f.node];
        const bool la = a < nl, lb = b < nl;
        if (la && lb) {
            s.program.push_back(encode_op(OP_CHERRY, a, b));
            st.pop_back();
        } else if (la || lb) {
            const int inner = la ? b : a, leaf = la ? a : b;
            if (f.stage == 0) {
                f.stage = 1;
                st.push_back({inner, f.depth, 0});
            } else {
                emit_matmul(inner);
                s.program.push_back(encode_op(OP_MUL_LEAF, leaf));
                st.pop_back();
            }
        } else {
            const int first = (need[b] > need[a]) ? b : a, second = (first == a) ? b : a;
            if (f.stage == 0) {
                f.stage = 1;
                st.push_back({first, f.depth, 0});
            } else if (f.stage == 1) {
                emit_matmul(first);
                s.program.push_back(encode_op(OP_PUSH, deep ? 0 : f.depth));
                f.stage = 2;
                st.push_back({second, f.depth + 1, 0});
            } else {
                emit_matmul(second);
                s.program.push_back(encode_op(OP_MUL_POP, deep ? 0 : f.depth));
                st.pop_back();
            }
        }
    }
    s.program.push_back(encode_op(OP_ROOT));
    s.program = fuse_program(s.program);
    return s;
}

// Static facts about every op of the fused program, for kernels that skip the ops of all-gap subtrees (factored.hip): the leaves
// whose messages the op's result (`cur` after the op) contains, the program slot and stack depth the op starts with, and for a
// pop the op that parked the tile it takes back.  Trees of at most 128 leaves (two 64-bit leaf sets); `ok` is false otherwise.
struct ProgramFacts {
    bool ok = false;
    std::vector<uint64_t> leaves;   // [n_ops][2]: bit l of word l >> 6
    std::vector<uint32_t> meta;     // [n_ops]: slot | depth << 14 | parking op << 18
};

// Subtree tables (round 6; tracks mode, the model's own branch lengths).  A cherry -- an internal node whose two children are leaves --
// sends its parent P_c (P_A[:, a] o P_B[:, b]), a vector that depends on the column only through the codon PAIR (a, b): 65 x 65 rows
// of 64 doubles per cherry and model, made once per context by the product the kernel would run (kernels.hip combine_table_kernel:
// the same multiplies, the same MFMA sequence, so the same bits).  The cherry then IS a leaf -- of a reduced tree whose program has
// one product less per cherry -- with the pair code 65 a + b as its "codon".  One level up the same holds for a node over a cherry and
// a leaf: its message is a row of a 65^3 table (code 65 (65 a + b) + c), made from the cherry's rows and the leaf's; two products less.
// (58mammals: 56 products per model in the full program, 39 with the 17 cherries as leaves, 30 with the 9 triples too.)
// One level further (PCSF_TABLE_LEAVES=4 and room for it: 65^4 rows = 9.1 GB per node and model) a node over two cherries, or over a triple and
// a leaf.  An op-word operand below PAIR_OPERAND_BASE is a leaf of the original tree, PAIR_OPERAND_BASE + p the p-th pair, TRIPLE_OPERAND_BASE + t
// the t-th triple, QUAD_OPERAND_BASE + q the q-th quad.
constexpr int PAIR_OPERAND_BASE = 8192, TRIPLE_OPERAND_BASE = 12288, QUAD_OPERAND_BASE = 14336;
constexpr int PAIR_CODES = 65 * 65, TRIPLE_CODES = 65 * 65 * 65;
constexpr long long QUAD_CODES = 65LL * 65 * 65 * 65;

struct CherryReduction {
    // Tables are made bottom-up and only some are operands of the program: the FREE ones come first in their lists (the kernel stages codes for
    // those only); the others lie inside a larger table, whose rows are made from theirs.
    std::vector<std::array<int, 3>> pairs;     // leaf A (child1), leaf B (child2), the cherry node -- every cherry below the root
    int n_free_pairs = 0;
    std::vector<std::array<int, 3>> triples;   // pair index, the third leaf, the node over them
    int n_free_triples = 0;
    struct Quad {
        int x_triple;       // >= 0: X is that triple and Y the leaf `y`; -1: X is pair `x_pair` and Y pair `y`
        int x_pair, y, node;
        int leaves[4];      // in code order: code = ((a 65 + b) 65 + c) 65 + d
    };
    std::vector<Quad> quads;                   // all free
    std::vector<int16_t> child1, child2;       // the reduced tree, numbered like newick_flatten numbers a tree (leaves first, children before parents, root last)
    std::vector<int> operand;                  // [reduced leaf]: what the op words name
    std::vector<int> node;                     // [reduced node] -> node of the original tree (its branch matrix; a table leaf: the subtree's root)
    bool any() const { return !pairs.empty(); }
    // the leaves of the original tree behind an operand
    std::vector<int> leaves_of(int operand_id) const
    {
        if (operand_id >= QUAD_OPERAND_BASE) {
            const Quad& q = quads[static_cast<size_t>(operand_id - QUAD_OPERAND_BASE)];
            return {q.leaves[0], q.leaves[1], q.leaves[2], q.leaves[3]};
        }
        if (operand_id >= TRIPLE_OPERAND_BASE) {
            const auto& t = triples[static_cast<size_t>(operand_id - TRIPLE_OPERAND_BASE)];
            return {pairs[static_cast<size_t>(t[0])][0], pairs[static_cast<size_t>(t[0])][1], t[1]};
        }
        if (operand_id >= PAIR_OPERAND_BASE) return {pairs[static_cast<size_t>(operand_id - PAIR_OPERAND_BASE)][0], pairs[static_cast<size_t>(operand_id - PAIR_OPERAND_BASE)][1]};
        return {operand_id};
    }
};

// max_leaves: 2 = cherries only, 3 = also the nodes over a cherry and a leaf, 4 = also the nodes over two cherries or over a triple and a leaf
inline CherryReduction reduce_cherries(const TreeView& t, int max_leaves = 2)
{
    validate_tree(t);
    CherryReduction r;
    const int n = t.n_nodes, nl = t.n_leaves();
    std::vector<int> pair_of(n, -1), triple_of(n, -1), quad_of(n, -1);
    for (int v = nl; v < n - 1; ++v)     // (the root sends no message: nothing is absorbed into it)
        if (t.child1[v] < nl && t.child2[v] < nl) {
            pair_of[v] = static_cast<int>(r.pairs.size());
            r.pairs.push_back({t.child1[v], t.child2[v], v});
        }
    if (r.pairs.empty() || nl >= PAIR_OPERAND_BASE || static_cast<int>(r.pairs.size()) >= TRIPLE_OPERAND_BASE - PAIR_OPERAND_BASE) {
        r.pairs.clear();
        return r;
    }
    if (max_leaves >= 3)
        for (int v = nl; v < n - 1; ++v) {
            const int a = t.child1[v], b = t.child2[v];
            const int pr = (a >= nl && pair_of[a] >= 0 && b < nl) ? a : ((b >= nl && pair_of[b] >= 0 && a < nl) ? b : -1);
            if (pr < 0 || static_cast<int>(r.triples.size()) >= QUAD_OPERAND_BASE - TRIPLE_OPERAND_BASE) continue;
            triple_of[v] = static_cast<int>(r.triples.size());
            r.triples.push_back({pair_of[pr], pr == a ? b : a, v});
        }
    if (max_leaves >= 4)
        for (int v = nl; v < n - 1; ++v) {
            const int a = t.child1[v], b = t.child2[v];
            if (static_cast<int>(r.quads.size()) >= 16383 - QUAD_OPERAND_BASE) break;
            CherryReduction::Quad q{};
            q.node = v;
            if (a >= nl && b >= nl && pair_of[a] >= 0 && pair_of[b] >= 0) {
                q.x_triple = -1;
                q.x_pair = pair_of[a];
                q.y = pair_of[b];
            } else {
                const int tr = (a >= nl && triple_of[a] >= 0 && b < nl) ? a : ((b >= nl && triple_of[b] >= 0 && a < nl) ? b : -1);
                if (tr < 0) continue;
                q.x_triple = triple_of[tr];
                q.x_pair = -1;
                q.y = tr == a ? b : a;
            }
            quad_of[v] = static_cast<int>(r.quads.size());
            r.quads.push_back(q);
        }
    // which tables lie inside a larger one
    std::vector<char> inner_pair(r.pairs.size(), 0), inner_triple(r.triples.size(), 0);
    for (const auto& q : r.quads) {
        if (q.x_triple >= 0) inner_triple[static_cast<size_t>(q.x_triple)] = 1;
        else inner_pair[static_cast<size_t>(q.x_pair)] = inner_pair[static_cast<size_t>(q.y)] = 1;
    }
    for (const auto& tr : r.triples) inner_pair[static_cast<size_t>(tr[0])] = 1;
    // the free ones first
    auto reorder = [](std::vector<std::array<int, 3>>& list, const std::vector<char>& inner, int& n_free) {
        std::vector<int> moved(list.size(), -1);
        std::vector<std::array<int, 3>> ordered;
        n_free = 0;
        for (int pass = 0; pass < 2; ++pass)
            for (size_t k = 0; k < list.size(); ++k)
                if (inner[k] == pass) {
                    moved[k] = static_cast<int>(ordered.size());
                    ordered.push_back(list[k]);
                    if (pass == 0) ++n_free;
                }
        list.swap(ordered);
        return moved;
    };
    const std::vector<int> moved_pair = reorder(r.pairs, inner_pair, r.n_free_pairs);
    const std::vector<int> moved_triple = reorder(r.triples, inner_triple, r.n_free_triples);
    for (auto& tr : r.triples) tr[0] = moved_pair[static_cast<size_t>(tr[0])];
    for (auto& q : r.quads) {
        if (q.x_triple >= 0) q.x_triple = moved_triple[static_cast<size_t>(q.x_triple)];
        else {
            q.x_pair = moved_pair[static_cast<size_t>(q.x_pair)];
            q.y = moved_pair[static_cast<size_t>(q.y)];
        }
    }
    for (int v = nl; v < n - 1; ++v) {
        if (pair_of[v] >= 0) pair_of[v] = moved_pair[static_cast<size_t>(pair_of[v])];
        if (triple_of[v] >= 0) triple_of[v] = moved_triple[static_cast<size_t>(triple_of[v])];
    }
    for (auto& q : r.quads) {   // the leaves in code order: X's, then Y's
        if (q.x_triple >= 0) {
            const auto& tr = r.triples[static_cast<size_t>(q.x_triple)];
            q.leaves[0] = r.pairs[static_cast<size_t>(tr[0])][0];
            q.leaves[1] = r.pairs[static_cast<size_t>(tr[0])][1];
            q.leaves[2] = tr[1];
            q.leaves[3] = q.y;
        } else {
            q.leaves[0] = r.pairs[static_cast<size_t>(q.x_pair)][0];
            q.leaves[1] = r.pairs[static_cast<size_t>(q.x_pair)][1];
            q.leaves[2] = r.pairs[static_cast<size_t>(q.y)][0];
            q.leaves[3] = r.pairs[static_cast<size_t>(q.y)][1];
        }
    }
    // the operands: quads, free triples, free pairs -- each absorbs the whole subtree below its node
    std::vector<int> table_root(n, -1), operand_of(n, -1);
    auto absorb = [&](int root_node, int operand_id) {
        operand_of[root_node] = operand_id;
        std::vector<int> todo{root_node};
        while (!todo.empty()) {
            const int v = todo.back();
            todo.pop_back();
            table_root[v] = root_node;
            if (v >= nl) {
                todo.push_back(t.child1[v]);
                todo.push_back(t.child2[v]);
            }
        }
    };
    for (size_t k = 0; k < r.quads.size(); ++k) absorb(r.quads[k].node, QUAD_OPERAND_BASE + static_cast<int>(k));
    for (int k = 0; k < r.n_free_triples; ++k) absorb(r.triples[static_cast<size_t>(k)][2], TRIPLE_OPERAND_BASE + k);
    for (int k = 0; k < r.n_free_pairs; ++k) absorb(r.pairs[static_cast<size_t>(k)][2], PAIR_OPERAND_BASE + k);
    std::vector<int> new_id(n, -1);
    for (int l = 0; l < nl; ++l)
        if (table_root[l] < 0) {
            new_id[l] = static_cast<int>(r.node.size());
            r.operand.push_back(l);
            r.node.push_back(l);
        }
    for (int v = nl; v < n - 1; ++v) {
        if (table_root[v] != v) continue;
        new_id[v] = static_cast<int>(r.node.size());
        r.operand.push_back(operand_of[v]);
        r.node.push_back(v);
    }
    const size_t n_red_leaves = r.node.size();
    r.child1.assign(n_red_leaves, -1);
    r.child2.assign(n_red_leaves, -1);
    for (int v = nl; v < n; ++v) {
        if (table_root[v] >= 0) continue;    // inside a table leaf (or one itself)
        new_id[v] = static_cast<int>(r.node.size());
        r.node.push_back(v);
        r.child1.push_back(static_cast<int16_t>(new_id[t.child1[v]]));
        r.child2.push_back(static_cast<int16_t>(new_id[t.child2[v]]));
    }
    return r;
}

// the program of the reduced tree, its leaf operands renamed to what the kernel looks up, its slots to the original tree's nodes
inline Schedule build_reduced_schedule(const CherryReduction& r)
{
    Schedule s = build_schedule(TreeView{static_cast<int>(r.node.size()), r.child1.data(), r.child2.data()});
    for (uint32_t& op : s.program) {
        const uint32_t code = op >> 28;
        const int a = static_cast<int>((op >> 14) & 0x3fff), b = static_cast<int>(op & 0x3fff);
        if (code == F_CHERRY || (code >= F_MM_PUSH_CHERRY0 && code < F_MM_POP0)) op = (code << 28) | (static_cast<uint32_t>(r.operand[a]) << 14) | static_cast<uint32_t>(r.operand[b]);
        else if (code == F_MM_LEAF) op = (code << 28) | (static_cast<uint32_t>(r.operand[a]) << 14);
    }
    for (int& v : s.slot_node) v = r.node[v];
    return s;
}

// (red: the table leaves behind operands >= PAIR_OPERAND_BASE of a reduced program; n_leaves: leaves of the ORIGINAL tree, which the sets are over)
inline ProgramFacts program_facts(const Schedule& s, const CherryReduction* red = nullptr, int n_leaves = -1)
{
    ProgramFacts f;
    const size_t n_ops = s.program.size();
    if ((n_leaves < 0 ? s.n_leaves : n_leaves) > 128 || n_ops >= (1u << 14)) return f;
    f.leaves.assign(2 * n_ops, 0);
    f.meta.assign(n_ops, 0);
    struct Set { uint64_t w[2] = {0, 0}; };
    auto with = [red](Set x, int leaf) {
        if (leaf >= PAIR_OPERAND_BASE && red) {
            for (int l : red->leaves_of(leaf)) x.w[l >> 6] |= uint64_t(1) << (l & 63);
            return x;
        }
        x.w[leaf >> 6] |= uint64_t(1) << (leaf & 63);
        return x;
    };
    Set cur;
    std::vector<std::pair<Set, uint32_t>> parked;   // (leaves of the parked message, index of the parking op)
    uint32_t slot = 0;
    for (size_t i = 0; i < n_ops; ++i) {
        const uint32_t op = s.program[i], code = op >> 28;
        const int a = static_cast<int>((op >> 14) & 0x3fff), b = static_cast<int>(op & 0x3fff);
        uint32_t parking = 0;
        const uint32_t depth = static_cast<uint32_t>(parked.size());
        const uint32_t slot_here = slot;
        if (code == F_CHERRY) {
            cur = with(with(Set{}, a), b);
        } else if (code == F_MM_LEAF) {
            ++slot;
            cur = with(cur, a);
        } else if (code >= F_MM_PUSH_CHERRY0 && code < F_MM_POP0) {
            ++slot;
            parked.push_back({cur, static_cast<uint32_t>(i)});
            cur = with(with(Set{}, a), b);
        } else if (code >= F_MM_POP0 && code < F_ROOT) {
            ++slot;
            if (parked.empty()) throw std::logic_error("program_facts: pop without a parked tile");
            parking = parked.back().second;
            cur.w[0] |= parked.back().first.w[0];
            cur.w[1] |= parked.back().first.w[1];
            parked.pop_back();
        }
        if (depth > 15 || slot_here >= (1u << 14)) return ProgramFacts{};
        f.leaves[2 * i] = cur.w[0];
        f.leaves[2 * i + 1] = cur.w[1];
        f.meta[i] = slot_here | (depth << 14) | (parking << 18);
    }
    f.ok = true;
    return f;
}

// ---- device table layouts -------------------------------------------------------------------
// MFMA v_mfma_f64_16x16x4_f64 computes D[16x16] += A[16x4] * B[4x16]; lane l supplies
// A[l&15][l>>4] and B[l>>4][l&15] and receives D[(l>>4)+4r][l&15] in result register r.
// With A = a 16x4 piece of P (rows: parent states, k: child states) and B = partial
// likelihoods (k: child state, column: site) the result registers of output tile t' are
// exactly the B operands of k-steps 4t'..4t'+3 of the next product, so a tile never moves
// between lanes.  "fragment order" below is the order in which lanes consume P:
//   frag[q][h][lane][e] = P[16*(2h+e) + (lane&15)][4q + (lane>>4)],  q<16, h<2, e<2
constexpr int FRAG_DOUBLES = 64 * 64;

inline void to_fragment_order(const double* P, double* frag)
{
    for (int q = 0; q < 16; ++q)
        for (int h = 0; h < 2; ++h)
            for (int lane = 0; lane < 64; ++lane)
                for (int e = 0; e < 2; ++e)
                    frag[((q * 2 + h) * 64 + lane) * 2 + e] = P[(16 * (2 * h + e) + (lane & 15)) * 64 + 4 * q + (lane >> 4)];
}

// leaf messages: table[code][q2][g][e] = P[4(2 q2 + e) + g][code] for code < 64, and for code 64
// (marginalise) the row sum of P accumulated in index order like the reference's loop
// (src/fixed_lik.hpp:111-117).  Lane (site, g) of a tile reads the 16-byte pieces (q2, g, 0..1), q2 = 0..7,
// so the four lanes of a site read 64 contiguous bytes.
constexpr int LEAF_TABLE_DOUBLES = 65 * 64;

inline void to_leaf_table(const double* P, double* table)
{
    for (int code = 0; code < 65; ++code)
        for (int g = 0; g < 4; ++g)
            for (int q = 0; q < 16; ++q) {
                const int a = 4 * q + g;
                double v;
                if (code < 64) v = P[a * 64 + code];
                else {
                    v = 0.0;
                    for (int b = 0; b < 64; ++b) v += P[a * 64 + b];
                }
                table[((code * 8 + (q >> 1)) * 4 + g) * 2 + (q & 1)] = v;
            }
}

inline void to_lane_order(const double* vec64, double* out)  // out[g][q] = vec[4q+g]
{
    for (int g = 0; g < 4; ++g)
        for (int q = 0; q < 16; ++q) out[g * 16 + q] = vec64[4 * q + g];
}

}  // namespace pcsf
