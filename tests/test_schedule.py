"""Host logic without a GPU: the fused pruning program built by the C-ABI library is interpreted
here in numpy with the ORACLE's branch matrices and must reproduce the oracle's per-site z."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_binding as ob
import phylocsfpp_amd as pa
from phylocsfpp_amd import synth

F_CHERRY, F_MM_LEAF, F_PUSH0, F_POP0, F_ROOT = 0, 1, 2, 5, 8


def interpret(prog, slots, P, prior, cod_col):
    def leafmsg(l):
        c = cod_col[l]
        return P[l][:, c] if c < 64 else np.array([P[l][a, :].sum() for a in range(64)])
    cur, stk, slot, z = None, {}, 0, None
    for o in prog:
        code, a, b = int(o >> 28), int((o >> 14) & 0x3fff), int(o & 0x3fff)
        if code == F_CHERRY:
            cur = leafmsg(a) * leafmsg(b)
            continue
        if code == F_ROOT:
            z = float(prior @ cur)
            continue
        acc = P[slots[slot]] @ cur
        slot += 1
        if code == F_MM_LEAF:
            cur = acc * leafmsg(a)
        elif F_PUSH0 <= code < F_POP0:
            stk[code - F_PUSH0] = acc
            cur = leafmsg(a) * leafmsg(b)
        else:
            cur = stk[code - F_POP0] * acc
    assert slot == len(slots)
    return z


@pytest.mark.parametrize("model", ["53birds", "58mammals", "100vertebrates"])
def test_program_reproduces_oracle(model, oracle_models):
    m = oracle_models(model)
    prog, slots, depth = pa.build_schedule(m.child1, m.child2)
    assert depth <= 3
    assert len(slots) == m.nl - 2                      # one product per internal non-root node
    assert sorted(slots.tolist()) == list(range(m.nl, m.n - 1))
    assert int(prog[-1] >> 28) == F_ROOT and int(prog[0] >> 28) == F_CHERRY
    qd = m.qdiag(True)
    rc, P = m.all_P(qd, 1.0)
    assert rc == 0
    prior = np.array(qd.prior)
    cod = synth.synthetic_codon_columns(m.child1, m.child2, m.branch_len, m.S_c, m.pi_c, m.S_nc, m.pi_nc, 12, seed=5)
    lpr, anc = C.c_double(), C.c_double()
    per = np.empty(cod.shape[1])
    ob.lib().po_lpr_leaves(C.byref(qd), C.byref(m.m.tree), 1.0, cod.ctypes.data_as(C.POINTER(C.c_uint8)), cod.shape[1],
                           cod.shape[1], 0, C.byref(lpr), C.byref(anc), per.ctypes.data_as(C.POINTER(C.c_double)))
    for s in range(cod.shape[1]):
        z = interpret(prog, slots, P, prior, cod[:, s])
        assert abs(np.log(z) - per[s]) < 1e-11


def _random_tree(nl, rng):
    """random binary tree in newick_flatten numbering (leaves first, internal nodes in post-order)"""
    def build(leaves):
        if len(leaves) == 1:
            return leaves[0]
        k = rng.integers(1, len(leaves))
        return (build(leaves[:k]), build(leaves[k:]))
    t = build(list(range(nl)))
    c1, c2 = [-1] * (2 * nl - 1), [-1] * (2 * nl - 1)
    nxt = [nl]

    def number(node):
        if not isinstance(node, tuple):
            return node
        a, b = number(node[0]), number(node[1])
        i = nxt[0]
        nxt[0] += 1
        c1[i], c2[i] = a, b
        return i
    number(t)
    return np.array(c1, dtype=np.int16), np.array(c2, dtype=np.int16)


def test_random_trees_schedule_is_a_valid_walk():
    rng = np.random.default_rng(7)
    done = 0
    for _ in range(200):
        nl = int(rng.integers(2, 40))
        c1, c2 = _random_tree(nl, rng)
        prog, slots, depth = pa.build_schedule(c1, c2)
        done += 1
        deep = depth > 3                               # parked tiles beyond three registers: the op codes carry slot 0 and
        # the kernels keep a stack pointer (pushes and pops nest)
        # symbolic interpretation: every leaf is consumed exactly once, every internal edge once
        leaves, slot, stk, cur, deep_stk = [], 0, {}, None, []
        for o in prog:
            code, a, b = int(o >> 28), int((o >> 14) & 0x3fff), int(o & 0x3fff)
            if code == F_CHERRY:
                leaves += [a, b]
                cur = {a, b}
            elif code == F_ROOT:
                assert cur == set(range(nl))
            else:
                node = slots[slot]
                slot += 1
                assert c1[node] >= 0
                if code == F_MM_LEAF:
                    leaves.append(a)
                    cur = cur | {a}
                elif code < F_POP0:
                    if deep:
                        assert code == F_PUSH0
                        deep_stk.append(cur)
                    else:
                        stk[code - F_PUSH0] = cur
                    leaves += [a, b]
                    cur = {a, b}
                else:
                    if deep:
                        assert code == F_POP0
                        cur = cur | deep_stk.pop()
                    else:
                        cur = cur | stk.pop(code - F_POP0)
        assert not deep_stk and len(deep_stk) <= 14
        assert sorted(leaves) == list(range(nl))
        assert slot == len(slots) == nl - 2
    assert done == 200


def test_invalid_trees_are_rejected():
    with pytest.raises(pa.PcsfError):
        pa.build_schedule(np.array([-1, -1, 0, 1], dtype=np.int16), np.array([-1, -1, 1, 2], dtype=np.int16))
    with pytest.raises(pa.PcsfError):
        pa.build_schedule(np.array([-1, -1, 0], dtype=np.int16), np.array([-1, -1, 0], dtype=np.int16))
    # a perfectly balanced 64-leaf tree needs 5 parked tiles: more than the three the kernels keep in registers
    c1, c2 = [-1] * 127, [-1] * 127
    nxt = [64]

    def bal(lo, hi):
        if hi - lo == 1:
            return lo
        mid = (lo + hi) // 2
        a, b = bal(lo, mid), bal(mid, hi)
        i = nxt[0]
        nxt[0] += 1
        c1[i], c2[i] = a, b
        return i
    bal(0, 64)
    prog, slots, depth = pa.build_schedule(np.array(c1, dtype=np.int16), np.array(c2, dtype=np.int16))
    assert depth == 5 and len(slots) == 62             # accepted: the DEEP kernel variants park tiles in scratch memory


def test_eigen_solver_pair_schedule_covers_every_pair_once():
    """The register-resident Jacobi kernel (csrc/eigen.hip) rotates 32 disjoint index pairs per step; its 63 steps must
    be a one-factorisation of K64 (every pair exactly once per sweep), each step's pairs must stay inside one super-group
    of the round (that is what keeps the rotations in registers), and locate() must invert the round's grouping."""
    import ctypes as C
    import phylocsfpp_amd as pa
    lib = pa.load_library()
    pairs = (C.c_int32 * (63 * 32 * 2))()
    located = (C.c_int32 * (15 * 16))()
    lib.pcsf_debug_eigen_schedule.restype = None
    lib.pcsf_debug_eigen_schedule(pairs, located)
    pairs = np.array(pairs).reshape(63, 32, 2)
    located = np.array(located).reshape(15, 16)
    seen = set()
    for step in range(63):
        idx = pairs[step].ravel()
        assert sorted(idx) == list(range(64)), step                      # disjoint: every index in exactly one pair
        for p, q in pairs[step]:
            key = (min(p, q), max(p, q))
            assert p != q and key not in seen
            seen.add(key)
    assert len(seen) == 64 * 63 // 2
    # rounds: 7 steps (3 inside the groups + 4 across), then 14 x 4; within a round the super-groups (8 indices) are fixed
    bounds = [0, 7] + [7 + 4 * k for k in range(1, 15)]
    for r in range(15):
        groups_of_sg = None
        for step in range(bounds[r], bounds[r + 1]):
            sgs = [frozenset(int(x) // 4 for x in pairs[step, 4 * G:4 * G + 4].ravel()) for G in range(8)]
            assert all(len(s) <= 2 for s in sgs)
            if step >= bounds[r] + (3 if r == 0 else 0):                  # cross steps name both groups of every super-group
                assert all(len(s) == 2 for s in sgs)
                groups_of_sg = groups_of_sg or sgs
                assert sgs == groups_of_sg
        for G, s in enumerate(groups_of_sg):
            assert sorted(int(located[r, g]) & 7 for g in s) == [G, G]
            assert sorted(int(located[r, g]) >> 3 for g in s) == [0, 1]


def test_program_facts_match_an_independent_walk_of_the_program():
    """pcsf_debug_program_facts (csrc/schedule.hpp program_facts): per op, the leaves below its result, the slot and stack depth it
    starts with, the op whose parked tile a pop takes back -- against a plain Python walk of the same program, on random trees."""
    import ctypes as C
    import phylocsfpp_amd as pa
    lib = pa.load_library()
    rng = np.random.default_rng(3)
    for trial in range(40):
        nl = int(rng.integers(2, 129))
        c1, c2 = _random_tree(nl, rng)
        n = 2 * nl - 1
        prog = (C.c_uint32 * (4 * n))()
        slots = (C.c_int32 * n)()
        n_ops, n_slots, depth = C.c_int32(), C.c_int32(), C.c_int32()
        a1 = np.ascontiguousarray(c1, np.int16)
        a2 = np.ascontiguousarray(c2, np.int16)
        p16 = C.POINTER(C.c_int16)
        assert lib.pcsf_build_schedule(n, a1.ctypes.data_as(p16), a2.ctypes.data_as(p16), prog, 4 * n, slots, C.byref(n_ops), C.byref(n_slots), C.byref(depth)) == 0
        leaves = (C.c_uint64 * (2 * n_ops.value))()
        meta = (C.c_uint32 * n_ops.value)()
        lib.pcsf_debug_program_facts.argtypes = [C.c_int32, p16, p16, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), C.c_int32]
        rc = lib.pcsf_debug_program_facts(n, a1.ctypes.data_as(p16), a2.ctypes.data_as(p16), leaves, meta, n_ops.value)
        if depth.value > 15:
            assert rc != 0
            continue
        assert rc == 0
        cur, parked, slot = set(), [], 0
        for i in range(n_ops.value):
            op = prog[i]
            code, a, b = op >> 28, (op >> 14) & 0x3fff, op & 0x3fff
            want_slot, want_depth, want_parking = slot, len(parked), 0
            if code == 0:
                cur = {a, b}
            elif code == 1:
                slot += 1
                cur = cur | {a}
            elif 2 <= code < 5:
                slot += 1
                parked.append((cur, i))
                cur = {a, b}
            elif 5 <= code < 8:
                slot += 1
                s, want_parking = parked.pop()
                cur = cur | s
            got = {l for l in range(128) if (leaves[2 * i + (l >> 6)] >> (l & 63)) & 1}
            assert got == cur, (trial, i)
            assert meta[i] == want_slot | (want_depth << 14) | (want_parking << 18), (trial, i)
        assert cur == set(range(nl)) and not parked


def test_eigen_solver_quad_form_seats_follow_the_same_schedule():
    """The seven-wave form of the Jacobi kernel (PCSF_EIGEN_FORM=2) keeps the indices on physical seats: B is re-seated through a
    look-up table, V by register moves and a four-column exchange between its two waves.  Replayed on indices over two sweeps, both
    must agree with the seat -> index map after every re-seating, and the pairs the seats give must be, step by step, the SETS of pairs
    of the register-resident form's schedule (same sweep, same order of steps)."""
    import ctypes as C
    import phylocsfpp_amd as pa
    lib = pa.load_library()
    quads = (C.c_int32 * (63 * 32 * 2))()
    lib.pcsf_debug_eigen_schedule_quads.restype = C.c_int
    assert lib.pcsf_debug_eigen_schedule_quads(quads) == 0
    pairs = (C.c_int32 * (63 * 32 * 2))()
    located = (C.c_int32 * (15 * 16))()
    lib.pcsf_debug_eigen_schedule.restype = None
    lib.pcsf_debug_eigen_schedule(pairs, located)
    quads = np.array(quads).reshape(63, 32, 2)
    pairs = np.array(pairs).reshape(63, 32, 2)
    for step in range(63):
        a = {(min(p, q), max(p, q)) for p, q in quads[step]}
        b = {(min(p, q), max(p, q)) for p, q in pairs[step]}
        assert a == b and len(a) == 32, step


# ---- cherry tables (csrc/schedule.hpp reduce_cherries; kernels.hip PAIRS): the program of the tree whose cherries are leaves ----
PAIR_BASE, TRIPLE_BASE = 8192, 12288


def interpret_with_pair_tables(prog, slots, pairs, P, prior, cod_col, triples=()):
    """the reduced program in numpy: an operand >= 8192 is a cherry, whose message is P_c (leafmsg(A) o leafmsg(B)) -- what the context
    keeps per codon pair --, an operand >= 12288 a node over a cherry and a leaf: P_v (cherry message o leafmsg(C))"""
    def leafmsg(l):
        c = cod_col[l]
        return P[l][:, c] if c < 64 else np.array([P[l][a, :].sum() for a in range(64)])

    def operand(x):
        if x < PAIR_BASE:
            return leafmsg(x)
        if x >= TRIPLE_BASE:
            pr, lc, node = triples[x - TRIPLE_BASE]
            return P[node] @ (operand(PAIR_BASE + pr) * leafmsg(lc))
        la, lb, node = pairs[x - PAIR_BASE]
        return P[node] @ (leafmsg(la) * leafmsg(lb))
    cur, stk, slot, z, products = None, {}, 0, None, 0
    for o in prog:
        code, a, b = int(o >> 28), int((o >> 14) & 0x3fff), int(o & 0x3fff)
        if code == F_CHERRY:
            cur = operand(a) * operand(b)
            continue
        if code == F_ROOT:
            z = float(prior @ cur)
            continue
        acc = P[slots[slot]] @ cur
        slot += 1
        products += 1
        if code == F_MM_LEAF:
            cur = acc * operand(a)
        elif F_PUSH0 <= code < F_POP0:
            stk[code - F_PUSH0] = acc
            cur = operand(a) * operand(b)
        else:
            cur = stk[code - F_POP0] * acc
    assert slot == len(slots)
    return z, products


@pytest.mark.parametrize("with_triples", [False, True])
@pytest.mark.parametrize("model", ["53birds", "58mammals", "100vertebrates"])
def test_cherry_table_program_reproduces_oracle(model, with_triples, oracle_models):
    m = oracle_models(model)
    red = pa.build_cherry_schedule(m.child1, m.child2, triples=with_triples)
    prog, slots, depth, pairs = red[:4]
    triples = red[4] if with_triples else np.zeros((0, 3), np.int32)
    full_prog, full_slots, full_depth = pa.build_schedule(m.child1, m.child2)
    cherries = [v for v in range(m.nl, m.n - 1) if m.child1[v] < m.nl and m.child2[v] < m.nl]
    assert sorted(int(p[2]) for p in pairs) == cherries and all(m.child1[p[2]] == p[0] and m.child2[p[2]] == p[1] for p in pairs)
    if with_triples:
        assert len(triples) == {"53birds": 7, "58mammals": 9, "100vertebrates": 15}[model]
        for pr, lc, node in triples:
            kids = {int(m.child1[node]), int(m.child2[node])}
            assert kids == {int(pairs[pr][2]), int(lc)} and lc < m.nl
    absorbed = cherries + [int(t[2]) for t in triples]
    assert len(slots) == len(full_slots) - len(absorbed) and depth <= full_depth      # one product less per table node
    assert sorted(slots.tolist() + absorbed) == sorted(full_slots.tolist())
    qd = m.qdiag(True)
    rc, P = m.all_P(qd, 1.0)
    assert rc == 0
    prior = np.array(qd.prior)
    cod = synth.synthetic_codon_columns(m.child1, m.child2, m.branch_len, m.S_c, m.pi_c, m.S_nc, m.pi_nc, 12, seed=5)
    cod[::3, 2] = 64
    cod[:, 3] = 64
    lpr, anc = C.c_double(), C.c_double()
    per = np.empty(cod.shape[1])
    ob.lib().po_lpr_leaves(C.byref(qd), C.byref(m.m.tree), 1.0, cod.ctypes.data_as(C.POINTER(C.c_uint8)), cod.shape[1],
                           cod.shape[1], 0, C.byref(lpr), C.byref(anc), per.ctypes.data_as(C.POINTER(C.c_double)))
    for s in range(cod.shape[1]):
        z, products = interpret_with_pair_tables(prog, slots, pairs, P, prior, cod[:, s], triples)
        assert products == m.nl - 2 - len(pairs) - len(triples)
        assert abs(np.log(z) - per[s]) < 1e-11


def test_cherry_table_programs_of_random_trees_use_every_leaf_once():
    rng = np.random.default_rng(11)
    with_cherries = 0
    for _ in range(300):
        nl = int(rng.integers(2, 40))
        c1, c2 = _random_tree(nl, rng)
        use_triples = bool(rng.integers(0, 2))
        red = pa.build_cherry_schedule(c1, c2, triples=use_triples)
        n = 2 * nl - 1
        cherries = [v for v in range(nl, n - 1) if c1[v] < nl and c2[v] < nl]
        if not cherries:
            assert red is None
            continue
        with_cherries += 1
        prog, slots, depth, pairs = red[:4]
        triples = red[4] if use_triples else []
        assert sorted(int(p[2]) for p in pairs) == cherries
        want_triples = [v for v in range(nl, n - 1) if (c1[v] in cherries and c2[v] < nl) or (c2[v] in cherries and c1[v] < nl)]
        assert [int(t[2]) for t in triples] == (want_triples if use_triples else [])
        cherries = [v for v in cherries] + [int(t[2]) for t in triples]      # every node whose product is in a table
        deep = depth > 3

        def leaves_of(x):
            if x >= TRIPLE_BASE:
                pr, lc, _ = triples[x - TRIPLE_BASE]
                return {int(pairs[pr][0]), int(pairs[pr][1]), int(lc)}
            return {x} if x < PAIR_BASE else {int(pairs[x - PAIR_BASE][0]), int(pairs[x - PAIR_BASE][1])}
        used, slot, stk, cur, deep_stk = [], 0, {}, None, []
        for o in prog:
            code, a, b = int(o >> 28), int((o >> 14) & 0x3fff), int(o & 0x3fff)
            if code == F_CHERRY:
                cur = leaves_of(a) | leaves_of(b)
                used += sorted(leaves_of(a)) + sorted(leaves_of(b))
            elif code == F_ROOT:
                assert cur == set(range(nl))
            else:
                node = slots[slot]
                slot += 1
                assert node not in cherries                      # never a table node: its product is in the table

                def below(v):
                    return {v} if v < nl else below(c1[v]) | below(c2[v])
                assert below(node) == cur                         # the product of an edge is applied to the message of the subtree below it
                if code == F_MM_LEAF:
                    used += sorted(leaves_of(a))
                    cur = cur | leaves_of(a)
                elif code < F_POP0:
                    if deep:
                        deep_stk.append(cur)
                    else:
                        stk[code - F_PUSH0] = cur
                    used += sorted(leaves_of(a)) + sorted(leaves_of(b))
                    cur = leaves_of(a) | leaves_of(b)
                else:
                    cur = cur | (deep_stk.pop() if deep else stk[code - F_POP0])
        assert sorted(used) == list(range(nl)) and slot == len(slots) == nl - 2 - len(cherries)
    assert with_cherries > 250


def _interpret_tables(prog, slots, operand_node, c1, c2, nl, P, prior, cod_col):
    """a reduced program in numpy for any table depth: an operand >= 8192 stands for the message the node `operand_node[x]` of the ORIGINAL
    tree sends its parent, computed here by plain recursion (what the context keeps per code)"""
    def leafmsg(l):
        c = cod_col[l]
        return P[l][:, c] if c < 64 else np.array([P[l][a, :].sum() for a in range(64)])

    def message(v):
        return leafmsg(v) if v < nl else P[v] @ (message(c1[v]) * message(c2[v]))

    def operand(x):
        return leafmsg(x) if x < PAIR_BASE else message(operand_node[x])
    cur, stk, slot, z = None, {}, 0, None
    for o in prog:
        code, a, b = int(o >> 28), int((o >> 14) & 0x3fff), int(o & 0x3fff)
        if code == F_CHERRY:
            cur = operand(a) * operand(b)
        elif code == F_ROOT:
            z = float(prior @ cur)
        else:
            acc = P[slots[slot]] @ cur
            slot += 1
            if code == F_MM_LEAF:
                cur = acc * operand(a)
            elif F_PUSH0 <= code < F_POP0:
                stk[code - F_PUSH0] = acc
                cur = operand(a) * operand(b)
            else:
                cur = stk[code - F_POP0] * acc
    assert slot == len(slots)
    return z


@pytest.mark.parametrize("max_leaves", [2, 3, 4])
@pytest.mark.parametrize("model", ["53birds", "58mammals", "100vertebrates"])
def test_table_programs_of_every_depth_reproduce_the_oracle(model, max_leaves, oracle_models):
    m = oracle_models(model)
    prog, slots, depth, operand_node = pa.build_table_schedule(m.child1, m.child2, max_leaves)
    size = [1] * m.nl + [0] * (m.n - m.nl)
    for v in range(m.nl, m.n):
        size[v] = size[m.child1[v]] + size[m.child2[v]]
    # the operands are exactly the maximal subtrees of at most max_leaves leaves that are chains of (table, leaf) or two cherries
    assert all(2 <= size[v] <= max_leaves for v in operand_node.values())
    absorbed = sum(size[v] - 1 for v in operand_node.values())
    assert len(slots) == m.nl - 2 - absorbed
    want = {2: {"53birds": 39, "58mammals": 39, "100vertebrates": 70}, 3: {"53birds": 32, "58mammals": 30, "100vertebrates": 55},
            4: {"53birds": 28, "58mammals": 25, "100vertebrates": 47}}[max_leaves][model]
    assert len(slots) == want
    qd = m.qdiag(True)
    rc, P = m.all_P(qd, 1.0)
    assert rc == 0
    prior = np.array(qd.prior)
    cod = synth.synthetic_codon_columns(m.child1, m.child2, m.branch_len, m.S_c, m.pi_c, m.S_nc, m.pi_nc, 6, seed=9)
    cod[::4, 1] = 64
    lpr, anc = C.c_double(), C.c_double()
    per = np.empty(cod.shape[1])
    ob.lib().po_lpr_leaves(C.byref(qd), C.byref(m.m.tree), 1.0, cod.ctypes.data_as(C.POINTER(C.c_uint8)), cod.shape[1],
                           cod.shape[1], 0, C.byref(lpr), C.byref(anc), per.ctypes.data_as(C.POINTER(C.c_double)))
    for s in range(cod.shape[1]):
        z = _interpret_tables(prog, slots, operand_node, m.child1, m.child2, m.nl, P, prior, cod[:, s])
        assert abs(np.log(z) - per[s]) < 1e-11


def test_table_programs_of_random_trees_cover_every_leaf_and_edge_once():
    rng = np.random.default_rng(17)
    for _ in range(300):
        nl = int(rng.integers(3, 40))
        c1, c2 = _random_tree(nl, rng)
        max_leaves = int(rng.integers(2, 5))
        red = pa.build_table_schedule(c1, c2, max_leaves)
        if red is None:
            continue
        prog, slots, depth, operand_node = red

        def below(v):
            return {v} if v < nl else below(c1[v]) | below(c2[v])

        def inner(v):
            return set() if v < nl else {v} | inner(c1[v]) | inner(c2[v])
        covered_leaves, covered_nodes = [], set(int(v) for v in slots)
        for o in prog:
            code, a, b = int(o >> 28), int((o >> 14) & 0x3fff), int(o & 0x3fff)
            ops = [a, b] if code == F_CHERRY or F_PUSH0 <= code < F_POP0 else ([a] if code == F_MM_LEAF else [])
            for x in ops:
                if x < PAIR_BASE:
                    covered_leaves.append(x)
                else:
                    v = operand_node[x]
                    assert len(below(v)) <= max_leaves
                    covered_leaves += sorted(below(v))
                    assert not (inner(v) & covered_nodes)
                    covered_nodes |= inner(v)
        assert sorted(covered_leaves) == list(range(nl))
        assert covered_nodes == set(range(nl, 2 * nl - 2))      # every internal node but the root: by a product or inside a table
