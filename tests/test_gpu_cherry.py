"""GPU: subtree tables (csrc/schedule.hpp reduce_cherries, kernels.hip PAIRS / combine_table_kernel).  Tracks mode runs the program of the
tree whose cherries -- and the nodes over a cherry and a leaf -- are leaves: such a node's message to its parent is a row of a 65^2 (65^3)
table per node and model, made once per context by the product the full program would run.  The claim is BIT-IDENTITY with the full
program (PCSF_CHERRY_TABLES=0) and with the cherries-only form (PCSF_TABLE_LEAVES=2), on every entry point that runs prune_fixed in
tracks mode; parity with the oracle is what every other GPU test of the FIXED path already holds the default form to."""
import os

import numpy as np
import pytest

from phylocsfpp_amd import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _contexts(name, monkeypatch, **kw):
    """the full program, cherries only, cherries + triples (the default)"""
    import phylocsfpp_amd as pa
    model = pa.load_model(name)
    monkeypatch.setenv("PCSF_CHERRY_TABLES", "0")
    full = model.context(device=0, **kw)
    monkeypatch.delenv("PCSF_CHERRY_TABLES")
    monkeypatch.setenv("PCSF_TABLE_LEAVES", "2")
    pairs = model.context(device=0, **kw)
    monkeypatch.delenv("PCSF_TABLE_LEAVES")
    tab = model.context(device=0, **kw)
    return model, full, pairs, tab


@pytest.mark.parametrize("name", ["53birds", "58mammals", "100vertebrates", "7yeast", "12flies", "29mammals"])
def test_cherry_tables_give_the_full_programs_bits(name, monkeypatch):
    model, full, pairs_only, tab = _contexts(name, monkeypatch, with_bls=True)
    try:
        assert model.nl >= 7
        red = __import__("phylocsfpp_amd").build_cherry_schedule(model.child1, model.child2, triples=True)
        assert red is not None and len(red[3]) >= 1
        fi, pi, ti = full.tracks_program_info(), pairs_only.tracks_program_info(), tab.tracks_program_info()
        assert fi["cherries"] == 0 and fi["products_run"] == model.nl - 2
        assert pi["cherries"] == len(red[3]) and pi["triples"] == 0 and pi["products_run"] == model.nl - 2 - len(red[3])
        assert ti["cherries"] == len(red[3]) and ti["triples"] == len(red[4]) and ti["products_run"] == model.nl - 2 - len(red[3]) - len(red[4])
        assert ti["rows_per_table_set"] == model.nl * 65 + len(red[3]) * 65 ** 2 + len(red[4]) * 65 ** 3
        cod = synth.synthetic_codon_columns(model.child1, model.child2, model.branch_len, model.S_c, model.pi_c, model.S_nc, model.pi_nc, 4000, seed=3)
        rng = np.random.default_rng(1)
        cod[:, 100:140] = 64                                   # all-gap columns
        cod[rng.integers(0, model.nl, 500), rng.integers(0, 4000, 500)] = 64
        cod[::2, 200:300] = 64                                 # one leaf of many cherries a gap
        cod[:, 300:400] = rng.integers(0, 65, size=(model.nl, 100))   # unrelated codons: every pair row is somebody's
        a, b, b2 = full.score_sites_fixed(cod), tab.score_sites_fixed(cod), pairs_only.score_sites_fixed(cod)
        assert np.array_equal(a, b, equal_nan=True), np.abs(a - b).max()
        assert np.array_equal(a, b2, equal_nan=True), np.abs(a - b2).max()
        assert np.isfinite(a[:100]).all()
        # the six-frame entry point (fused front end, all-gap subtrees skipped per work-group) and the branch-length scores beside it
        nuc = np.frombuffer(b"ACGTacgtN-", dtype=np.uint8)[rng.integers(0, 10, size=(model.nl, 3000))]
        nuc[model.nl // 2:, 500:1500] = ord("-")               # half the species missing over a long stretch
        nuc[:, 2000:2200] = ord("N")
        alns = [(0, 1700, 1001, 50000), (1700, 1300, 77, 90000)]
        ra, _, ba = full.score_alignments_6frames(nuc, alns, with_bls=True)
        for other in (tab, pairs_only):
            rb, _, bb = other.score_alignments_6frames(nuc, alns, with_bls=True)
            assert np.array_equal(ba, bb)
            for k in range(2):
                for f in range(6):
                    assert np.array_equal(ra[k][f], rb[k][f], equal_nan=True), (k, f)
        # FIXED block scores run the same kernel
        off = np.array([0, 40, 41, 140, 1000, 4000], dtype=np.uint64)
        pa_, aa = full.score_blocks(1, cod, off, compute_anc=True)
        pb_, ab = tab.score_blocks(1, cod, off, compute_anc=True)
        assert np.array_equal(pa_, pb_, equal_nan=True) and np.array_equal(aa, ab, equal_nan=True)
    finally:
        full.close()
        pairs_only.close()
        tab.close()


def test_cherry_tables_on_a_tree_deeper_than_the_register_stack(monkeypatch):
    """a balanced 64-leaf tree: 32 cherries, five parked tiles in the full program (the DEEP variants), fewer in the reduced one"""
    import phylocsfpp_amd as pa
    m = pa.load_model("58mammals")
    nl = 64
    c1, c2 = [-1] * (2 * nl - 1), [-1] * (2 * nl - 1)
    nxt = [nl]

    def build(lo, hi):
        if hi - lo == 1:
            return lo
        mid = (lo + hi) // 2
        a, b = build(lo, mid), build(mid, hi)
        i = nxt[0]
        nxt[0] += 1
        c1[i], c2[i] = a, b
        return i
    build(0, nl)
    rng = np.random.default_rng(4)
    bl = rng.uniform(0.01, 0.3, size=2 * nl - 1).astype(np.float32)
    c1, c2 = np.array(c1, np.int16), np.array(c2, np.int16)
    prog, slots, depth, pairs = pa.build_cherry_schedule(c1, c2)
    assert len(pairs) == 32 and pa.build_schedule(c1, c2)[2] == 5 and depth == 4      # (no node over a cherry and a leaf in a balanced tree)
    ctxs = []
    for env in ("0", None):
        if env is None:
            monkeypatch.delenv("PCSF_CHERRY_TABLES", raising=False)
        else:
            monkeypatch.setenv("PCSF_CHERRY_TABLES", env)
        ctxs.append(pa.ScoringContext(c1, c2, bl, m.S_c, m.pi_c, m.S_nc, m.pi_nc, device=0))
    try:
        cod = synth.synthetic_codon_columns(c1, c2, bl, m.S_c, m.pi_c, m.S_nc, m.pi_nc, 1000, seed=8)
        cod[:, 10:20] = 64
        cod[:32, 20:90] = 64
        a, b = ctxs[0].score_sites_fixed(cod), ctxs[1].score_sites_fixed(cod)
        assert np.array_equal(a, b) and np.isfinite(a).all()
    finally:
        for c in ctxs:
            c.close()


def test_a_table_budget_too_small_for_the_triples_keeps_the_cherries(monkeypatch):
    """PCSF_TABLE_BYTES: the rows of both models must fit; 58mammals' triples need 2.6 GB, its cherries 77 MB"""
    import phylocsfpp_amd as pa
    model = pa.load_model("58mammals")
    cod = synth.synthetic_codon_columns(model.child1, model.child2, model.branch_len, model.S_c, model.pi_c, model.S_nc, model.pi_nc, 500, seed=2)
    got = {}
    for budget, want in (("1000000000", (17, 0)), ("1000000", (0, 0)), (None, (17, 9))):
        if budget is None:
            monkeypatch.delenv("PCSF_TABLE_BYTES", raising=False)
        else:
            monkeypatch.setenv("PCSF_TABLE_BYTES", budget)
        c = model.context(device=0)
        try:
            info = c.tracks_program_info()
            assert (info["cherries"], info["triples"]) == want, (budget, info)
            got[budget] = c.score_sites_fixed(cod)
        finally:
            c.close()
    assert np.array_equal(got["1000000000"], got[None]) and np.array_equal(got["1000000"], got[None])


def test_tables_on_a_tree_of_300_species(monkeypatch):
    """A random 300-leaf tree: more than 128 leaves (no all-gap skipping), a codon matrix that leaves one work-group per CU either way, ~100
    cherries; its ~50 triples would take 14 GB and stay out under the default budget -- and come in under a larger one.  Same bits in all forms."""
    import phylocsfpp_amd as pa
    from test_schedule import _random_tree
    m = pa.load_model("58mammals")
    rng = np.random.default_rng(12)
    nl = 300
    c1, c2 = _random_tree(nl, rng)
    bl = rng.uniform(0.005, 0.1, size=2 * nl - 1).astype(np.float32)
    red = pa.build_cherry_schedule(c1, c2, triples=True)
    n_pairs, n_triples = len(red[3]), len(red[4])
    assert n_pairs > 60 and n_triples > 20
    cod = synth.synthetic_codon_columns(c1, c2, bl, m.S_c, m.pi_c, m.S_nc, m.pi_nc, 700, seed=8)
    cod[:, 10:20] = 64
    cod[::3, 50:200] = 64
    got = {}
    for tag, env in (("full", {"PCSF_CHERRY_TABLES": "0"}), ("default", {}), ("roomy", {"PCSF_TABLE_BYTES": str(40 << 30)})):
        for k in ("PCSF_CHERRY_TABLES", "PCSF_TABLE_BYTES"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c = pa.ScoringContext(c1, c2, bl, m.S_c, m.pi_c, m.S_nc, m.pi_nc, device=0)
        try:
            info = c.tracks_program_info()
            got[tag] = (c.score_sites_fixed(cod), info)
        finally:
            c.close()
    assert got["full"][1]["cherries"] == 0
    assert got["default"][1]["cherries"] == n_pairs and got["default"][1]["triples"] == 0          # 2 x n_triples x 140 MB > 8 GiB
    assert got["roomy"][1]["triples"] == n_triples and got["roomy"][1]["products_run"] == nl - 2 - n_pairs - n_triples
    assert np.isfinite(got["full"][0]).all()
    assert np.array_equal(got["full"][0], got["default"][0]) and np.array_equal(got["full"][0], got["roomy"][0])


def test_tables_on_the_smallest_trees(monkeypatch):
    """Trees of 2 to 9 leaves, random shapes: a tree whose only internal nodes are table nodes and the root runs a program without any product
    (three leaves: one cherry and a leaf under the root); two leaves: nothing to tabulate.  Same bits as the full program, finite scores."""
    import phylocsfpp_amd as pa
    from test_schedule import _random_tree
    m = pa.load_model("58mammals")
    rng = np.random.default_rng(21)
    seen_zero_product = False
    for nl in (2, 3, 3, 4, 4, 4, 5, 5, 6, 7, 9):
        c1, c2 = _random_tree(nl, rng)
        bl = rng.uniform(0.01, 0.2, size=2 * nl - 1).astype(np.float32)
        cod = synth.synthetic_codon_columns(c1, c2, bl, m.S_c, m.pi_c, m.S_nc, m.pi_nc, 200, seed=nl)
        cod[:, 5] = 64
        cod[0, 10:30] = 64
        res = []
        for env in ("0", None):
            if env is None:
                monkeypatch.delenv("PCSF_CHERRY_TABLES", raising=False)
            else:
                monkeypatch.setenv("PCSF_CHERRY_TABLES", env)
            c = pa.ScoringContext(c1, c2, bl, m.S_c, m.pi_c, m.S_nc, m.pi_nc, device=0)
            try:
                res.append((c.score_sites_fixed(cod), c.tracks_program_info()))
            finally:
                c.close()
        assert np.array_equal(res[0][0], res[1][0]) and np.isfinite(res[0][0]).all(), nl
        assert res[1][1]["products_run"] == nl - 2 - res[1][1]["cherries"] - res[1][1]["triples"]
        seen_zero_product |= res[1][1]["products_run"] == 0 and res[1][1]["cherries"] > 0
    assert seen_zero_product


def test_quad_tables_give_the_same_bits(monkeypatch):
    """PCSF_TABLE_LEAVES=4 (opt-in: 65^4 rows = 9.1 GB per node and model): a node over two cherries, and a node over a triple and a leaf, as
    leaves of the program.  Two small trees with one such node each (18 GB of rows per context), against the default form and the full program."""
    import phylocsfpp_amd as pa
    m = pa.load_model("58mammals")
    rng = np.random.default_rng(31)
    trees = {
        # leaves 0..6; ((0,1),(2,3)) is a node over two cherries; the rest keeps it away from the root
        "two cherries": ([-1] * 7 + [0, 2, 7, 4, 10, 11], [-1] * 7 + [1, 3, 8, 5, 9, 6]),
        # (((0,1),2),3): a triple and a leaf
        "triple and leaf": ([-1] * 7 + [0, 7, 8, 4, 10, 11], [-1] * 7 + [1, 2, 3, 5, 9, 6]),
    }
    for tag, (c1, c2) in trees.items():
        c1, c2 = np.array(c1, np.int16), np.array(c2, np.int16)
        nl = 7
        red = pa.build_table_schedule(c1, c2, 4)
        assert red is not None and any(x >= 14336 for x in red[3]), tag                  # a quad operand
        bl = rng.uniform(0.01, 0.2, size=2 * nl - 1).astype(np.float32)
        cod = synth.synthetic_codon_columns(c1, c2, bl, m.S_c, m.pi_c, m.S_nc, m.pi_nc, 3000, seed=3)
        cod[:, 7] = 64
        cod[1, 100:400] = 64
        cod[:, 1000:1200] = rng.integers(0, 65, size=(nl, 200))                          # unrelated codons: cold rows of the big table
        got = {}
        for form, env in (("full", {"PCSF_CHERRY_TABLES": "0"}), ("default", {}), ("quads", {"PCSF_TABLE_LEAVES": "4", "PCSF_TABLE_BYTES": str(40 << 30)})):
            for k in ("PCSF_CHERRY_TABLES", "PCSF_TABLE_LEAVES", "PCSF_TABLE_BYTES"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            c = pa.ScoringContext(c1, c2, bl, m.S_c, m.pi_c, m.S_nc, m.pi_nc, device=0)
            try:
                got[form] = (c.score_sites_fixed(cod), c.tracks_program_info())
            finally:
                c.close()
        assert got["quads"][1]["quads"] == 1 and got["default"][1]["quads"] == 0, (tag, got["quads"][1])
        assert got["quads"][1]["products_run"] == got["default"][1]["products_run"] - 1
        assert np.isfinite(got["full"][0]).all()
        assert np.array_equal(got["full"][0], got["default"][0]) and np.array_equal(got["full"][0], got["quads"][0]), tag


@pytest.mark.parametrize("form", ["full", "leaves2", "default"])
def test_command_line_writes_the_golden_tracks_in_every_form(tmp_path, form):
    """phylocsf-amd build-tracks on the reference's example MAF under --table-leaves 2, the default, and PCSF_CHERRY_TABLES=0: the reference's
    golden files, byte for byte (six-frame entry point, fused front end, all-gap subtrees skipped, text formatted on the device)."""
    import gzip
    import shutil
    import subprocess
    gold = os.path.join(ROOT, "tests", "golden")
    cli = os.path.join(ROOT, "phylocsfpp_amd", "host", "phylocsf-amd")
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(cli)])
    maf = tmp_path / "galGal6.maf"
    with gzip.open(os.path.join(gold, "build-tracks", "galGal6_chr22_25_28_each_30k_bases.maf.gz"), "rb") as fi, open(maf, "wb") as fo:
        shutil.copyfileobj(fi, fo)
    out = tmp_path / "out"
    env = dict(os.environ)
    env.pop("PCSF_CHERRY_TABLES", None)
    extra = []
    if form == "full":
        env["PCSF_CHERRY_TABLES"] = "0"
    elif form == "leaves2":
        extra = ["--table-leaves", "2"]
    subprocess.check_call([cli, "build-tracks", "--output", str(out)] + extra + [os.path.join(gold, "models", "53birds"), str(maf)], env=env, stderr=subprocess.DEVNULL)
    for n in ["PhyloCSFpower"] + ["PhyloCSFRaw%s%d" % (s, f) for s in "+-" for f in (1, 2, 3)]:
        want = gzip.open(os.path.join(gold, "build-tracks", n + ".wig.gz"), "rb").read()
        assert open(out / (n + ".wig"), "rb").read() == want, (form, n)
