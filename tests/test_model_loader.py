"""The package's own model reader (C-ABI pcsf_model_open -> phylocsfpp_amd.load_model; reference load_model,
src/models.hpp:1757-1857) against the oracle's independent C readers (oracle/pcsf_oracle.c po_model_open): identical
arrays for the three shipped models, the species-name table, and the --species reduction.  Host-only code: runs
without a GPU; the gpu-marked twin checks the same thing on the box where the contexts are built from it."""
import os

import numpy as np
import pytest

import oracle_binding as ob
import phylocsfpp_amd as pa

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
FIELDS = ("child1", "child2", "parent", "branch_len", "branch_len_d", "S_c", "pi_c", "S_nc", "pi_nc")


def _same(name):
    prefix = os.path.join(GOLD, "models", name)
    m, o = pa.load_model(prefix), ob.OracleModel(prefix)
    assert (m.n, m.nl) == (o.n, o.nl)
    for k in FIELDS:
        a, b = getattr(m, k), getattr(o, k)
        assert a.dtype == b.dtype and a.tobytes() == b.tobytes(), (name, k)
    assert m.labels == o.labels
    return m


@pytest.mark.parametrize("name", ["53birds", "58mammals", "100vertebrates"])
def test_package_loader_equals_oracle_loader(name):
    _same(name)


@pytest.mark.gpu
def test_package_loader_equals_oracle_loader_on_the_gpu_box():
    for name in ("53birds", "58mammals", "100vertebrates"):
        m = _same(name)
        ctx = m.context(device=0)
        assert ctx.n_leaves == m.nl
        ctx.close()


def test_species_names_and_reduction():
    prefix = os.path.join(GOLD, "models", "58mammals")
    m = pa.load_model(prefix)
    assert m.leaf_of("hg38") == m.leaf_of("Human") == m.labels.index("human") == 0
    assert m.leaf_of("no-such-species") == -1
    r = pa.load_model(prefix, species="human,mm10,dog")
    assert r.nl == 3 and sorted(r.labels[:3]) == ["dog", "human", "mouse"]
    # total tree length spanned by the three species is preserved by the merge rule (branch lengths are ADDED)
    full_parent = m.parent
    keep = {m.labels.index(x) for x in ("human", "mouse", "dog")}
    on_path = set()
    for leaf in keep:
        v = leaf
        while v >= 0:
            on_path.add(v)
            v = int(full_parent[v])
    # nodes on the union of the root paths, minus the part above the three species' last common ancestor
    def below(v):
        return {l for l in keep if _is_ancestor(full_parent, v, l)}
    lca_and_above = {v for v in on_path if below(v) == keep}
    want = sum(float(m.branch_len_d[v]) for v in on_path - lca_and_above)
    got = float(r.branch_len_d[:r.n - 1].sum())
    # the reduced root's two children keep their full path lengths to the last common ancestor
    assert abs(got - want) < 1e-12
    with pytest.raises(pa.PcsfError):
        pa.load_model(prefix, species="human,unicorn")
    with pytest.raises(pa.PcsfError):
        pa.load_model(os.path.join(GOLD, "models", "no-such-model"))


def _is_ancestor(parent, anc, v):
    while v >= 0:
        if v == anc:
            return True
        v = int(parent[v])
    return False


def test_shipped_models_resolve_by_name_like_the_references_command_line(tmp_path, monkeypatch):
    """`phylocsf++ build-tracks 58mammals x.maf` names a built-in model (reference src/models.hpp:1443-1455, load_model :1757-1783); here
    the models ship as files next to the library, and a name without a slash that is not a file prefix is looked up there."""
    import numpy as np
    import phylocsfpp_amd as pa
    for name in ("53birds", "58mammals", "100vertebrates"):
        by_name = pa.load_model(name)
        by_path = pa.load_model(os.path.join(GOLD, "models", name))
        assert by_name.nl == by_path.nl and by_name.labels == by_path.labels
        assert np.array_equal(by_name.S_c, by_path.S_c) and np.array_equal(by_name.pi_nc, by_path.pi_nc)
        assert np.array_equal(by_name.branch_len, by_path.branch_len)
    assert pa.load_model("58mammals").leaf_of("hg38") == pa.load_model("58mammals").leaf_of("human")   # the alias table ships with the models
    with pytest.raises(pa.PcsfError, match="neither a shipped model"):
        pa.load_model("59mammals")
    # a prefix that names existing files wins over the shipped model of the same name
    import shutil
    for ext in ("_coding.ECM", "_noncoding.ECM", ".nh"):
        shutil.copyfile(os.path.join(GOLD, "models", "53birds" + ext), str(tmp_path / ("58mammals" + ext)))
    monkeypatch.chdir(tmp_path)
    assert pa.load_model("58mammals").nl == 53
    # PCSF_MODEL_DIR comes first
    monkeypatch.chdir("/")
    d = tmp_path / "mine"
    d.mkdir()
    for ext in ("_coding.ECM", "_noncoding.ECM", ".nh"):
        shutil.copyfile(os.path.join(GOLD, "models", "53birds" + ext), str(d / ("mymodel" + ext)))
    monkeypatch.setenv("PCSF_MODEL_DIR", str(d))
    assert pa.load_model("mymodel").nl == 53


ALL_MODELS = {"20flies": 20, "26worms": 26, "12flies": 12, "23flies": 23, "100vertebrates": 100, "49birds": 49, "53birds": 53, "7yeast": 7, "58mammals": 58,
              "21mosquitoes": 21, "29mammals": 29}


@pytest.mark.parametrize("name", sorted(ALL_MODELS))
def test_every_model_the_reference_compiles_in_resolves_by_name(name):
    """The reference's command line takes any of its eleven built-in models by name (src/models.hpp:1443-1455); they ship here as files in
    the reference's own format (tests/golden/make_models.py wrote them from the reference's literals, digit for digit).  Every one loads
    through the package's reader AND the oracle's independent readers with the same arrays; the tables are what an ECM must be (symmetric,
    zero diagonal, frequencies that sum to 1); the oracle scores columns under it."""
    import numpy as np
    import phylocsfpp_amd as pa
    import oracle_binding as ob
    m = pa.load_model(name)
    assert m.nl == ALL_MODELS[name] and m.n == 2 * m.nl - 1
    prefix = os.path.join(ROOT, "phylocsfpp_amd", "models", name)
    o = ob.OracleModel(prefix)
    assert o.nl == m.nl and o.labels == m.labels
    for a, b in ((m.S_c, o.S_c), (m.S_nc, o.S_nc), (m.pi_c, o.pi_c), (m.pi_nc, o.pi_nc), (m.branch_len, o.branch_len), (m.child1, o.child1), (m.child2, o.child2)):
        assert np.array_equal(np.asarray(a).reshape(-1), np.asarray(b).reshape(-1))
    for S, pi in ((o.S_c, o.pi_c), (o.S_nc, o.pi_nc)):
        assert np.array_equal(S, S.T) and not S.diagonal().any() and (S >= 0).all() and abs(pi.sum() - 1.0) < 1e-4 and (pi > 0).all()
    from phylocsfpp_amd import synth
    cod = synth.synthetic_codon_columns(o.child1, o.child2, o.branch_len, o.S_c, o.pi_c, o.S_nc, o.pi_nc, 50, seed=len(name))   # columns evolved down the tree
    s = o.run_tracks(cod)
    assert np.all(np.isfinite(s)) and np.abs(s).max() > 0.1
    # every tree label has its row in the name table (a MAF may use any of the species' other names)
    aliases = {l.split("\t")[0] for l in open(os.path.join(ROOT, "phylocsfpp_amd", "models", "species_aliases.tsv"))}
    leaves = [l for l in m.labels if l]
    assert len(leaves) == m.nl and all(m.leaf_of(l) >= 0 for l in leaves)
    assert sum(l in aliases for l in leaves) >= m.nl - 2, [l for l in leaves if l not in aliases]
