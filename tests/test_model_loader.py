"""The package's own model reader (C-ABI pcsf_model_open -> phylocsfpp_amd.load_model; reference load_model,
src/models.hpp:1757-1857) against the oracle's independent C readers (oracle/pcsf_oracle.c po_model_open): identical
arrays for the three shipped models, the species-name table, and the --species reduction.  Host-only code: runs
without a GPU; the gpu-marked twin checks the same thing on the box where the contexts are built from it."""
import os

import numpy as np
import pytest

import oracle_binding as ob
import phylocsfpp_amd as pa

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
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
