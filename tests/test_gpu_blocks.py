"""GPU parity of whole-block scoring (run, src/run.hpp:57-211) through the C-ABI: FIXED, MLE, OMEGA
against the CPU oracle on the same blocks, and against the reference's golden .scores files."""
import ctypes as C
import os

import numpy as np
import pytest

import maf_blocks
import oracle_binding as ob
import phylocsfpp_amd as pa

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SMALL = os.path.join(GOLD, "score-msa", "chr22.50alignments.maf")
ALIASES = os.path.join(GOLD, "models", "species_aliases.tsv")


def _gold(name):
    rows = [l.rstrip("\n").split("\t") for l in open(os.path.join(GOLD, "score-msa", name))][1:]
    return rows


def _oracle_blocks(m, algo, cod, off, compute_anc=True):
    nb = len(off) - 1
    ph = np.full(nb, np.nan, np.float32)
    an = np.full(nb, np.nan, np.float32)
    lib = ob.lib()
    lib.po_run_blocks.argtypes = [C.POINTER(ob.PoModel), C.c_int, C.POINTER(C.c_uint8), C.POINTER(C.c_uint64), C.c_size_t,
                                  C.c_size_t, C.c_int, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int]
    lib.po_run_blocks(C.byref(m.m), algo, cod.ctypes.data_as(C.POINTER(C.c_uint8)), off.ctypes.data_as(C.POINTER(C.c_uint64)),
                      nb, cod.shape[1], int(compute_anc), 42, ph.ctypes.data_as(C.POINTER(C.c_float)),
                      an.ctypes.data_as(C.POINTER(C.c_float)), 0)
    return ph, an


@pytest.fixture(scope="module")
def small_blocks(oracle_models):
    m = oracle_models("100vertebrates")
    cod, off = maf_blocks.block_codons(SMALL, m.labels, ALIASES)
    return m, cod, off


def test_fixed_blocks_match_golden_and_oracle(small_blocks, gpu_contexts):
    m, cod, off = small_blocks
    ctx = gpu_contexts("100vertebrates")
    ph, an = ctx.score_blocks(pa.FIXED, cod, off, compute_anc=True)
    gold = _gold("chr22.50alignments.fixed.scores")
    assert ["%.6f" % v for v in ph] == [g[4] for g in gold]         # the reference's test is an exact diff
    assert ["%.6f" % v for v in an] == [g[5] for g in gold]
    oph, oan = _oracle_blocks(m, ob.FIXED, cod, off)
    np.testing.assert_allclose(ph, oph, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(an, oan, rtol=1e-6, atol=1e-6)


def test_mle_blocks(small_blocks, gpu_contexts):
    m, cod, off = small_blocks
    ctx = gpu_contexts("100vertebrates")
    ph, an = ctx.score_blocks(pa.MLE, cod, off, compute_anc=True)
    gold = _gold("chr22.50alignments.mle.scores")
    gp = np.array([float(g[4]) for g in gold])
    ga = np.array([float(g[5]) for g in gold])
    assert np.all((ph - gp) ** 2 <= 0.001) and np.all((an - ga) ** 2 <= 0.001)   # reference test/tests.sh:41
    assert np.sum((np.abs(ph - gp) <= 5e-6) & (np.abs(an - ga) <= 5e-6)) >= 48
    print("MLE rows equal to the golden at print precision:", int(np.sum((np.abs(ph - gp) <= 5e-6) & (np.abs(an - ga) <= 5e-6))))
    # against the oracle (same optimiser, same evaluation order): 1e-6 relative.  Blocks 5 and 48 sit on a
    # near-tie inside Brent (the oracle itself differs from the GSL golden there by 1.5e-4): a 1e-13
    # difference in f re-routes them, so up to three rows may take the other branch.
    oph, oan = _oracle_blocks(m, ob.MLE, cod, off)
    close = np.isclose(ph, oph, rtol=1e-6, atol=2e-6) & np.isclose(an, oan, rtol=1e-6, atol=2e-6)
    assert close.sum() >= 47, (ph - oph, an - oan)
    assert np.all((ph - oph) ** 2 <= 0.001)


def test_omega_blocks(small_blocks, gpu_contexts):
    m, cod, off = small_blocks
    ctx = gpu_contexts("100vertebrates")
    ph, an = ctx.score_blocks(pa.OMEGA, cod, off, compute_anc=False)
    assert np.all(np.isnan(an))
    gold = _gold("chr22.50alignments.omega.scores")
    n_cod = np.diff(off.astype(np.int64))
    for i, g in enumerate(gold):
        if n_cod[i] == 0:
            assert np.isnan(ph[i]) and g[4] == "nan"
        elif g[4] != "nan":
            assert (ph[i] - float(g[4])) ** 2 <= 0.1                 # reference test/tests.sh:46
        else:
            assert np.isfinite(ph[i])                                 # GSL-degradation NaN rows: parity unpinned
    sel = np.nonzero(n_cod > 0)[0][:16]
    sub_off = np.concatenate([[0], np.cumsum(n_cod[sel])]).astype(np.uint64)
    sub = np.concatenate([cod[:, int(off[i]):int(off[i + 1])] for i in sel], axis=1)
    sub = np.ascontiguousarray(sub)
    gph, _ = ctx.score_blocks(pa.OMEGA, sub, sub_off, compute_anc=False)
    np.testing.assert_array_equal(gph, ph[sel])                       # batching must not change a block's result
    # GPU vs oracle: same optimiser logic (tests/test_optimizer.py pins it bit for bit) and objective values
    # equal to ~1e-12 (test_objective_values_match_oracle), but twelve chained 1-%-bracket Brent runs that
    # each report their LAST probe amplify that to the reference's own reproducibility level (tests.sh:46)
    oph, _ = _oracle_blocks(m, ob.OMEGA, sub, sub_off, compute_anc=False)
    assert np.all((gph - oph) ** 2 <= 0.1)


def _oracle_lpr(m, qd, cod, x):
    lpr, anc = C.c_double(), C.c_double()
    rc = ob.lib().po_lpr_leaves(C.byref(qd), C.byref(m.m.tree), x, cod.ctypes.data_as(C.POINTER(C.c_uint8)), cod.shape[1],
                                cod.shape[1], 0, C.byref(lpr), C.byref(anc), None)
    assert rc == 0
    return lpr.value


def test_objective_values_match_oracle(small_blocks, gpu_contexts):
    """One objective evaluation (batched expm on the GPU + block-mode pruning) against the oracle at fixed
    parameters: this is where the 1e-6 target of north_star is meaningful for MLE / OMEGA."""
    m, cod, off = small_blocks
    ctx = gpu_contexts("100vertebrates")
    n_cod = np.diff(off.astype(np.int64))
    sel = [i for i in np.argsort(-n_cod)[:6]] + [int(np.nonzero(n_cod == 1)[0][0])]
    subs = [np.ascontiguousarray(cod[:, int(off[i]):int(off[i + 1])]) for i in sel]
    sub = np.ascontiguousarray(np.concatenate(subs, axis=1))
    sub_off = np.concatenate([[0], np.cumsum([s.shape[1] for s in subs])]).astype(np.uint64)
    for model, coding in ((0, True), (1, False)):
        qd = m.qdiag(coding)
        for x in (0.01, 0.37, 1.0, 2.5, 10.0):
            got, failed = ctx.block_loglik(model, sub, sub_off, x)
            assert not failed.any()
            want = np.array([_oracle_lpr(m, qd, s, x) for s in subs])
            np.testing.assert_allclose(got, want, rtol=1e-9 if x >= 0.05 else 1e-6)
    # kappa/omega model at several (kappa, omega, sigma, ratios, rho)
    rng = np.random.default_rng(4)
    for kappa, omega, sigma, rho in ((2.5, 1.0, 1.0, 1.0), (1.0, 0.2, 0.01, 0.3), (7.3, 0.2, 0.01, 5.9), (2.8, 1.0, 1.0, 0.02)):
        qs = np.concatenate([[kappa, omega, sigma], rng.uniform(0.5, 1.5, 9)])
        Q, pi = np.empty((64, 64)), np.empty(64)
        dp = C.POINTER(C.c_double)
        ob.lib().po_q_from_omega(qs.ctypes.data_as(dp), Q.ctypes.data_as(dp), pi.ctypes.data_as(dp))
        qd = ob.PoQdiag()
        ob.lib().po_eig(Q.ctypes.data_as(dp), pi.ctypes.data_as(dp), C.byref(qd))
        got, failed = ctx.block_loglik(2, sub, sub_off, rho, q_settings=qs)
        assert not failed.any()
        want = np.array([_oracle_lpr(m, qd, s, rho) for s in subs])
        # Here the eigen-systems come from two different solvers (device: parallel-order Jacobi, oracle: row-cyclic
        # Jacobi).  The objective itself is only defined to that level: the reference clamps noise-level negative
        # entries of P to zero and rebuilds the diagonal, so columns that need an a-priori impossible double change
        # see the solver's rounding -- the oracle's Jacobi and LAPACK's eigh differ by 3e-9 (rho = 0.3, sigma = 0.01)
        # and 2e-6 (rho = 0.02) on these very blocks.
        np.testing.assert_allclose(got, want, rtol=2e-8 if rho >= 0.05 else 1e-5)


def test_device_eigen_solver_of_the_omega_rate_matrix(gpu_contexts):
    """eigen.hip (parallel-order Jacobi, one work-group per matrix) against the oracle's rate matrix: Q = R diag(lambda) L,
    L R = I, the spectrum of the symmetrised matrix, and the equilibrium prior pi / sum(pi)."""
    ctx = gpu_contexts("100vertebrates")
    rng = np.random.default_rng(12)
    qs = np.empty((97, 12))
    qs[:, 0] = rng.uniform(1.0, 10.0, 97)                 # kappa
    qs[:, 1] = rng.choice([1.0, 0.2, 0.05, 3.0], 97)       # omega
    qs[:, 2] = rng.choice([1.0, 0.01, 0.5], 97)            # sigma
    qs[:, 3:] = np.exp(rng.uniform(np.log(0.05), np.log(20.0), (97, 9)))
    qs[0] = [2.5, 1.0, 1.0] + [1.0] * 9                    # the uniform starting point of every block
    qs[1] = [1.0, 0.2, 0.01] + [1.0] * 9
    qs[96, 5] = -0.5                                       # a negative frequency ratio: invalid model
    right, left, lam, prior, failed = ctx.debug_eigen_omega(qs)
    assert failed[96] == 1 and not failed[:96].any()
    dp = C.POINTER(C.c_double)
    for i in range(96):
        Q, pi = np.empty((64, 64)), np.empty(64)
        ob.lib().po_q_from_omega(np.ascontiguousarray(qs[i]).ctypes.data_as(dp), Q.ctypes.data_as(dp), pi.ctypes.data_as(dp))
        scale = np.abs(Q).max()
        assert np.abs((right[i] * lam[i]) @ left[i] - Q).max() <= 2e-13 * scale
        assert np.abs(left[i] @ right[i] - np.eye(64)).max() <= 1e-12
        B = np.sqrt(pi)[:, None] * Q / np.sqrt(pi)[None, :]
        want = np.linalg.eigvalsh(0.5 * (B + B.T))
        np.testing.assert_allclose(np.sort(lam[i]), want, rtol=1e-11, atol=1e-14 * scale)
        np.testing.assert_allclose(prior[i], pi / pi.sum(), rtol=1e-11)
        assert abs(prior[i].sum() - 1.0) < 1e-14
    # warm start (what every kappa probe of OMEGA does): the iteration begins in the eigenbasis of a neighbouring -- or,
    # for random restarts and the H0 -> H1 switch, a distant -- matrix and must end at the same eigen-system
    ws = qs[:96].copy()
    ws[:48, 0] *= rng.uniform(0.97, 1.03, 48)              # kappa moved by a Brent step
    ws[48:, 0] = rng.uniform(1.0, 10.0, 48)                # kappa anywhere
    ws[80:, 1:3] = [1.0, 1.0]                              # and omega / sigma as under H0
    ws[:, 0] = np.clip(ws[:, 0], 1.0, 10.0)
    r2, l2, lam2, prior2, failed2 = ctx.debug_eigen_omega(qs[:96], warm_settings=ws)
    assert not failed2.any()
    for i in range(96):
        Q, pi = np.empty((64, 64)), np.empty(64)
        ob.lib().po_q_from_omega(np.ascontiguousarray(qs[i]).ctypes.data_as(dp), Q.ctypes.data_as(dp), pi.ctypes.data_as(dp))
        scale = np.abs(Q).max()
        assert np.abs((r2[i] * lam2[i]) @ l2[i] - Q).max() <= 2e-13 * scale
        assert np.abs(l2[i] @ r2[i] - np.eye(64)).max() <= 1e-12
        np.testing.assert_allclose(np.sort(lam2[i]), np.sort(lam[i]), rtol=1e-11, atol=1e-14 * scale)
        np.testing.assert_allclose(prior2[i], prior[i], rtol=1e-11)


def test_random_blocks_fixed_with_empty_and_ragged(oracle_models, gpu_contexts):
    from phylocsfpp_amd import synth
    m = oracle_models("58mammals")
    ctx = gpu_contexts("58mammals")
    cod = synth.synthetic_codon_columns(m.child1, m.child2, m.branch_len, m.S_c, m.pi_c, m.S_nc, m.pi_nc, 700, seed=21)
    off = np.array([0, 0, 1, 17, 17, 81, 145, 146, 700], dtype=np.uint64)
    ph, an = ctx.score_blocks(pa.FIXED, cod, off)
    oph, oan = _oracle_blocks(m, ob.FIXED, cod, off)
    np.testing.assert_allclose(ph, oph, rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(an, oan, rtol=1e-6, atol=1e-6)
    assert ph[0] == 0.0 and ph[3] == 0.0
    with pytest.raises(pa.PcsfError):
        ctx.score_blocks(pa.FIXED, cod, np.array([0, 5, 3], dtype=np.uint64))


def test_mle_and_omega_on_trees_whose_branch_count_is_not_a_chunk_multiple(oracle_models, gpu_contexts):
    """53birds (104 branches) and 58mammals (114): the batched P(t) kernel's nine-job chunks straddle evaluations, so a
    work-group changes eigen-system in mid-chunk (100vertebrates' 198 branches never do)."""
    from phylocsfpp_amd import synth
    for name in ("53birds", "58mammals"):
        m = oracle_models(name)
        ctx = gpu_contexts(name)
        lens = [7, 1, 30, 12, 3]
        cols = synth.synthetic_codon_columns(m.child1, m.child2, m.branch_len, m.S_c, m.pi_c, m.S_nc, m.pi_nc, sum(lens), seed=21, dropout=0.2, gap_rate=0.05)
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
        # objective level, kappa/omega model: every block its own rate matrix -> one eigen-system per evaluation
        rng = np.random.default_rng(8)
        qs = np.column_stack([rng.uniform(1.5, 6.0, len(lens)), np.full(len(lens), 0.2), np.full(len(lens), 0.5), rng.uniform(0.6, 1.6, (len(lens), 9))])
        got, failed = ctx.block_loglik(2, cols, off, np.array([0.4, 1.0, 2.2, 0.9, 1.7]), q_settings=qs)
        assert not failed.any()
        dp = C.POINTER(C.c_double)
        for b in range(len(lens)):
            Q, pi = np.empty((64, 64)), np.empty(64)
            ob.lib().po_q_from_omega(np.ascontiguousarray(qs[b]).ctypes.data_as(dp), Q.ctypes.data_as(dp), pi.ctypes.data_as(dp))
            qd = ob.PoQdiag()
            ob.lib().po_eig(Q.ctypes.data_as(dp), pi.ctypes.data_as(dp), C.byref(qd))
            want = _oracle_lpr(m, qd, np.ascontiguousarray(cols[:, int(off[b]):int(off[b + 1])]), [0.4, 1.0, 2.2, 0.9, 1.7][b])
            assert abs(got[b] - want) <= 2e-8 * abs(want)
        # end to end against the oracle's own optimiser runs
        for algo, tol in ((pa.MLE, 0.001), (pa.OMEGA, 0.1)):
            ph, _ = ctx.score_blocks(algo, cols, off, compute_anc=False)
            oph, _ = _oracle_blocks(m, algo, cols, off, compute_anc=False)
            assert np.all(np.isfinite(ph) == np.isfinite(oph))
            ok = np.isfinite(oph)
            assert np.all((ph[ok] - oph[ok]) ** 2 <= tol), (name, algo, ph, oph)


def test_device_eigen_solver_in_its_lds_form(gpu_contexts, monkeypatch):
    """launch_eigen_omega runs the seven-wave form of the Jacobi kernel unless PCSF_EIGEN_LDS=1 / PCSF_EIGEN_FORM (read at every launch)
    ask for an earlier one; the first version, matrices in LDS: the same accuracy test under that setting."""
    monkeypatch.setenv("PCSF_EIGEN_LDS", "1")
    test_device_eigen_solver_of_the_omega_rate_matrix(gpu_contexts)


def test_device_eigen_solver_in_its_two_wave_form(gpu_contexts, monkeypatch):
    """PCSF_EIGEN_FORM=1: rounds 2-4's register-resident form (wave 0 holds B in 8x8 blocks, wave 1 V; plain rotations) -- the same
    accuracy test.  (The default, FORM 2: a rotation wave, B in quads of lanes over four waves, V on two waves, scaled rotations.)"""
    monkeypatch.setenv("PCSF_EIGEN_FORM", "1")
    test_device_eigen_solver_of_the_omega_rate_matrix(gpu_contexts)


def test_eigen_solver_flags_a_matrix_it_cannot_diagonalise(gpu_contexts, monkeypatch):
    """A NaN among the settings makes every stop test of the Jacobi iteration false: both device forms must use up their rounds and
    FLAG the job (fail -> NaN row) instead of handing whatever the registers hold to the pruning kernels."""
    ctx = gpu_contexts("100vertebrates")
    qs = np.array([[2.5, 1.0, 1.0] + [1.0] * 9, [float("nan"), 1.0, 1.0] + [1.0] * 9])
    for form in ("0", "1"):
        monkeypatch.setenv("PCSF_EIGEN_LDS", form)
        failed = ctx.debug_eigen_omega(qs)[-1]
        assert list(failed) == [0, 1], (form, failed)
    monkeypatch.delenv("PCSF_EIGEN_LDS")
    for form in ("1", "2"):
        monkeypatch.setenv("PCSF_EIGEN_FORM", form)
        failed = ctx.debug_eigen_omega(qs)[-1]
        assert list(failed) == [0, 1], ("PCSF_EIGEN_FORM", form, failed)


def test_block_groups_do_not_change_any_result(oracle_models, golden_dir):
    """score_blocks advances long lists as two concurrent groups (MLE from 512 blocks, OMEGA from 1,536; PCSF_BLOCK_GROUPS
    overrides): every block's result must be bit-identical to the single-group run."""
    import os
    import maf_blocks
    import phylocsfpp_amd as pa
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    m = oracle_models("100vertebrates")
    cod, offs = maf_blocks.block_codons(os.path.join(golden_dir, "score-msa", "chr22.516alignments.maf.gz"), m.labels,
                                        os.path.join(root, "tests", "golden", "models", "species_aliases.tsv"))
    reps, n0 = 3, offs[-1]
    cod = np.ascontiguousarray(np.tile(cod, (1, reps)))
    offs = np.concatenate([[0]] + [offs[1:] + r * n0 for r in range(reps)]).astype(np.uint64)
    assert len(offs) - 1 >= 1536
    ctx = pa.load_model(os.path.join(root, "tests", "golden", "models", "100vertebrates")).context()
    try:
        for algo, anc in ((pa.MLE, True), (pa.OMEGA, False)):
            os.environ["PCSF_BLOCK_GROUPS"] = "1"
            p1, a1 = ctx.score_blocks(algo, cod, offs, compute_anc=anc)
            del os.environ["PCSF_BLOCK_GROUPS"]
            p2, a2 = ctx.score_blocks(algo, cod, offs, compute_anc=anc)      # the default: two groups at this size
            os.environ["PCSF_BLOCK_GROUPS"] = "3"
            p3, a3 = ctx.score_blocks(algo, cod, offs, compute_anc=anc)
            assert p1.tobytes() == p2.tobytes() == p3.tobytes()
            if anc:
                assert a1.tobytes() == a2.tobytes() == a3.tobytes()
            # the three copies of the file are the same blocks: the same results
            nb = (len(offs) - 1) // reps
            assert p1[:nb].tobytes() == p1[nb:2 * nb].tobytes() == p1[2 * nb:].tobytes()
    finally:
        os.environ.pop("PCSF_BLOCK_GROUPS", None)
        ctx.close()
