"""The MLE / OMEGA optimiser ON THE DEVICE (phylocsfpp_amd/csrc/chain.hip + machine.hpp + optimizer.hpp) takes the decisions of the
reference's fit_find_init + GSL Brent (src/fixed_lik.hpp:469-544): the oracle's search records its evaluation points and values on
an analytic objective, the device's search is fed the recorded VALUES and must ask for exactly the recorded POINTS (equal histories
give equal requests: any decision taken differently shows as a different point).  The CPU twin of this test is tests/test_optimizer.py
(the same class compiled for the host).  And the two engines that drive the per-block machines -- the device's advance_blocks kernel and
the host's lock-step rounds (PCSF_BLOCK_ENGINE=host) -- give the same bits for every block, MLE and OMEGA, on both pruning kernels."""
import ctypes as C
import math
import os

import numpy as np
import pytest

import maf_blocks
import oracle_binding as ob
import phylocsfpp_amd as pa

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ALIASES = os.path.join(GOLD, "models", "species_aliases.tsv")
FN = C.CFUNCTYPE(C.c_double, C.c_double, C.c_void_p)

OBJECTIVES = [
    ("interior quadratic in log x", lambda x: -(math.log(x) - 0.3) ** 2, 1.0, 1e-2, 10.0),
    ("likelihood-like", lambda x: -50.0 * x + 120.0 * math.log(x) - 3.0 / x, 1.0, 1e-2, 10.0),
    ("maximum at the upper end", lambda x: x, 1.0, 1e-2, 10.0),                      # restarts exhaust -> hi
    ("maximum at the lower end", lambda x: -x, 1.0, 1e-2, 10.0),
    ("flat (a block without codons)", lambda x: 0.0, 1.0, 1e-2, 10.0),
    ("narrow peak found by restarts", lambda x: -abs(math.log(x) - 1.5) ** 0.5, 1.0, 1e-3, 10.0),
    ("kappa-like range", lambda x: -(x - 2.8) ** 2 + math.log(x), 2.5, 1.0, 10.0),
    ("asymmetric", lambda x: -math.exp(0.7 * x) + 4 * x, 1.0, 1e-3, 10.0),
]


@pytest.mark.parametrize("name,fn,init,lo,hi", OBJECTIVES)
@pytest.mark.parametrize("skip", [0, 7])
def test_device_search_asks_for_the_oracles_points(gpu_contexts, name, fn, init, lo, hi, skip):
    f = ob.lib().po_debug_max_lik
    f.argtypes = [FN, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_uint32, C.c_int, C.POINTER(C.c_double), C.c_int,
                  C.POINTER(C.c_int), C.POINTER(C.c_double)]
    xs = np.zeros(1200)
    n, last = C.c_int(), C.c_double()
    cb = FN(lambda x, _u: fn(x))
    assert f(cb, None, init, lo, hi, 42, skip, xs.ctypes.data_as(C.POINTER(C.c_double)), len(xs), C.byref(n), C.byref(last)) == 0
    xs = xs[:n.value]
    values = np.array([fn(x) for x in xs])
    ctx = gpu_contexts("53birds")
    got, n_req, last_dev = ctx.debug_maxlik_device(init, lo, hi, values, seed=42, skip_draws=skip)
    assert n_req == len(xs), name                      # -1: the device's search wanted more values than the oracle's took
    assert np.array_equal(got[:n_req], xs), name       # bit-identical points, hence identical decisions
    assert last_dev == last.value


@pytest.mark.parametrize("kernel", ["tiles", "factored"])
def test_device_engine_equals_the_host_engine_bit_for_bit(oracle_models, kernel, monkeypatch):
    """The block machines on the device (advance_blocks) against the same machines on host threads: every float of every block."""
    m = oracle_models("100vertebrates")
    cod, off = maf_blocks.block_codons(os.path.join(GOLD, "score-msa", "chr22.50alignments.maf"), m.labels, ALIASES)
    ctx = pa.load_model(os.path.join(GOLD, "models", "100vertebrates")).context()
    try:
        ctx.set_block_kernel(pa.ScoringContext.BLOCK_KERNEL_TILES if kernel == "tiles" else pa.ScoringContext.BLOCK_KERNEL_FACTORED)
        for algo, anc in ((pa.MLE, True), (pa.OMEGA, False)):
            res = {}
            for engine in ("device", "host"):
                monkeypatch.setenv("PCSF_BLOCK_ENGINE", engine)
                res[engine] = ctx.score_blocks(algo, cod, off, compute_anc=anc)
            assert res["device"][0].tobytes() == res["host"][0].tobytes(), (kernel, algo)
            assert res["device"][1].tobytes() == res["host"][1].tobytes(), (kernel, algo)
            assert np.isfinite(res["device"][0]).sum() >= 40
            if kernel == "factored":   # the device engine's prune_factored makes its own edge scales; with edge_scale_kernel's rows: the same bits
                monkeypatch.setenv("PCSF_BLOCK_ENGINE", "device")
                monkeypatch.setenv("PCSF_FACTORED_OWN_SCALES", "0")
                rows = ctx.score_blocks(algo, cod, off, compute_anc=anc)
                monkeypatch.delenv("PCSF_FACTORED_OWN_SCALES")
                assert rows[0].tobytes() == res["device"][0].tobytes() and rows[1].tobytes() == res["device"][1].tobytes(), algo
    finally:
        ctx.close()


def test_a_blocks_result_does_not_depend_on_the_list_it_is_scored_in(oracle_models, monkeypatch):
    """src/phylocsf++score_msa.hpp:115 for the library: the 50 blocks scored as one list, one by one, and in two halves give the same
    floats (the pruning kernel pinned: left to AUTO, a one-block list may pick the other kernel)."""
    m = oracle_models("100vertebrates")
    cod, off = maf_blocks.block_codons(os.path.join(GOLD, "score-msa", "chr22.50alignments.maf"), m.labels, ALIASES)
    ctx = pa.load_model(os.path.join(GOLD, "models", "100vertebrates")).context()
    try:
        ctx.set_block_kernel(ctx.choose_block_kernel(cod, off))
        for algo, anc in ((pa.MLE, True), (pa.OMEGA, False)):
            whole = ctx.score_blocks(algo, cod, off, compute_anc=anc)[0]
            half = len(off) // 2
            halves = np.concatenate([ctx.score_blocks(algo, cod, off[:half + 1], compute_anc=anc)[0], ctx.score_blocks(algo, cod, off[half:], compute_anc=anc)[0]])
            assert whole.tobytes() == halves.tobytes(), algo
            singles = np.concatenate([ctx.score_blocks(algo, cod, off[b:b + 2], compute_anc=anc)[0] for b in range(0, len(off) - 1, 7)])
            assert whole[::7].tobytes() == singles.tobytes(), algo
    finally:
        ctx.close()


TRACE_CHILD = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np
import phylocsfpp_amd as pa
d = np.load(sys.argv[1])
ctx = pa.load_model(sys.argv[2]).context()
ctx.set_block_kernel(int(sys.argv[3]))
ph = ctx.score_blocks(pa.OMEGA, d["cols"], d["off"], compute_anc=False)[0]
print("RESULT", " ".join(repr(float(x)) for x in ph))
ctx.close()
'''


def _omega_path_of_a_block_against_the_oracle(m, model_dir, cols, off, block, kernel, tmp_path, expect):
    """The OMEGA searches of one block as the HIP path ran them (the host engine prints every evaluation of PCSF_TRACE_BLOCK: kind, point,
    ln z, tree scale, the twelve settings; the device engine gives the same bits), each evaluation repeated with the ORACLE's objective
    (po_q_from_omega, po_eig, po_lpr_leaves).  How well two implementations CAN agree depends on the tree scale: the reference clamps
    noise-level entries of P(t) and rebuilds the diagonal, so columns that need an a-priori impossible double change see the eigen-solver's
    rounding, more the shorter the branches.  The yardstick (profiles/r05/eigen_forms/objective_small_scales.txt, tools/
    eigen_form_small_scale.py) is the oracle against itself with its Jacobi sweeps run in the opposite order (po_set_perturbation 32), on
    this test's 700-site block at the starting model, an H0-like and an H1-like model: up to 3e-4 relative at a tree scale of 0.001 (the
    lower end of every rho search) and 0.0025, 2e-5 at 0.0054, 6e-7 at 0.022, 3e-8 at 0.061, 6e-10 at 0.24, 1e-11 at 1 -- and the three
    forms of the device's eigen-solver sit in the same band at every scale (FORM 2: 3e-4, 3e-4, 3e-5, 4e-7, 2e-8, 3e-10, 3e-12).  Allowed
    here: 5e-11 scale^-2.9 from a tree scale of 0.05 up (1.7e-7 at 0.06, 3e-9 at 0.24, 5e-11 from 1 on: the evaluations that decide where
    a search ends), and 2e-3 below 0.05, where the band depends on the model as much as on the scale (an H1 model the search itself
    found, scale 0.0054: 2.4e-4) and the evaluations are bracketing probes 10,000 log-units below the optimum.  tests/test_gpu_blocks.py::
    test_objective_values_match_oracle holds single evaluations to 2e-8 from 0.05 up and 1e-5 at 0.02 on real blocks.  Returns the number
    of evaluations and the worst relative difference per range of the tree scale."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    npz = os.path.join(str(tmp_path), "edge_blocks.npz")
    np.savez(npz, cols=cols, off=off)
    env = dict(os.environ, PCSF_BLOCK_ENGINE="host", PCSF_TRACE_BLOCK=str(block))
    r = subprocess.run([sys.executable, "-c", TRACE_CHILD % {"root": root}, npz, model_dir, str(kernel)], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    got = [float(x) for x in [l for l in r.stdout.splitlines() if l.startswith("RESULT")][0].split()[1:]]
    assert np.float32(got[block]) == np.float32(expect)                       # the traced run is the run in question
    sub = np.ascontiguousarray(cols[:, int(off[block]):int(off[block + 1])])
    dp = C.POINTER(C.c_double)
    n, worst = 0, [0.0, 0.0, 0.0]
    for line in r.stderr.splitlines():
        if not line.startswith("TRACE omega stage"):
            continue
        t = line.split()
        lnz, scale, qs = float(t[9]), float(t[13]), np.array([float(x) for x in t[15:27]])
        if not np.isfinite(lnz):
            continue
        Q, pi = np.empty((64, 64)), np.empty(64)
        ob.lib().po_q_from_omega(qs.ctypes.data_as(dp), Q.ctypes.data_as(dp), pi.ctypes.data_as(dp))
        qd = ob.PoQdiag()
        ob.lib().po_eig(Q.ctypes.data_as(dp), pi.ctypes.data_as(dp), C.byref(qd))
        lpr, anc = C.c_double(), C.c_double()
        assert ob.lib().po_lpr_leaves(C.byref(qd), C.byref(m.m.tree), scale, sub.ctypes.data_as(C.POINTER(C.c_uint8)), sub.shape[1], sub.shape[1], 0,
                                      C.byref(lpr), C.byref(anc), None) == 0
        rel = abs(lnz - lpr.value) / abs(lpr.value)
        allowed = 5e-11 * max(1.0, scale ** -2.9) if scale >= 0.05 else 2e-3
        assert rel <= allowed, (line[:120], scale, lpr.value, rel, allowed)
        n += 1
        tier = 0 if scale >= 0.05 else 1 if scale >= 0.01 else 2
        worst[tier] = max(worst[tier], rel)
    assert n >= 100
    return n, worst


def test_engines_agree_on_empty_single_site_and_long_blocks(oracle_models, monkeypatch, tmp_path):
    """Edges of the device engine's bookkeeping: blocks without a site (MLE: lpr = 0 at every scale, the searches exhaust their 250
    restarts; OMEGA: NaN by policy), a one-site block, a block of one tile and a site, a 700-site block (44 tiles: several work-groups
    per evaluation, per-site sums over many chunks), a block whose every species shows the same codon (a flat objective: restarts run
    long and the value store is compacted) -- device engine against host engine bit for bit on both pruning kernels, and against the
    oracle's own optimiser runs inside the reference's tolerances.  An OMEGA row further from the oracle's than that must be the
    PROCEDURE's doing, not the objective's: twelve chained Brent searches, each handing on its LAST probe (not its best), end elsewhere
    when an objective value moves in its tenth digit -- then every evaluation of the block's searches, as the HIP path ran them, has to
    agree with the oracle's objective at the same parameters (_omega_path_of_a_block_against_the_oracle)."""
    from phylocsfpp_amd import synth
    m = oracle_models("53birds")
    lens = [0, 1, 17, 0, 700, 5, 40]
    cols = synth.synthetic_codon_columns(m.child1, m.child2, m.branch_len, m.S_c, m.pi_c, m.S_nc, m.pi_nc, sum(lens), seed=5, dropout=0.2, gap_rate=0.05)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    cols[:, int(off[5]):int(off[6])] = 17          # the 5-site block: one codon everywhere
    ctx = pa.load_model(os.path.join(GOLD, "models", "53birds")).context()
    lib = ob.lib()
    lib.po_run_blocks.argtypes = [C.POINTER(ob.PoModel), C.c_int, C.POINTER(C.c_uint8), C.POINTER(C.c_uint64), C.c_size_t, C.c_size_t, C.c_int, C.c_uint32,
                                  C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int]
    try:
        for kernel in (pa.ScoringContext.BLOCK_KERNEL_TILES, pa.ScoringContext.BLOCK_KERNEL_FACTORED):
            ctx.set_block_kernel(kernel)
            for algo, o_algo, anc, tol in ((pa.MLE, ob.MLE, True, 0.001), (pa.OMEGA, ob.OMEGA, False, 0.1)):
                res = {}
                for engine in ("device", "host"):
                    monkeypatch.setenv("PCSF_BLOCK_ENGINE", engine)
                    res[engine] = ctx.score_blocks(algo, cols, off, compute_anc=anc)
                assert res["device"][0].tobytes() == res["host"][0].tobytes(), (kernel, algo)
                assert res["device"][1].tobytes() == res["host"][1].tobytes(), (kernel, algo)
                oph = np.full(len(lens), np.nan, np.float32)
                oan = np.full(len(lens), np.nan, np.float32)
                lib.po_run_blocks(C.byref(m.m), o_algo, cols.ctypes.data_as(C.POINTER(C.c_uint8)), off.ctypes.data_as(C.POINTER(C.c_uint64)), len(lens), cols.shape[1],
                                  int(anc), 42, oph.ctypes.data_as(C.POINTER(C.c_float)), oan.ctypes.data_as(C.POINTER(C.c_float)), 0)
                ph = res["device"][0]
                assert np.array_equal(np.isnan(ph), np.isnan(oph)), (kernel, algo, ph, oph)
                ok = np.isfinite(oph)
                beyond = [int(b) for b in np.flatnonzero(ok) if (float(ph[b]) - float(oph[b])) ** 2 > tol]
                if algo == pa.OMEGA:
                    for b in beyond:
                        n, worst = _omega_path_of_a_block_against_the_oracle(m, os.path.join(GOLD, "models", "53birds"), cols, off, b, kernel, tmp_path, ph[b])
                        print("OMEGA, pruning kernel %d, block %d (%d sites): %.4f against the oracle's %.4f; its %d evaluations against the oracle's objective: worst relative "
                              "difference %.1e (tree scale >= 0.05), %.1e (0.01 .. 0.05), %.1e (below 0.01)" % (kernel, b, lens[b], ph[b], oph[b], n, *worst))
                    beyond = []
                assert not beyond, (kernel, algo, ph, oph)
                if algo == pa.MLE:
                    assert ph[0] == 0.0 and ph[3] == 0.0          # blocks without a site
    finally:
        ctx.close()
