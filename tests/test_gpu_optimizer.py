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


def test_engines_agree_on_empty_single_site_and_long_blocks(oracle_models, monkeypatch):
    """Edges of the device engine's bookkeeping: blocks without a site (MLE: lpr = 0 at every scale, the searches exhaust their 250
    restarts; OMEGA: NaN by policy), a one-site block, a block of one tile and a site, a 700-site block (44 tiles: several work-groups
    per evaluation, per-site sums over many chunks), a block whose every species shows the same codon (a flat objective: restarts run
    long and the value store is compacted) -- device engine against host engine bit for bit on both pruning kernels, and against the
    oracle's own optimiser runs inside the reference's tolerances."""
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
                assert np.all((ph[ok] - oph[ok]) ** 2 <= tol), (kernel, algo, ph, oph)
                if algo == pa.MLE:
                    assert ph[0] == 0.0 and ph[3] == 0.0          # blocks without a site
    finally:
        ctx.close()
