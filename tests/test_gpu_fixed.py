"""GPU parity of the FIXED path (run_tracks, src/run.hpp:35-55) against the CPU oracle,
through the C-ABI.  Tolerance: north_star asks for <= 1e-6 relative on the LLR scores; the
kernel and the oracle differ only in fp64 summation order, so we hold it to 1e-9."""
import os

import numpy as np
import pytest

from phylocsfpp_amd import synth

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-9, 1e-9


def _columns(m, n_sites, seed):
    return synth.synthetic_codon_columns(m.child1, m.child2, m.branch_len, m.S_c, m.pi_c, m.S_nc, m.pi_nc,
                                         n_sites, seed=seed)


@pytest.mark.parametrize("model", ["53birds", "58mammals", "100vertebrates"])
@pytest.mark.parametrize("n_sites", [1, 15, 16, 17, 63, 64, 65, 1000])
def test_fixed_matches_oracle(model, n_sites, oracle_models, gpu_contexts):
    m = oracle_models(model)
    ctx = gpu_contexts(model)
    cod = _columns(m, n_sites, seed=1000 + n_sites)
    got = ctx.score_sites_fixed(cod)
    want = m.run_tracks(cod)
    assert got.shape == want.shape
    assert np.all(np.isfinite(got))
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=ATOL)


def test_branch_matrices_match_oracle(oracle_models, gpu_contexts):
    m = oracle_models("58mammals")
    ctx = gpu_contexts("58mammals")
    P = ctx.branch_matrices()
    for k, coding in enumerate((True, False)):
        rc, Po = m.all_P(m.qdiag(coding), 1.0)
        assert rc == 0
        np.testing.assert_allclose(P[k], Po, rtol=0, atol=1e-13)


def test_all_gap_and_extreme_columns(oracle_models, gpu_contexts):
    m = oracle_models("58mammals")
    ctx = gpu_contexts("58mammals")
    cod = np.full((m.nl, 40), 64, dtype=np.uint8)          # all species missing: z = 1, score 0
    cod[:, 10:20] = 0                                      # every species AAA
    cod[0, 20:30] = 48                                     # a lone stop codon in the reference
    cod[::2, 30:40] = 63
    got = ctx.score_sites_fixed(cod)
    want = m.run_tracks(cod)
    np.testing.assert_allclose(got, want, rtol=RTOL, atol=ATOL)
    assert abs(got[0]) < 1e-9


def test_strided_input_and_empty(oracle_models, gpu_contexts):
    m = oracle_models("53birds")
    ctx = gpu_contexts("53birds")
    big = _columns(m, 300, seed=7)
    view = big[:, 50:250]                                  # leaf_stride (300) > n_sites (200)
    np.testing.assert_allclose(ctx.score_sites_fixed(view), m.run_tracks(np.ascontiguousarray(view)), rtol=RTOL, atol=ATOL)
    assert ctx.score_sites_fixed(np.zeros((m.nl, 0), dtype=np.uint8)).shape == (0,)


def test_device_pointer_entry_point_and_lnz(oracle_models, gpu_contexts):
    import torch
    m = oracle_models("58mammals")
    ctx = gpu_contexts("58mammals")
    cod = _columns(m, 5000, seed=11)
    d = torch.from_numpy(cod).cuda()
    lnz = torch.empty((2, cod.shape[1]), dtype=torch.float64, device="cuda")
    out = ctx.score_sites_fixed_torch(d, lnz=lnz)
    torch.cuda.synchronize()
    want = m.run_tracks(cod)
    np.testing.assert_allclose(out.cpu().numpy(), want, rtol=RTOL, atol=ATOL)
    l = lnz.cpu().numpy()
    np.testing.assert_allclose(10.0 * (l[0] - l[1]) / np.log(10.0), want, rtol=RTOL, atol=ATOL)


def test_linearity_property_large(oracle_models, gpu_contexts):
    """Size-independent property at a BASELINE-sized batch: every site is independent, so the
    score vector of a permuted batch is the permutation of the score vector."""
    m = oracle_models("58mammals")
    ctx = gpu_contexts("58mammals")
    base = _columns(m, 4096, seed=3)
    reps = 64
    cod = np.tile(base, (1, reps))
    rng = np.random.default_rng(5)
    perm = rng.permutation(cod.shape[1])
    a = ctx.score_sites_fixed(cod)
    b = ctx.score_sites_fixed(np.ascontiguousarray(cod[:, perm]))
    assert np.array_equal(a[perm], b)
    assert np.array_equal(a[:4096], a[4096:8192])
    np.testing.assert_allclose(a[:4096], m.run_tracks(base), rtol=RTOL, atol=ATOL)


@pytest.mark.parametrize("model", ["58mammals", "100vertebrates"])
def test_all_gap_subtrees_are_skipped_without_changing_a_score(oracle_models, model, monkeypatch):
    """Tracks mode drops, per work-group of 64 columns, every product above a subtree that is gaps in all 64 columns (kernels.hip SKIP;
    the message of such a subtree does not depend on the column, the context records it once with the full program and the kernel
    substitutes it; reference: src/instance.hpp:625-639, src/fixed_lik.hpp:111-117).  Windows with no species at all, one species, two
    species of different clades, one clade, a random third, everything -- plus ordinary gaps -- against the oracle, and against the
    same library with the skipping switched off (PCSF_SKIP_ALLGAP=0): bit for bit the same scores."""
    import phylocsfpp_amd as pa
    m = oracle_models(model)
    rng = np.random.default_rng(77)
    n_win = 48
    cod = _columns(m, 64 * n_win, seed=31)
    nl = m.nl
    # clades: the leaves below a random internal node
    c1, c2 = np.asarray(m.child1), np.asarray(m.child2)
    def leaves_below(v):
        return [v] if v < nl else leaves_below(int(c1[v])) + leaves_below(int(c2[v]))
    for w in range(n_win):
        kind = w % 8
        if kind == 0:
            keep = []
        elif kind == 1:
            keep = [int(rng.integers(nl))]
        elif kind == 2:
            keep = [0, nl - 1]
        elif kind == 3:
            keep = leaves_below(int(rng.integers(nl, 2 * nl - 2)))
        elif kind == 4:
            keep = list(np.flatnonzero(rng.random(nl) < 0.33))
        elif kind == 5:
            keep = leaves_below(int(rng.integers(nl, 2 * nl - 2))) + [int(rng.integers(nl))]
        elif kind == 6:
            keep = list(range(nl))
        else:
            keep = list(np.flatnonzero(rng.random(nl) < 0.08))
        drop = np.ones(nl, bool)
        drop[keep] = False
        cod[drop, 64 * w:64 * (w + 1)] = 64
    cod[rng.random(cod.shape) < 0.03] = 64
    want = m.run_tracks(cod)
    prefix = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "models", model)
    got = {}
    for flag in ("1", "0"):
        monkeypatch.setenv("PCSF_SKIP_ALLGAP", flag)
        ctx = pa.load_model(prefix).context(device=0)
        got[flag] = ctx.score_sites_fixed(cod)
        # a window boundary that is not a work-group boundary: the same columns shifted by 17
        shifted = ctx.score_sites_fixed(np.ascontiguousarray(cod[:, 17:]))
        np.testing.assert_allclose(shifted, want[17:], rtol=RTOL, atol=ATOL)
        ctx.close()
        np.testing.assert_allclose(got[flag], want, rtol=RTOL, atol=ATOL)
    assert np.array_equal(got["1"], got["0"])


def test_a_context_set_up_beside_a_busy_one_scores_like_it(golden_dir):
    """Regression (round 4; found by tests/test_gpu_full_size.py with four contexts on one GPU): pcsf_create filled the all-gap columns
    and zeroed the table of all-gap messages with hipMemset -- a fill kernel on the NULL stream that returns before it has run -- and
    recorded the messages with a kernel on the context's non-blocking stream, which does not wait for the null stream.  On an idle device
    the fills ran first; on a device busy with another context's scoring the kernel could overtake them and the new context then scored
    every work-group that skips an all-gap subtree wrongly (-inf, or finite and wrong).  The end-to-end test reproduced it in four of nine
    runs with the old library and in none of thirteen with the fix, same boxes.  Here the same situation through the binding: contexts
    are created, four at a time on threads of their own, while two threads keep the device busy; each must score a drop-out-heavy batch
    exactly like the first context.  (This form has not reproduced the failure by itself: it guards the property, the end-to-end test
    the bug.)"""
    import threading
    import phylocsfpp_amd as pa
    m = pa.load_model(os.path.join(golden_dir, "models", "58mammals"))
    from phylocsfpp_amd import synth
    cod = synth.synthetic_codon_columns(m.child1, m.child2, m.branch_len, m.S_c, m.pi_c, m.S_nc, m.pi_nc, 64 * 700, seed=3, dropout=0.6, gap_rate=0.05)
    busy = np.tile(synth.synthetic_codon_columns(m.child1, m.child2, m.branch_len, m.S_c, m.pi_c, m.S_nc, m.pi_nc, 100_000, seed=4, dropout=0.3), (1, 12))   # 1.2 M sites: ~16 ms of kernel per call
    first = m.context(device=0)
    want = first.score_sites_fixed(cod)
    assert np.isfinite(want).all()
    second = m.context(device=0)
    stop = threading.Event()

    def keep_busy(ctx):                    # (the C call releases the interpreter lock)
        while not stop.is_set():
            ctx.score_sites_fixed(busy)
    threads = [threading.Thread(target=keep_busy, args=(c,)) for c in (first, second)]
    for t in threads:
        t.start()
    others, bad = [], []
    lock = threading.Lock()

    def make_and_score(k):                 # like build-tracks --devices: several contexts set up at once, on threads of their own
        c = m.context(device=0)            # its set-up work queues up behind the busy contexts' (and the other new contexts')
        got = c.score_sites_fixed(cod)
        with lock:
            others.append(c)
            if not np.array_equal(got, want):
                bad.append((k, int((got != want).sum()), int((~np.isfinite(got)).sum())))
    try:
        for rnd in range(3):
            makers = [threading.Thread(target=make_and_score, args=(4 * rnd + k,)) for k in range(4)]
            for t in makers:
                t.start()
            for t in makers:
                t.join()
    finally:
        stop.set()
        for t in threads:
            t.join()
    second.close()
    assert not bad, bad
    for c in others:
        assert np.array_equal(c.score_sites_fixed(cod), want)
        c.close()
    first.close()


def test_bench_sized_batch_of_two_million_sites(oracle_models, gpu_contexts):
    """BASELINE.json configs[2]'s batch as bench.py scores it (2,000,000 codon-frames in one call, 31,250 work-groups): the oracle on a
    strided sample of 64,516 sites, and for ALL sites the size-independent property that a site's score depends on its column alone --
    the batch draws its columns from a pool, so every score must equal, bit for bit, the score of its pool column in a small call."""
    m = oracle_models("58mammals")
    ctx = gpu_contexts("58mammals")
    pool = _columns(m, 1 << 15, seed=21)
    rng = np.random.default_rng(22)
    idx = rng.integers(0, pool.shape[1], 2_000_000)
    cod = np.ascontiguousarray(pool[:, idx])
    got = ctx.score_sites_fixed(cod)
    assert got.shape == (2_000_000,) and np.isfinite(got).all()
    sample = np.arange(0, 2_000_000, 31)
    np.testing.assert_allclose(got[sample], m.run_tracks(np.ascontiguousarray(cod[:, sample])), rtol=RTOL, atol=ATOL)
    assert np.array_equal(got, ctx.score_sites_fixed(pool)[idx])
    perm = rng.permutation(2_000_000)
    assert np.array_equal(ctx.score_sites_fixed(np.ascontiguousarray(cod[:, perm])), got[perm])


def test_tree_deeper_than_the_register_stack(tmp_path, golden_dir):
    """A perfectly balanced 64-leaf tree needs 5 parked tiles; the kernels keep 3 in registers (enough for every tree the
    reference ships) and run the DEEP variants -- stack in scratch memory -- for anything deeper, where
    newick_parse / newick_flatten (src/newick.hpp:31-229) accept any binary tree.  Tracks mode, the fused six-frame front
    end, block mode (explicit and factored) against the oracle."""
    import shutil
    import oracle_binding as ob
    import phylocsfpp_amd as pa
    from phylocsfpp_amd import synth
    rng = np.random.default_rng(64)

    def newick(lo, hi):
        if hi - lo == 1:
            return "s%02d:%.4f" % (lo, rng.uniform(0.01, 0.3))
        mid = (lo + hi) // 2
        return "(%s,%s):%.4f" % (newick(lo, mid), newick(mid, hi), rng.uniform(0.01, 0.2))
    prefix = str(tmp_path / "deep")
    text = newick(0, 64)
    open(prefix + ".nh", "w").write(text[:text.rindex(":")] + ";\n")
    for kind in ("coding", "noncoding"):
        shutil.copyfile(os.path.join(golden_dir, "models", "58mammals_%s.ECM" % kind), "%s_%s.ECM" % (prefix, kind))
    pm, om = pa.load_model(prefix, aliases=""), ob.OracleModel(prefix)
    assert pm.nl == 64 and pa.build_schedule(pm.child1, pm.child2)[2] == 5
    ctx = pm.context(device=0)
    cod = synth.synthetic_codon_columns(pm.child1, pm.child2, pm.branch_len, pm.S_c, pm.pi_c, pm.S_nc, pm.pi_nc, 333, seed=8, dropout=0.2, gap_rate=0.05)
    np.testing.assert_allclose(ctx.score_sites_fixed(cod), om.run_tracks(cod), rtol=1e-9, atol=1e-9)
    # six frames from nucleotide rows (fused front end, DEEP)
    lut = np.frombuffer(b"ACGT", dtype=np.uint8)
    nuc = lut[rng.integers(0, 4, size=(64, 700))]
    nuc[rng.random(nuc.shape) < 0.05] = ord("-")
    res, skip = ctx.score_alignments_6frames(nuc, [(0, 700, 17, 5000)])
    codes = np.full(256, 4, np.int64)
    for i, ch in enumerate(b"ACGT"):
        codes[ch] = i
    ids = codes[nuc[:, int(skip[0, 0]):]]
    n = ids.shape[1] // 3
    tri = ids[:, :3 * n].reshape(64, n, 3)
    frame1 = np.where((tri == 4).any(axis=2), 64, 16 * tri[:, :, 0] + 4 * tri[:, :, 1] + tri[:, :, 2]).astype(np.uint8)
    np.testing.assert_allclose(res[0][0], om.run_tracks(np.ascontiguousarray(frame1)), rtol=1e-9, atol=1e-9)
    # block mode: explicit matrices for the long block, factored pruning for the short ones
    off = np.array([0, 5, 22, 333], dtype=np.uint64)
    got, failed = ctx.block_loglik(0, cod, off, 0.8)
    qd = om.qdiag(True)
    import ctypes as C
    for b in range(3):
        sub = np.ascontiguousarray(cod[:, int(off[b]):int(off[b + 1])])
        lpr, anc = C.c_double(), C.c_double()
        assert ob.lib().po_lpr_leaves(C.byref(qd), C.byref(om.m.tree), 0.8, sub.ctypes.data_as(C.POINTER(C.c_uint8)), sub.shape[1], sub.shape[1], 0,
                                      C.byref(lpr), C.byref(anc), None) == 0
        assert not failed[b] and abs(got[b] - lpr.value) <= 1e-9 * abs(lpr.value)
    ctx.close()


def test_tree_of_more_than_128_leaves(tmp_path, golden_dir):
    """Species sets are two 64-bit words in the kernels that skip all-gap subtrees (schedule.hpp program_facts, kernels.hip SKIP,
    tiles.hip, factored.hip); a tree of 150 leaves has no such facts and must take the plain paths: tracks mode, six frames, and block
    mode (prune_tiles walks every op, leaf presence from LDS) against the oracle."""
    import shutil
    import ctypes as C
    import oracle_binding as ob
    import phylocsfpp_amd as pa
    rng = np.random.default_rng(150)

    def newick(lo, hi):   # random binary tree over leaves lo..hi-1
        if hi - lo == 1:
            return "s%03d:%.4f" % (lo, rng.uniform(0.005, 0.08))
        mid = int(rng.integers(lo + 1, hi))
        return "(%s,%s):%.4f" % (newick(lo, mid), newick(mid, hi), rng.uniform(0.005, 0.05))
    prefix = str(tmp_path / "big")
    text = newick(0, 150)
    open(prefix + ".nh", "w").write(text[:text.rindex(":")] + ";\n")
    for kind in ("coding", "noncoding"):
        shutil.copyfile(os.path.join(golden_dir, "models", "58mammals_%s.ECM" % kind), "%s_%s.ECM" % (prefix, kind))
    pm, om = pa.load_model(prefix, aliases=""), ob.OracleModel(prefix)
    assert pm.nl == 150
    ctx = pm.context(device=0)
    cod = synth.synthetic_codon_columns(pm.child1, pm.child2, pm.branch_len, pm.S_c, pm.pi_c, pm.S_nc, pm.pi_nc, 400, seed=8, dropout=0.3, gap_rate=0.05)
    cod[20:140, 64:192] = 64            # whole windows in which most species are gaps
    np.testing.assert_allclose(ctx.score_sites_fixed(cod), om.run_tracks(cod), rtol=1e-9, atol=1e-9)
    off = np.array([0, 7, 40, 400], dtype=np.uint64)
    got, failed = ctx.block_loglik(0, cod, off, 1.3)
    qd = om.qdiag(True)
    for b in range(3):
        sub = np.ascontiguousarray(cod[:, int(off[b]):int(off[b + 1])])
        lpr, anc = C.c_double(), C.c_double()
        assert ob.lib().po_lpr_leaves(C.byref(qd), C.byref(om.m.tree), 1.3, sub.ctypes.data_as(C.POINTER(C.c_uint8)), sub.shape[1], sub.shape[1], 0,
                                      C.byref(lpr), C.byref(anc), None) == 0
        assert not failed[b] and abs(got[b] - lpr.value) <= 1e-9 * abs(lpr.value), (b, got[b], lpr.value)
    ph, an = ctx.score_blocks(pa.MLE, cod, off, compute_anc=True)
    assert np.isfinite(ph).all() and np.isfinite(an).all()
    ctx.close()


OTHER_MODELS = ["20flies", "26worms", "12flies", "23flies", "49birds", "7yeast", "21mosquitoes", "29mammals"]


@pytest.mark.parametrize("model", OTHER_MODELS)
def test_the_references_other_built_in_models_score_like_the_oracle(model):
    """The eight other models the reference compiles in (src/models.hpp:1443-1455; shipped as files, tests/golden/make_models.py): trees of
    7 to 49 species, rate tables of their own (flies, worms, yeast, mosquitoes).  Per-codon FIXED scores against the oracle at the same
    1e-9, six frames + branch-length scores through the fused front end, and the block strategies FIXED / MLE against the oracle's."""
    import oracle_binding as ob
    import phylocsfpp_amd as pa
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    m = ob.OracleModel(os.path.join(root, "phylocsfpp_amd", "models", model))
    ctx = pa.load_model(model).context(device=0, with_bls=True)
    try:
        cod = _columns(m, 333, seed=len(model) * 7)
        cod[:, 5] = 64
        cod[::3, 40:60] = 64
        got = ctx.score_sites_fixed(cod)
        want = m.run_tracks(cod)
        np.testing.assert_allclose(got, want, rtol=RTOL, atol=ATOL)
        # six frames of two alignments from nucleotide rows, with the per-base branch-length scores
        rng = np.random.default_rng(5)
        nuc = np.frombuffer(b"ACGTacgtN-", dtype=np.uint8)[rng.integers(0, 10, size=(m.nl, 400))]
        alns = [(0, 250, 1001, 5000), (250, 150, 77, 9000)]
        from test_gpu_frontend import _frames_reference
        res, skip, bls = ctx.score_alignments_6frames(nuc, alns, with_bls=True)
        for k, (c0, L, sp, cl) in enumerate(alns):
            _, want_bls = m.bls(np.ascontiguousarray(nuc[:, c0:c0 + L]))
            assert np.array_equal(bls[c0:c0 + L], want_bls)
            ref = _frames_reference([bytes(nuc[l, c0:c0 + L]).decode() for l in range(m.nl)], sp, cl)
            for f in range(6):
                rskip, fc = ref[f]
                assert skip[k, f] == rskip and len(res[k][f]) == fc.shape[1]
                want6 = m.run_tracks(fc)
                np.testing.assert_allclose(res[k][f], want6[::-1] if f >= 3 else want6, rtol=RTOL, atol=ATOL)
        # whole blocks: FIXED and MLE
        off = np.array([0, 40, 41, 140, 333], dtype=np.uint64)
        for algo, tol in ((1, 1e-5), (0, 0.0316)):
            ph, an = ctx.score_blocks(algo, cod, off, compute_anc=True)
            wph, wan = m.run_blocks(algo, cod, off, compute_anc=True)
            assert np.array_equal(np.isnan(ph), np.isnan(wph))
            ok = ~np.isnan(wph)
            assert np.all(np.abs(ph[ok] - wph[ok]) <= tol + 1e-6 * np.abs(wph[ok])), (model, algo, ph, wph)
    finally:
        ctx.close()
