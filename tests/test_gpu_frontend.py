"""GPU: the six-frame front end (device translation of nucleotide rows + run_tracks) against a plain
Python restatement of update_seqs / reverse complement (reference src/parallel_file_reader.hpp:61-113,
src/phylocsf++build_tracks.hpp:162-227) feeding the oracle.  Column indexing must be bit-exact."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

DNA = {"A": 0, "a": 0, "C": 1, "c": 1, "G": 2, "g": 2, "T": 3, "t": 3}
COMP = str.maketrans("ACGTacgt", "TGCAtgca")


def _frames_reference(rows, start_pos, chrom_len):
    """rows: list of str (one per leaf).  Returns [(skip, codons uint8 [nl, n]) for the 6 frames], scores order = reference order"""
    out = []
    L = len(rows[0])
    for strand in "+-":
        for frame in (1, 2, 3):
            if strand == "+":
                skip = (frame - start_pos) % 3
            else:
                skip = (frame - (chrom_len - (start_pos + L) + 2)) % 3
            skip = min(skip, L)
            n = (L - skip) // 3
            cod = np.full((len(rows), n), 64, np.uint8)
            for l, r in enumerate(rows):
                for c in range(n):
                    t = r[skip + 3 * c: skip + 3 * c + 3]
                    if all(ch in DNA for ch in t):
                        cod[l, c] = 16 * DNA[t[0]] + 4 * DNA[t[1]] + DNA[t[2]]
            out.append((skip, cod))
        rows = [r[::-1].translate(COMP) for r in rows]
    return out


@pytest.mark.parametrize("model", ["53birds", "58mammals"])
def test_six_frames_match_host_translation(model, oracle_models, gpu_contexts):
    m = oracle_models(model)
    ctx = gpu_contexts(model)
    rng = np.random.default_rng(77)
    alphabet = np.frombuffer(b"ACGTacgtNn-", dtype=np.uint8)
    prob = np.array([.2, .2, .2, .2, .03, .03, .03, .03, .02, .01, .05])
    alns, cols = [], 0
    lens = [1, 2, 3, 4, 5, 17, 64, 191, 192, 193, 1000]
    for i, L in enumerate(lens):
        alns.append((cols, L, int(rng.integers(1, 10**6)), int(rng.integers(2 * 10**6, 3 * 10**6))))
        cols += L + int(rng.integers(0, 5))           # unused padding columns between alignments
    nuc = rng.choice(alphabet, size=(m.nl, cols), p=prob / prob.sum())
    res, skip = ctx.score_alignments_6frames(nuc, alns)
    for k, (c0, L, sp, cl) in enumerate(alns):
        rows = [bytes(nuc[l, c0:c0 + L]).decode() for l in range(m.nl)]
        ref = _frames_reference(rows, sp, cl)
        for f in range(6):
            rskip, cod = ref[f]
            assert skip[k, f] == rskip                                   # bit-exact frame assignment
            assert len(res[k][f]) == cod.shape[1]
            if cod.shape[1] == 0:
                continue
            want = m.run_tracks(cod)
            if f >= 3:
                want = want[::-1]                                        # the reference reverses '-' strand scores
            np.testing.assert_allclose(res[k][f], want, rtol=1e-9, atol=1e-9)


# ---- branch-length score on the device (reference src/additional_scores.hpp:5-84) --------------------------
def _random_rows(rng, nl, n_cols):
    """rows with realistic structure: runs of absent species (N / -), lower case, and fully random columns"""
    alphabet = np.frombuffer(b"ACGTacgtNn-.", dtype=np.uint8)
    prob = np.array([.2, .2, .2, .2, .03, .03, .03, .03, .02, .01, .04, .01])
    nuc = rng.choice(alphabet, size=(nl, n_cols), p=prob / prob.sum())
    for l in range(nl):                                     # species missing over long stretches, as in real MAFs
        for _ in range(int(rng.integers(0, 4))):
            a = int(rng.integers(0, n_cols))
            b = min(n_cols, a + int(rng.integers(1, n_cols)))
            nuc[l, a:b] = ord("N") if rng.random() < 0.5 else ord("-")
    nuc[:, :3] = ord("-")                                   # empty set
    nuc[0, 1] = ord("A")                                    # one species only -> 0
    nuc[[0, nl - 1], 2] = ord("c")                          # first and last leaf: the set's ancestor is the root
    nuc[:, 3] = ord("G")                                    # all species -> exactly 1
    return nuc


@pytest.mark.parametrize("model", ["53birds", "58mammals", "100vertebrates"])
def test_branch_length_scores_identical_to_the_recursion(model, oracle_models, gpu_contexts):
    """bit-identical doubles: the float threshold of build-tracks is decided from them"""
    m = oracle_models(model)
    ctx = gpu_contexts(model)
    ctx.set_bls_branch_lengths(m.branch_len_d)
    rng = np.random.default_rng(5)
    for n_cols in (1, 7, 255, 256, 257, 5000):
        nuc = _random_rows(rng, m.nl, max(n_cols, 4))[:, :n_cols] if n_cols < 4 else _random_rows(rng, m.nl, n_cols)
        got = ctx.branch_length_scores(nuc)
        _, want = m.bls(nuc)
        assert got.tobytes() == want.tobytes(), (model, n_cols, np.abs(got - want).max())
    assert got[3] == 1.0 and got[0] == 0.0 and got[1] == 0.0 and 0.0 < got[2] < 1.0


def test_branch_length_scores_with_the_six_frames(oracle_models, gpu_contexts):
    """one call: frames + per-base scores from the same uploaded rows; padding columns between alignments stay 0
    and are not checked for characters"""
    m = oracle_models("53birds")
    ctx = gpu_contexts("53birds")
    ctx.set_bls_branch_lengths(m.branch_len_d)
    rng = np.random.default_rng(9)
    alns, cols = [], 0
    for L in (1, 2, 3, 50, 700):
        alns.append((cols, L, int(rng.integers(1, 10**6)), 3 * 10**6))
        cols += L + 3
    nuc = _random_rows(rng, m.nl, cols)
    for c0, L, _, _ in alns:
        nuc[:, c0 + L:c0 + L + 3] = ord("#")                # not part of any alignment
    res, skip, bls = ctx.score_alignments_6frames(nuc, alns, with_bls=True)
    res2, skip2 = ctx.score_alignments_6frames(nuc, alns)
    for k in range(len(alns)):
        for f in range(6):
            assert np.array_equal(res[k][f], res2[k][f])
    for c0, L, _, _ in alns:
        _, want = m.bls(nuc[:, c0:c0 + L])
        assert bls[c0:c0 + L].tobytes() == want.tobytes()
        assert (bls[c0 + L:c0 + L + 3] == 0).all()
    # alignments too short for a codon still get their scores
    two = _random_rows(rng, m.nl, 50)[:, 10:12].copy()
    _, _, b2 = ctx.score_alignments_6frames(two, [(0, 2, 5, 1000)], with_bls=True)
    assert b2.tobytes() == m.bls(two)[1].tobytes()


def test_branch_length_scores_reject_other_characters(oracle_models, gpu_contexts):
    m = oracle_models("53birds")
    ctx = gpu_contexts("53birds")
    ctx.set_bls_branch_lengths(m.branch_len_d)
    nuc = np.full((m.nl, 300), ord("A"), np.uint8)
    nuc[7, 123] = ord("X")
    with pytest.raises(RuntimeError, match="unexpected character 'X'"):
        ctx.branch_length_scores(nuc)
    with pytest.raises(RuntimeError, match="unexpected character 'X'"):
        ctx.score_alignments_6frames(nuc, [(0, 300, 1, 1000)], with_bls=True)
    assert (ctx.branch_length_scores(np.full((m.nl, 10), ord("t"), np.uint8)) == 1.0).all()   # the context stays usable


def test_fused_front_end_equals_the_separate_kernel(golden_dir):
    """The pruning kernel forms its codon columns from the nucleotide rows itself (kernels.hip, FUSED); PCSF_FUSED_FRONTEND=0
    keeps extract_frames_kernel + the codon matrix.  Same kernel arithmetic on the same codons: the scores must be
    bit-identical, for ragged alignments whose frame segments straddle work-group windows."""
    import os
    import phylocsfpp_amd as pa
    m = pa.load_model(os.path.join(golden_dir, "models", "58mammals"))
    rng = np.random.default_rng(123)
    alphabet = np.frombuffer(b"ACGTacgtNn-.", dtype=np.uint8)
    prob = np.array([.2, .2, .2, .2, .03, .03, .03, .03, .02, .01, .04, .01])
    alns, cols = [], 0
    for L in [1, 2, 3, 5, 64, 191, 193, 3 * 64, 3 * 64 + 1, 1000, 4097, 20000] + [int(x) for x in rng.integers(1, 700, 40)]:
        alns.append((cols, L, int(rng.integers(1, 10**6)), int(rng.integers(2 * 10**6, 3 * 10**6))))
        cols += L + int(rng.integers(0, 4))
    nuc = rng.choice(alphabet, size=(m.nl, cols), p=prob / prob.sum())
    res = {}
    old = os.environ.get("PCSF_FUSED_FRONTEND")
    try:
        for mode in ("1", "0"):
            os.environ["PCSF_FUSED_FRONTEND"] = mode
            ctx = m.context(device=0, with_bls=True)
            res[mode] = (ctx.score_alignments_6frames(nuc, alns), ctx.score_alignments_6frames(nuc, alns, with_bls=True))
            ctx.close()
    finally:
        if old is None:
            os.environ.pop("PCSF_FUSED_FRONTEND", None)
        else:
            os.environ["PCSF_FUSED_FRONTEND"] = old
    for variant in (0, 1):
        a, b = res["1"][variant], res["0"][variant]
        assert np.array_equal(a[1], b[1])                                  # skip_bases
        for k in range(len(alns)):
            for f in range(6):
                assert a[0][k][f].tobytes() == b[0][k][f].tobytes(), (variant, k, f)
    assert res["1"][1][2].tobytes() == res["0"][1][2].tobytes()            # per-base branch-length scores


def test_six_frame_call_without_branch_length_scores_rejects_other_characters(oracle_models, gpu_contexts):
    """both six-frame entry points reject what the reference exits on (src/translation.hpp:29-88), not only the one that
    also computes the branch-length scores"""
    m = oracle_models("53birds")
    ctx = gpu_contexts("53birds")
    nuc = np.full((m.nl, 300), ord("A"), np.uint8)
    nuc[7, 123] = ord("X")
    with pytest.raises(RuntimeError, match="unexpected character 'X'"):
        ctx.score_alignments_6frames(nuc, [(0, 300, 1, 1000)])
    nuc[7, 123] = ord("n")
    res, _ = ctx.score_alignments_6frames(nuc, [(0, 300, 1, 1000)])        # the context stays usable
    assert len(res[0][0]) == 100


def test_repeated_six_frame_calls_are_bit_identical(golden_dir):
    """Regression: the pruning kernel double-buffers its branch matrices in LDS with LDS-DMA.  A work-group barrier waits
    for LDS operations only, not for vector memory, so the DMA pieces of the current slot need an explicit vmcnt(0) before
    the barrier publishes them; without it a product behind a POP op could read pieces of the matrix two slots back when
    memory latency exceeded one product (seen with the fused front end: one work-group in ~30,000 with 64 slightly wrong
    scores, different ones on every call).  Forty calls on the same 600,000 codon-frames must agree bit for bit with each
    other and with the separate-kernel path."""
    import os
    import phylocsfpp_amd as pa
    m = pa.load_model(os.path.join(golden_dir, "models", "58mammals"))
    rng = np.random.default_rng(2)
    L = 300_000
    nuc = np.frombuffer(b"ACGTacgtN-", dtype=np.uint8)[rng.integers(0, 10, size=(m.nl, L))]
    alns, c = [], 0
    while c < L:
        k = int(min(rng.integers(50, 20000), L - c))
        alns.append((c, k, int(rng.integers(1, 10**6)), 3 * 10**6))
        c += k
    old = os.environ.get("PCSF_FUSED_FRONTEND")
    ref = None
    try:
        for fused, calls in (("0", 3), ("1", 40)):
            os.environ["PCSF_FUSED_FRONTEND"] = fused
            ctx = m.context(device=0, with_bls=True)
            for r in range(calls):
                res = ctx.score_alignments_6frames(nuc, alns, with_bls=(r % 2 == 0))
                flat = np.concatenate([res[0][k][f] for k in range(len(alns)) for f in range(6)])
                if ref is None:
                    ref = flat
                assert flat.tobytes() == ref.tobytes(), (fused, r, int((flat != ref).sum()))
            ctx.close()
    finally:
        if old is None:
            os.environ.pop("PCSF_FUSED_FRONTEND", None)
        else:
            os.environ["PCSF_FUSED_FRONTEND"] = old
