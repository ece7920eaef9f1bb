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
