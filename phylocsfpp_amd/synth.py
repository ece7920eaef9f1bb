"""Seeded synthetic alignment columns for tests and bench.py (numpy only).

Codon columns are evolved down the model's own tree: the root state is drawn from the codon
frequencies and every child state from P(t_branch)[parent state, :], with P = expm(Q t) of the
ECM rate matrix -- alternating coding / non-coding segments -- and then masked like real
multiz blocks (per-block species drop-out, gap runs), following SURVEY.md section 8(d) config 3.
The generator is independent of both the HIP path and the oracle (it uses scipy's expm).
"""
import numpy as np


def ecm_rate_matrix(S, pi):
    Q = np.array(S, dtype=np.float64) * np.asarray(pi, dtype=np.float64)[None, :]
    np.fill_diagonal(Q, 0.0)
    np.fill_diagonal(Q, -Q.sum(axis=1))
    return Q / float(-(np.asarray(pi) * np.diag(Q)).sum())


def _branch_matrices(Q, branch_len):
    from scipy.linalg import expm
    out = []
    for t in branch_len:
        P = np.clip(expm(Q * float(t)), 0.0, None)
        out.append(P / P.sum(axis=1, keepdims=True))
    return out


def evolve_codons(child1, child2, branch_len, S, pi, n_sites, rng):
    """uint8 [n_leaves, n_sites] of codon ids 0..63 evolved under one ECM."""
    n = len(child1)
    nl = (n + 1) // 2
    Q = ecm_rate_matrix(S, pi)
    P = _branch_matrices(Q, np.asarray(branch_len, dtype=np.float64)[:n - 1])
    p0 = np.asarray(pi, dtype=np.float64)
    p0 = p0 / p0.sum()
    state = [None] * n
    state[n - 1] = rng.choice(64, size=n_sites, p=p0).astype(np.uint8)
    for node in range(n - 1, nl - 1, -1):  # parents have larger ids than their children
        for ch in (int(child1[node]), int(child2[node])):
            par = state[node]
            out = np.empty(n_sites, dtype=np.uint8)
            order = np.argsort(par, kind="stable")
            counts = np.bincount(par, minlength=64)
            pos = 0
            for a in range(64):
                c = int(counts[a])
                if c:
                    out[order[pos:pos + c]] = rng.choice(64, size=c, p=P[ch][a])
                    pos += c
            state[ch] = out
        state[node] = None
    return np.stack(state[:nl])


def synthetic_codon_columns(child1, child2, branch_len, S_c, pi_c, S_nc, pi_nc, n_sites, seed=20261002,
                            coding_fraction=0.10, segment_codons=120, block_codons=67, dropout=0.3,
                            gap_rate=0.02, gap_mean_len=2.0, keep_leaves=8, return_segments=False):
    """Leaf-major uint8 [n_leaves, n_sites] codon matrix (0..63, 64 = gap/N) that looks like the
    frame-1 translation of a multiz alignment: coding and non-coding segments alternate, every
    ~block_codons columns each species other than the first `keep_leaves` is missing with
    probability `dropout`, and gap runs hit single species.  return_segments: also return the boolean
    vector "segment s (segment_codons columns each) was evolved under the coding model"."""
    rng = np.random.default_rng(seed)
    n = len(child1)
    nl = (n + 1) // 2
    n_seg = max(1, (n_sites + segment_codons - 1) // segment_codons)
    is_coding = rng.random(n_seg) < coding_fraction
    n_cod_sites = int(is_coding.sum()) * segment_codons
    n_nc_sites = n_seg * segment_codons - n_cod_sites
    cod = evolve_codons(child1, child2, branch_len, S_c, pi_c, max(n_cod_sites, 1), rng)
    ncd = evolve_codons(child1, child2, branch_len, S_nc, pi_nc, max(n_nc_sites, 1), rng)
    out = np.empty((nl, n_seg * segment_codons), dtype=np.uint8)
    ic = inc = 0
    for s in range(n_seg):
        dst = slice(s * segment_codons, (s + 1) * segment_codons)
        if is_coding[s]:
            out[:, dst] = cod[:, ic:ic + segment_codons]
            ic += segment_codons
        else:
            out[:, dst] = ncd[:, inc:inc + segment_codons]
            inc += segment_codons
    out = np.ascontiguousarray(out[:, :n_sites])
    # per-block species drop-out
    n_blk = (n_sites + block_codons - 1) // block_codons
    drop = rng.random((nl, n_blk)) < dropout
    drop[:min(keep_leaves, nl)] = False
    mask = np.repeat(drop, block_codons, axis=1)[:, :n_sites]
    out[mask] = 64
    # gap runs
    starts = rng.random((nl, n_sites)) < gap_rate
    starts[0] = False
    li, si = np.nonzero(starts)
    lens = rng.geometric(1.0 / gap_mean_len, size=len(li))
    for l, s, k in zip(li, si, lens):
        out[l, s:s + k] = 64
    if return_segments:
        return out, is_coding
    return out
