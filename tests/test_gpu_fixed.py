"""GPU parity of the FIXED path (run_tracks, src/run.hpp:35-55) against the CPU oracle,
through the C-ABI.  Tolerance: north_star asks for <= 1e-6 relative on the LLR scores; the
kernel and the oracle differ only in fp64 summation order, so we hold it to 1e-9."""
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
