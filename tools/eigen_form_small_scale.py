#!/usr/bin/env python3
"""The OMEGA objective at SMALL tree scales under the eigen-solver's forms, against the oracle and the oracle with its Jacobi sweeps run in
the opposite order (po_set_perturbation 32): relative differences of ln z on the 700-site edge block of tests/test_gpu_optimizer.py."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import phylocsfpp_amd as pa
import oracle_binding as ob
from phylocsfpp_amd import synth
GOLD = os.path.join(ROOT, "tests", "golden")
m = ob.OracleModel(os.path.join(GOLD, "models", "53birds"))
lens = [0, 1, 17, 0, 700, 5, 40]
cols = synth.synthetic_codon_columns(m.child1, m.child2, m.branch_len, m.S_c, m.pi_c, m.S_nc, m.pi_nc, sum(lens), seed=5, dropout=0.2, gap_rate=0.05)
off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
sub = np.ascontiguousarray(cols[:, int(off[4]):int(off[5])])
sub_off = np.array([0, 700], dtype=np.uint64)
ctx = pa.load_model(os.path.join(GOLD, "models", "53birds")).context()
dp = C.POINTER(C.c_double)
def oracle(qs, scale, bits):
    ob.lib().po_set_perturbation(bits)
    Q, pi = np.empty((64, 64)), np.empty(64)
    ob.lib().po_q_from_omega(qs.ctypes.data_as(dp), Q.ctypes.data_as(dp), pi.ctypes.data_as(dp))
    qd = ob.PoQdiag()
    ob.lib().po_eig(Q.ctypes.data_as(dp), pi.ctypes.data_as(dp), C.byref(qd))
    lpr, anc = C.c_double(), C.c_double()
    assert ob.lib().po_lpr_leaves(C.byref(qd), C.byref(m.m.tree), scale, sub.ctypes.data_as(C.POINTER(C.c_uint8)), sub.shape[1], sub.shape[1], 0, C.byref(lpr), C.byref(anc), None) == 0
    ob.lib().po_set_perturbation(0)
    return lpr.value
ratios = [1.105, 0.733, 0.754, 0.974, 0.808, 0.831, 1.02, 0.95, 0.88]
models = {"starting model": np.array([2.5, 1.0, 1.0] + [1.0] * 9), "H0-like": np.array([3.29, 1.0, 1.0] + ratios), "H1-like": np.array([3.29, 0.2, 0.01] + ratios)}
print("relative difference of ln z from the oracle's: oracle with the other sweep order | device FORM 0 (LDS) | FORM 1 (two waves) | FORM 2 (seven waves)")
for name, qs in models.items():
    for scale in (0.001, 0.0025, 0.0054, 0.0216, 0.0607, 0.244, 1.0):
        base = oracle(qs, scale, 0)
        row = [abs(oracle(qs, scale, 32) - base) / abs(base)]
        for form in ("0", "1", "2"):
            os.environ["PCSF_EIGEN_FORM"] = form
            got, failed = ctx.block_loglik(2, sub, sub_off, scale, q_settings=qs)
            row.append(abs(got[0] - base) / abs(base))
        print("%-15s scale %-7g  %.1e | %.1e | %.1e | %.1e" % (name, scale, *row), flush=True)
