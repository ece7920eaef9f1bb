#!/usr/bin/env python3
"""How well is the whole-block objective defined?  (diagnostic; needs a GPU)

Relative difference to the oracle of ONE evaluation of the kappa/omega objective, per tree scale rho, for three device
paths: explicit matrices with the ORACLE's eigen-system injected (only the summation order of P(t) and the pruning
differ), explicit matrices with the device eigen-solver, factored pruning with the device eigen-solver.
"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import maf_blocks
import oracle_binding as ob
import phylocsfpp_amd as pa

G = os.path.join(ROOT, "tests/golden")
prefix = G + "/models/100vertebrates"
m = ob.OracleModel(prefix)
cod, off = maf_blocks.block_codons(G + "/score-msa/chr22.50alignments.maf", m.labels, G + "/models/species_aliases.tsv")
n_cod = np.diff(off.astype(np.int64))
sel = [i for i in range(len(n_cod)) if n_cod[i] > 0]
subs = [np.ascontiguousarray(cod[:, int(off[i]):int(off[i + 1])]) for i in sel]
sub = np.ascontiguousarray(np.concatenate(subs, axis=1))
sub_off = np.concatenate([[0], np.cumsum([s.shape[1] for s in subs])]).astype(np.uint64)
ctx = pa.load_model(prefix).context()


def oracle_lpr(qd, s, x):
    lpr, anc = C.c_double(), C.c_double()
    assert ob.lib().po_lpr_leaves(C.byref(qd), C.byref(m.m.tree), x, s.ctypes.data_as(C.POINTER(C.c_uint8)), s.shape[1], s.shape[1], 0,
                                  C.byref(lpr), C.byref(anc), None) == 0
    return lpr.value


rng = np.random.default_rng(4)
print("%d blocks of the 50-block golden file; relative |device - oracle| of the objective: median / max over blocks" % len(subs))
print("%-22s %8s  %-21s %-21s %-21s" % ("kappa,omega,sigma", "rho", "oracle eig, explicit", "device eig, explicit", "device eig, factored"))
for kappa, omega, sigma in ((2.5, 1.0, 1.0), (1.0, 0.2, 0.01), (7.3, 0.2, 0.01)):
    qs = np.concatenate([[kappa, omega, sigma], rng.uniform(0.5, 1.5, 9)])
    qd = m.eig_of_settings(qs)
    R, L = np.array(qd.Sr).reshape(64, 64), np.array(qd.Sinv).reshape(64, 64)
    lam, prior = np.array(qd.lam), np.array(qd.prior)
    for rho in (0.001, 0.003, 0.01, 0.02, 0.05, 0.1, 0.3, 1.0, 3.0, 10.0):
        want = np.array([oracle_lpr(qd, s, rho) for s in subs])
        cols = []
        got, _ = ctx.block_loglik_eig(sub, sub_off, rho, R, L, lam, prior)
        cols.append(np.abs(got - want) / np.abs(want))
        for tiles in ("0", "1000"):
            os.environ["PCSF_FACTORED_MAX_TILES"] = tiles
            got, _ = ctx.block_loglik(2, sub, sub_off, rho, q_settings=qs)
            cols.append(np.abs(got - want) / np.abs(want))
        print("%-22s %8g  %s" % ("%g,%g,%g" % (kappa, omega, sigma), rho, "  ".join("%9.2e / %9.2e" % (np.median(c), c.max()) for c in cols)))
