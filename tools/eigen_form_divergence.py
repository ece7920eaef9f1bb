#!/usr/bin/env python3
"""Where do the eigen-solver's forms (PCSF_EIGEN_FORM=1 | 2) lead OMEGA apart?  (1) the edge blocks of tests/test_gpu_optimizer.py under
both forms; (2) the eigen-systems of extreme settings under both forms against numpy: reconstruction of Q, L R = I, the prior."""
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
cols[:, int(off[5]):int(off[6])] = 17
ctx = pa.load_model(os.path.join(GOLD, "models", "53birds")).context()
for form in ("1", "2", "0"):
    os.environ["PCSF_EIGEN_FORM"] = form
    for kernel in (pa.ScoringContext.BLOCK_KERNEL_TILES, pa.ScoringContext.BLOCK_KERNEL_FACTORED):
        ctx.set_block_kernel(kernel)
        ph = ctx.score_blocks(pa.OMEGA, cols, off, compute_anc=False)[0]
        print("form", form, "kernel", kernel, ["%.5f" % x for x in ph], flush=True)
ctx.close()
ctx = pa.load_model(os.path.join(GOLD, "models", "100vertebrates")).context()
rng = np.random.default_rng(99)
n = 400
qs = np.empty((n, 12))
qs[:, 0] = rng.uniform(1.0, 10.0, n)
qs[:, 1] = np.exp(rng.uniform(np.log(1e-4), np.log(50.0), n))
qs[:, 2] = np.exp(rng.uniform(np.log(1e-4), np.log(50.0), n))
qs[:, 3:] = np.exp(rng.uniform(np.log(1e-3), np.log(1e3), (n, 9)))
dp = C.POINTER(C.c_double)
for form in ("1", "2"):
    os.environ["PCSF_EIGEN_FORM"] = form
    right, left, lam, prior, failed = ctx.debug_eigen_omega(qs)
    worst = np.zeros(4)
    for i in range(n):
        if failed[i]:
            continue
        Q, pi = np.empty((64, 64)), np.empty(64)
        ob.lib().po_q_from_omega(np.ascontiguousarray(qs[i]).ctypes.data_as(dp), Q.ctypes.data_as(dp), pi.ctypes.data_as(dp))
        scale = np.abs(Q).max()
        e = [np.abs((right[i] * lam[i]) @ left[i] - Q).max() / scale, np.abs(left[i] @ right[i] - np.eye(64)).max(),
             np.abs(prior[i] - pi / pi.sum()).max(), np.abs(prior[i] / (pi / pi.sum()) - 1).max()]
        worst = np.maximum(worst, e)
    print("form", form, "failed", int(failed.sum()), "worst: R L R - Q %.2e (of max |Q|), L R - I %.2e, prior abs %.2e rel %.2e" % tuple(worst), flush=True)
