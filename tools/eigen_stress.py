import os, sys, ctypes as C
ROOT = "/root/repo" if os.path.isdir("/root/repo/tests") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import phylocsfpp_amd as pa
import oracle_binding as ob
ctx = pa.load_model(os.path.join(ROOT, "tests/golden/models/100vertebrates")).context()
rng = np.random.default_rng(2026)
dp = C.POINTER(C.c_double)
def check(qs, right, left, lam, prior, failed, tag):
    worst = np.zeros(3); nf = int(failed.sum())
    for i in range(len(qs)):
        if failed[i]: continue
        Q, pi = np.empty((64, 64)), np.empty(64)
        ob.lib().po_q_from_omega(np.ascontiguousarray(qs[i]).ctypes.data_as(dp), Q.ctypes.data_as(dp), pi.ctypes.data_as(dp))
        scale = np.abs(Q).max()
        worst = np.maximum(worst, [np.abs((right[i] * lam[i]) @ left[i] - Q).max() / scale, np.abs(left[i] @ right[i] - np.eye(64)).max(), np.abs(prior[i] - pi / pi.sum()).max()])
    print(tag, "jobs", len(qs), "failed", nf, "worst R L R - Q %.2e, L R - I %.2e, prior %.2e" % tuple(worst), flush=True)
for form in ("2", "1"):
    os.environ["PCSF_EIGEN_FORM"] = form
    n = 3000
    qs = np.empty((n, 12))
    qs[:, 0] = rng.uniform(1.0, 10.0, n)
    qs[:, 1] = np.exp(rng.uniform(np.log(1e-6), np.log(100.0), n))
    qs[:, 2] = np.exp(rng.uniform(np.log(1e-6), np.log(100.0), n))
    qs[:, 3:] = np.exp(rng.uniform(np.log(1e-4), np.log(1e4), (n, 9)))
    qs[:50, 3:] = 1.0; qs[:50, 1:3] = 1.0          # the degenerate family: uniform frequencies
    qs[50:100, 0] = 1.0                             # kappa = 1: transitions = transversions
    r = ctx.debug_eigen_omega(qs)
    check(qs, *r, "form %s cold" % form)
    ws = qs.copy(); ws[:, 0] = rng.uniform(1.0, 10.0, n); ws[::3, 1:3] = [1.0, 1.0]
    r = ctx.debug_eigen_omega(qs, warm_settings=ws)
    check(qs, *r, "form %s warm from anywhere" % form)
