"""Which work-groups share a CU/SIMD, and are they in phase?  (diagnostic variant 2)"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import phylocsfpp_amd as pa
from phylocsfpp_amd import synth
import oracle_binding as ob
m = ob.OracleModel(os.path.join(ROOT, "tests/golden/models/58mammals"))
n_sites = 2_000_000
pool = synth.synthetic_codon_columns(m.child1, m.child2, m.branch_len, m.S_c, m.pi_c, m.S_nc, m.pi_nc, 16384, seed=1)
idx = np.random.default_rng(2).integers(0, 16384, size=n_sites)
d = torch.from_numpy(np.ascontiguousarray(pool[:, idx])).cuda()
out = torch.empty(n_sites, dtype=torch.float64, device="cuda")
for stag in ("0", "4"):
    os.environ["PCSF_PRUNE_VARIANT"] = "2"; os.environ["PCSF_STAGGER"] = stag
    c = pa.ScoringContext(m.child1, m.child2, m.branch_len, m.S_c, m.pi_c, m.S_nc, m.pi_nc)
    lib = pa.load_library()
    lib.pcsf_debug_stamps.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.c_int]
    n = lib.pcsf_debug_stamps(c._h, 0, None, 0)
    c.score_sites_fixed_torch(d, out=out); torch.cuda.synchronize()
    buf = np.zeros(n, dtype=np.uint64)
    lib.pcsf_debug_stamps(c._h, 0, buf.ctypes.data_as(C.POINTER(C.c_uint64)), n)
    cz = buf[-16384:].reshape(1024, 16).astype(np.int64)
    hw = cz[:, 0]
    T = cz[:, 1:9]                                       # 8 samples: model 0 ops 10,40,70,100 then model 1
    period = np.median((T[:512, 1] - T[:512, 0]) / 30.0)
    print("stagger", stag, "op period %.0f cycles" % period)
    for k in range(8):
        dd = T[:256, k] - T[256:512, k]                   # co-resident pairs are (b, b+256)
        ph = np.mod(dd, period) / period
        ph = np.minimum(ph, 1 - ph)                      # distance to in-phase, 0..0.5
        print("   sample %d: |phase offset| as a fraction of the op period: median %.2f  p10 %.2f  p90 %.2f" % (k, np.median(ph), np.percentile(ph, 10), np.percentile(ph, 90)))
    c.close()
