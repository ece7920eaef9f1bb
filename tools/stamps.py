"""In-kernel stamp diagnostic (variant 2): where one wave's time goes, as per-segment cycle sums kept in
scalar registers (nothing is stored inside the loop).  python tools/stamps.py [block ...]"""
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
names = {0: "dispatch (loop top -> arm start)", 1: "barrier (+vmcnt) wait", 2: "product (leaf / pop arms)", 4: "product (push+cherry arms)", 3: "epilogue", 5: "decode + LDS issue",
         7: "arm exit + loop latch"}
for extra in ("0", "30000"):
    os.environ["PCSF_PRUNE_VARIANT"] = os.environ.get("STAMP_VARIANT", "2"); os.environ["PCSF_EXTRA_LDS"] = extra
    c = pa.ScoringContext(m.child1, m.child2, m.branch_len, m.S_c, m.pi_c, m.S_nc, m.pi_nc)
    lib = pa.load_library()
    lib.pcsf_debug_stamps.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.c_int]
    for block in [int(a) for a in sys.argv[1:]] or [5000]:
        n = lib.pcsf_debug_stamps(c._h, block, None, 0)
        c.score_sites_fixed_torch(d, out=out); torch.cuda.synchronize()
        buf = np.zeros(n, dtype=np.uint64)
        lib.pcsf_debug_stamps(c._h, block, buf.ctypes.data_as(C.POINTER(C.c_uint64)), n)
        sums, cnts = buf[:8].astype(np.float64), buf[8:16].astype(np.float64)
        n_prod = cnts[2] + cnts[4]
        print("%s work-group(s) per CU, block %d: %d products; cycles per product-op (each segment includes one ~76-cycle stamp):"
              % ("two" if extra == "0" else "one", block, n_prod))
        tot = 0.0
        for k in (7, 5, 0, 1, 2, 4, 3):
            if cnts[k]:
                per = sums[k] / n_prod
                tot += per
                print("   %-36s %8.0f   (%d stamps, %.0f each)" % (names[k], per, cnts[k], sums[k] / cnts[k]))
        print("   %-36s %8.0f" % ("sum = op period", tot))
        print("   work-group prologue (once): %.0f cycles = %.1f product periods" % (sums[6], sums[6] / tot))
    c.close()
