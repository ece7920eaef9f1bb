"""In-kernel stamp diagnostic (variant 2): where one wave's op time goes. python tools/stamps.py [block]"""
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
VAR = os.environ.get("STAMP_VARIANT", "2")
for extra in (("0", "30000") if VAR == "2" else ("0",)):
    os.environ["PCSF_PRUNE_VARIANT"] = VAR; os.environ["PCSF_EXTRA_LDS"] = extra
    c = pa.ScoringContext(m.child1, m.child2, m.branch_len, m.S_c, m.pi_c, m.S_nc, m.pi_nc)
    lib = pa.load_library()
    lib.pcsf_debug_stamps.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_uint64), C.c_int]
    for block in [int(a) for a in sys.argv[1:]] or [5000, 20000]:
        n = lib.pcsf_debug_stamps(c._h, block, None, 0)
        c.score_sites_fixed_torch(d, out=out); torch.cuda.synchronize()
        buf = np.zeros(n, dtype=np.uint64)
        lib.pcsf_debug_stamps(c._h, block, buf.ctypes.data_as(C.POINTER(C.c_uint64)), n)
        n_ops = len(pa.build_schedule(m.child1, m.child2)[0])
        st = buf[:2 * n_ops * 8].reshape(2, -1, 8).astype(np.int64)
        ops = st[:, 1:-1, :]                      # ops with a product (skip first cherry and root)
        wait_bar = ops[:, :, 1] - ops[:, :, 0]
        prod = ops[:, :, 2] - ops[:, :, 1]
        epi = ops[:, :, 3] - ops[:, :, 2]
        gap = ops[:, 1:, 0] - ops[:, :-1, 3]      # epilogue end -> next op start (loop + decode)
        period = ops[:, 1:, 0] - ops[:, :-1, 0]
        print("extra_lds=%s block=%d: per-op cycles (s_memtime ticks @100MHz?? see ratio): barrier-wait %.0f  product %.0f  epilogue %.0f  loop/decode %.0f  period %.0f"
              % (extra, block, wait_bar.mean(), prod.mean(), epi.mean(), gap.mean(), period.mean()))
        prog, slots, depth = pa.build_schedule(m.child1, m.child2)
        codes = (prog >> 28).astype(int)[1:-1]
        for cname, sel in (("MM_LEAF", codes == 1), ("MM_PUSH_CHERRY", (codes >= 2) & (codes < 5)), ("MM_POP", (codes >= 5) & (codes < 8))):
            if sel.any():
                print("   %-15s n=%2d  barrier %.0f  product %.0f  epilogue %.0f  gap-after %.0f" % (
                    cname, sel.sum(), wait_bar[:, sel].mean(), prod[:, sel].mean(), epi[:, sel].mean(),
                    gap[:, sel[:-1]].mean()))
        leaf = codes == 1
        cal = (ops[:, :, 4] - ops[:, :, 3])[:, leaf]
        top = ops[:, 1:, 5] - ops[:, :-1, 3]          # epilogue end -> next loop top (after decode + LDS issue)
        disp = ops[:, :, 0] - ops[:, :, 5]            # loop top -> arm start (switch dispatch)
        print("   stamp cost (back-to-back) %.0f ; arm-exit+latch+decode %.0f ; switch dispatch %.0f" % (cal.mean(), top.mean(), disp.mean()))
        print("   first 12 ops product:", prod[0, :12].tolist(), "codes", codes[:12].tolist())
        print("   product min/max %d/%d  barrier min/max %d/%d  total %d" % (prod.min(), prod.max(), wait_bar.min(), wait_bar.max(), st[1, -1, 0] - st[0, 1, 0]))
    c.close()
