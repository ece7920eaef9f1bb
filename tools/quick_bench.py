"""Quick kernel timing on the GPU box: python tools/quick_bench.py [model] [n_sites]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import phylocsfpp_amd as pa
from phylocsfpp_amd import synth
import oracle_binding as ob

# A/B of experiment knobs: PCSF_PRIO_MODE, PCSF_STAGGER, PCSF_EXTRA_LDS are read at context creation

model = sys.argv[1] if len(sys.argv) > 1 else "58mammals"
n_sites = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
variants = [int(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0, 1]
m = ob.OracleModel(os.path.join(ROOT, "tests/golden/models", model))
pool = synth.synthetic_codon_columns(m.child1, m.child2, m.branch_len, m.S_c, m.pi_c, m.S_nc, m.pi_nc, 16384, seed=1)
cod = np.tile(pool, (1, (n_sites + 16383) // 16384))[:, :n_sites]
d = torch.from_numpy(np.ascontiguousarray(cod)).cuda()
out = torch.empty(n_sites, dtype=torch.float64, device="cuda")
nl = m.nl
flop_per_site = 2 * ((nl - 2) * 2 * 64 * 64 + (nl - 1) * 64 + 2 * 64)
want = m.run_tracks(pool[:, :512])
for v in variants:
    os.environ["PCSF_PRUNE_VARIANT"] = str(v)
    ctx = pa.ScoringContext(m.child1, m.child2, m.branch_len, m.S_c, m.pi_c, m.S_nc, m.pi_nc)
    ctx.score_sites_fixed_torch(d, out=out); torch.cuda.synchronize()
    err = np.abs(out[:512].cpu().numpy() - want).max()
    ctx.set_timing(True)
    ts = []
    for _ in range(5):
        ctx.score_sites_fixed_torch(d, out=out)
        ts.append(ctx.last_kernel_ms())
    ms = min(ts)
    print("variant %d: %.3f ms for %d sites -> %.2f Msites/s, %.2f TFLOP/s (algorithmic), max|err| %.2e"
          % (v, ms, n_sites, n_sites / ms * 1e-3, n_sites * flop_per_site / ms * 1e-9, err), flush=True)
    ctx.close()
