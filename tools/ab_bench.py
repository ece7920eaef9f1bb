"""A/B timing of kernel knobs in ONE process (interleaved rounds): python tools/ab_bench.py [n_sites]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import phylocsfpp_amd as pa
from phylocsfpp_amd import synth
import oracle_binding as ob

n_sites = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
model = sys.argv[2] if len(sys.argv) > 2 else "58mammals"
m = ob.OracleModel(os.path.join(ROOT, "tests/golden/models", model))
pool = synth.synthetic_codon_columns(m.child1, m.child2, m.branch_len, m.S_c, m.pi_c, m.S_nc, m.pi_nc, 16384, seed=1)
idx = np.random.default_rng(2).integers(0, 16384, size=n_sites)
d = torch.from_numpy(np.ascontiguousarray(pool[:, idx])).cuda()
out = torch.empty(n_sites, dtype=torch.float64, device="cuda")
want = m.run_tracks(pool[:, idx[:256]])
configs = [("v0", dict(PCSF_PRUNE_VARIANT="0")), ("v1 8w", dict(PCSF_PRUNE_VARIANT="1")),
           ("v0x1 one WG/CU", dict(PCSF_PRUNE_VARIANT="0", PCSF_EXTRA_LDS="30000"))]
if len(sys.argv) > 3:
    configs = [c for c in configs if c[0].split()[0] in sys.argv[3].split(",")]
ctxs = []
for name, env in configs:
    for k in ("PCSF_PRUNE_VARIANT", "PCSF_EXTRA_LDS", "PCSF_STAGGER"):
        os.environ.pop(k, None)
    os.environ.update(env)
    c = pa.ScoringContext(m.child1, m.child2, m.branch_len, m.S_c, m.pi_c, m.S_nc, m.pi_nc)
    c.score_sites_fixed_torch(d, out=out); torch.cuda.synchronize()
    err = np.abs(out[:256].cpu().numpy() - want).max()
    c.set_timing(True)
    ctxs.append((name, c, err, []))
for r in range(5):
    for name, c, err, ts in ctxs:
        c.score_sites_fixed_torch(d, out=out)
        ts.append(c.last_kernel_ms())
fl = 2 * ((m.nl - 2) * 8192 + (m.nl - 1) * 64 + 128)
for name, c, err, ts in ctxs:
    ms = float(np.median(ts))
    print("%-14s median %.3f ms (min %.3f)  %.2f Msites/s  %.2f TFLOP/s  err %.1e" % (name, ms, min(ts), n_sites / ms * 1e-3, n_sites * fl / ms * 1e-9, err), flush=True)
