"""Throughput of the batched P(t) kernel alone: block_loglik over blocks WITHOUT sites queues only their n-1 branch
matrices.  python tools/expm_bench.py [n_blocks]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import phylocsfpp_amd as pa
import oracle_binding as ob
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
m = ob.OracleModel(os.path.join(ROOT, "tests/golden/models/100vertebrates"))
ctx = pa.ScoringContext(m.child1, m.child2, m.branch_len, m.S_c, m.pi_c, m.S_nc, m.pi_nc)
cod = np.zeros((m.nl, 1), np.uint8)
off = np.zeros(nb + 1, np.uint64)
ctx.block_loglik(0, cod, off, 1.0)
ts = []
for r in range(3):
    t = time.time(); ctx.block_loglik(0, cod, off, 0.5 + 0.1 * r); ts.append(time.time() - t)
jobs = nb * (2 * m.nl - 2)
t = min(ts)
print("%d jobs in %.4f s: %.1f ns per job, %.2f TFLOP/s (2*64^3 per job), %.2f TB/s of tables written" % (jobs, t, t / jobs * 1e9, jobs * 524288 / t * 1e-12, jobs * 33000 / t * 1e-12))
