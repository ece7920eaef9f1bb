"""Sustained rate of the block-mode pruning kernel: python tools/tile_bench.py [blocks] [sites per block] [model] [dropout]
One objective evaluation (pcsf_block_loglik, ECM coding model, tree scale 1) of `blocks` synthetic blocks; prints the call's wall
time and, from it, an upper bound of the kernel's time per 16-column tile.  Run under rocprofv3 --kernel-trace --stats for the kernel's own time."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import phylocsfpp_amd as pa
from phylocsfpp_amd import synth
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
per = int(sys.argv[2]) if len(sys.argv) > 2 else 64
model = sys.argv[3] if len(sys.argv) > 3 else "58mammals"
dropout = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
m = pa.load_model(os.path.join(ROOT, "tests/golden/models", model))
pool = synth.synthetic_codon_columns(m.child1, m.child2, m.branch_len, m.S_c, m.pi_c, m.S_nc, m.pi_nc, 1 << 14, seed=5, dropout=dropout)
rng = np.random.default_rng(1)
cod = np.ascontiguousarray(pool[:, rng.integers(0, pool.shape[1], nb * per)])
offs = (np.arange(nb + 1) * per).astype(np.uint64)
ctx = m.context()
x = np.ones(nb)
ctx.block_loglik(0, cod[:, :per * 8], offs[:9], x[:8])
for rep in range(3):
    t = time.time()
    ctx.block_loglik(0, cod, offs, x)
    dt = time.time() - t
    tiles = nb * ((per + 15) // 16)
    nl = m.nl
    mfma = tiles * (nl + 2 * (nl - 2)) * 64
    print("%d blocks x %d sites (%d tiles), %s: %.4f s -> %.1f us per tile per 2048 wave slots; MFMA-peak time %.4f s (%.2f of the call)" % (
        nb, per, tiles, model, dt, dt / tiles * 2048 * 1e6, mfma * 64 / (1024 * 2.4e9), mfma * 64 / (1024 * 2.4e9) / dt))
