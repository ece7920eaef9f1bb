"""Cycles of one prune_factored work-group by tile content (PCSF_FACTORED_STAMPS=1): all-gap tiles (the op loop and prologue alone),
tiles with 1 / 4 / 20 / 100 species present.  python tools/factored_tile_cost.py [blocks per call]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PCSF_FACTORED_STAMPS", "1")   # 2: also cycles per class of op (every op that runs is stamped, ~400 cycles each)
import numpy as np
import phylocsfpp_amd as pa
m = pa.load_model(os.path.join(ROOT, "tests/golden/models/100vertebrates"))
ctx = m.context()
rng = np.random.default_rng(1)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 200     # <= one work-group per CU
cols = 6
offs = (np.arange(nb + 1) * cols).astype(np.uint64)
for present in (0, 1, 4, 20, 100):
    cod = np.full((m.nl, nb * cols), 64, np.uint8)
    for b in range(nb):
        for l in (rng.choice(m.nl, size=present, replace=False) if present else []):
            cod[l, b * cols:(b + 1) * cols] = rng.integers(0, 61, size=cols)
    sys.stderr.write("%d species present, %d blocks of %d columns: " % (present, nb, cols)); sys.stderr.flush()
    ctx.block_loglik(0, cod, offs, 1.0)
