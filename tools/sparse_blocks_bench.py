import os, sys, time
ROOT="/root/repo"
sys.path.insert(0, ROOT)
import numpy as np
import phylocsfpp_amd as pa
m = pa.load_model(os.path.join(ROOT, "tests/golden/models/100vertebrates"))
ctx = m.context()
rng = np.random.default_rng(1)
nb, cols = 4096, 6
offs = (np.arange(nb + 1) * cols).astype(np.uint64)
for name, present in (("all gaps", 0), ("4 species", 4), ("20 species", 20), ("100 species", 100)):
    cod = np.full((m.nl, nb * cols), 64, np.uint8)
    for b in range(nb):
        sp = rng.choice(m.nl, size=present, replace=False) if present else []
        for l in sp:
            cod[l, b * cols:(b + 1) * cols] = rng.integers(0, 61, size=cols)
    os.environ["PCSF_TIME_BLOCKS"] = "1"
    ctx.score_blocks(pa.MLE, cod[:, :4 * cols], offs[:5], compute_anc=True)
    t = time.time(); ctx.score_blocks(pa.MLE, cod, offs, compute_anc=True); dt = time.time() - t
    print("%-12s MLE %d blocks: %.3f s" % (name, nb, dt), flush=True)
