#!/usr/bin/env python3
"""Are repeated six-frame calls bit-identical?  (diagnostic; needs a GPU)  python tools/diag_determinism.py [calls]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import phylocsfpp_amd as pa
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 60
m = pa.load_model(os.path.join(ROOT, "tests/golden/models/58mammals"))
rng = np.random.default_rng(2)
L = 300_000
lut = np.frombuffer(b"ACGTacgtN-", dtype=np.uint8)
nuc = lut[rng.integers(0, 10, size=(m.nl, L))]
alns = []
c = 0
while c < L:
    k = int(min(rng.integers(50, 20000), L - c))
    alns.append((c, k, int(rng.integers(1, 10**6)), 3 * 10**6))
    c += k
for fused in ("1", "0"):
    os.environ["PCSF_FUSED_FRONTEND"] = fused
    ctx = m.context(device=0, with_bls=True)
    outs = []
    for r in range(calls):
        res = ctx.score_alignments_6frames(nuc, alns, with_bls=(r % 2 == 0))
        outs.append(np.concatenate([res[0][k][f] for k in range(len(alns)) for f in range(6)]))
    outs = np.stack(outs)
    ref = np.median(outs, axis=0)
    wrong = [(int((outs[r] != ref).sum())) for r in range(calls)]
    print("fused=%s: %d calls x %d codon-frames: calls with a wrong value: %d, wrong values per call: %s"
          % (fused, calls, outs.shape[1], sum(1 for w in wrong if w), [w for w in wrong if w]))
    ctx.close()
