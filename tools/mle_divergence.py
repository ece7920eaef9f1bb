#!/usr/bin/env python3
"""Where does an MLE row of the device leave the oracle's path?  ("a row beyond the reference's tolerance where the
control is still -- find the difference".)  For the given rows of the bench's synthetic MAF (BASELINE.json configs[2], 58mammals):

  1. the oracle's search is traced (po_run_trace: every evaluation point and value of the coding and of the non-coding search);
  2. the DEVICE evaluates the objective at the oracle's points (pcsf_block_loglik) -> relative differences of the values;
  3. the device's own search (the state machine of csrc/chain.hip, pcsf_debug_maxlik_device) is fed the DEVICE's values at the oracle's
     points: as long as it takes the oracle's decisions it asks for the oracle's next point; the first request that differs is where
     the two paths part, and the values there show what decided it.

    python tools/mle_divergence.py 889 1191 3912 3918 4146          (needs a GPU; ~1 min to read the MAF)"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import gen_synthetic_maf as gen
import maf_blocks
import oracle_binding as ob
import phylocsfpp_amd as pa

rows = [int(a) for a in sys.argv[1:]] or [889, 1191, 3912, 3918, 4146]
GOLD = os.path.join(ROOT, "tests", "golden")
m = ob.OracleModel(os.path.join(GOLD, "models", "58mammals"))
ctx = pa.load_model("58mammals").context()
with tempfile.TemporaryDirectory() as tmp:
    maf = os.path.join(tmp, "synthetic_1mb.maf")
    gen.generate(maf, bases=1_000_000, seed=20261002)
    cod, off = maf_blocks.block_codons(maf, m.labels, os.path.join(GOLD, "models", "species_aliases.tsv"), max_blocks=max(rows) + 1)
LN10 = np.log(10.0)
for b in rows:
    block = np.ascontiguousarray(cod[:, int(off[b]):int(off[b + 1])])
    ph_o, an_o, entries = m.run_trace(ob.MLE, block, compute_anc=True)
    ph_d, an_d = ctx.score_blocks(pa.MLE, block, np.array([0, block.shape[1]], dtype=np.uint64), compute_anc=True)
    print("row %d: %d codons; oracle %.6f, device %.6f (|d| %.4f)" % (b, block.shape[1], ph_o, ph_d[0], abs(ph_o - ph_d[0])))
    last = {}
    for kind, name in ((0, "coding"), (1, "non-coding")):
        E = [e for e in entries if e.kind == kind and not e.failed]
        xs = np.array([e.x for e in E])
        vo = np.array([e.lpr for e in E])
        tiled = np.ascontiguousarray(np.tile(block, (1, len(E))))
        toff = (np.arange(len(E) + 1) * block.shape[1]).astype(np.uint64)
        vd, failed = ctx.block_loglik(kind, tiled, toff, xs)
        rel = np.abs(vd - vo) / np.abs(vo)
        # the number of 32-bit draws the searches before this one consumed: the one that makes the search reproduce the oracle's own points
        skip = None
        for cand in range(0, 1100, 2):
            got, n_req, _ = ctx.debug_maxlik_device(1.0, 0.01, 10.0, vo, skip_draws=cand)
            if n_req == len(xs) and np.array_equal(got[:n_req], xs):
                skip = cand
                break
        assert skip is not None, "cannot reproduce the oracle's path from its own values"
        # the device's OWN path: its search is driven step by step, every point it asks for evaluated by the device
        one = (np.arange(2) * block.shape[1]).astype(np.uint64)
        dx = [float(xs[0])]                                   # (every search starts at the same point: its first request needs no value)
        dv = [float(ctx.block_loglik(kind, block, one, dx[0])[0][0])]
        while True:
            got, n_req, last_dev = ctx.debug_maxlik_device(1.0, 0.01, 10.0, np.array(dv), skip_draws=skip)
            if n_req >= 0:
                break
            x_next = float(got[len(dv)])
            dx.append(x_next)
            dv.append(float(ctx.block_loglik(kind, block, one, x_next)[0][0]))
        dx, dv = np.array(dx), np.array(dv)
        n_cmp = min(len(xs), len(dx))
        apart = np.abs(dx[:n_cmp] - xs[:n_cmp]) > 1e-9 * xs[:n_cmp]
        first = int(np.argmax(apart)) if apart.any() else None
        print("  %s search: the oracle takes %d evaluations and ends at x = %.9g (lpr %.9f); the device takes %d and ends at x = %.9g (lpr %.9f); at the oracle's points the "
              "two objectives agree to %.1e relative (worst), %.1e (median)" % (name, len(xs), xs[-1], vo[-1], len(dx), dx[-1], dv[-1], rel.max(), np.median(rel)))
        if first is None and len(dx) == len(xs):
            print("    same path (every point within 1e-9)")
        elif first is None:
            print("    same points up to evaluation %d, then one search stops and the other takes %d more step(s)" % (n_cmp - 1, abs(len(dx) - len(xs))))
        else:
            print("    the paths part at evaluation %d: oracle x = %.12g, device x = %.12g; the evaluations before it:" % (first, xs[first], dx[first]))
            for jj in range(max(0, first - 3), first):
                print("      evaluation %d: x = %.12g (device %.12g)  oracle %.15g  device %.15g  (relative difference %.1e)" % (jj, xs[jj], dx[jj], vo[jj], dv[jj], abs(dv[jj] - vo[jj]) / abs(vo[jj])))
        last[kind] = (vo[-1], dv[-1])
    print("  end result = 10 (lpr_coding - lpr_noncoding) / ln 10 at the LAST evaluation of each search: oracle %.6f, the device's own paths %.6f" % (
        10 * (last[0][0] - last[1][0]) / LN10, 10 * (last[0][1] - last[1][1]) / LN10))
ctx.close()
