#!/usr/bin/env python3
"""Where do the GPU's and the oracle's OMEGA runs part?  (diagnostic; needs a GPU)

  python tools/omega_divergence.py [n_blocks] [small|medium]

For the first n non-empty blocks of the golden score-msa file: the oracle's evaluation sequence (po_run_trace) beside the
device's (PCSF_TRACE_BLOCK: stage, x, objective incl. prior), evaluation by evaluation.  Prints, per block, the index of
the first evaluation whose POINT differs, the largest relative difference of the objective VALUES before that, and the
two end results -- i.e. whether the end results differ because an objective value differed visibly or because a
rounding-level difference flipped a decision of the optimiser.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

G = os.path.join(ROOT, "tests/golden")
FILES = {"small": "chr22.50alignments.maf", "medium": "chr22.516alignments.maf.gz"}

CHILD = r"""
import os, sys
sys.path.insert(0, %r); sys.path.insert(0, os.path.join(%r, 'tests'))
import numpy as np, maf_blocks, phylocsfpp_amd as pa
m = pa.load_model(%r)
cod, off = maf_blocks.block_codons(%r, m.labels, %r)
b = int(sys.argv[1])
sub = np.ascontiguousarray(cod[:, int(off[b]):int(off[b + 1])])
ctx = m.context()
ph, _ = ctx.score_blocks(pa.OMEGA, sub, np.array([0, sub.shape[1]], dtype=np.uint64), compute_anc=False)
print('RESULT %%.9g' %% ph[0])
"""


def main():
    n_blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    tag = sys.argv[2] if len(sys.argv) > 2 else "small"
    import maf_blocks
    import oracle_binding as ob
    prefix = G + "/models/100vertebrates"
    m = ob.OracleModel(prefix)
    maf = G + "/score-msa/" + FILES[tag]
    aliases = G + "/models/species_aliases.tsv"
    cod, off = maf_blocks.block_codons(maf, m.labels, aliases)
    child = CHILD % (ROOT, ROOT, prefix, maf, aliases)
    done = 0
    for b in range(len(off) - 1):
        if off[b + 1] == off[b]:
            continue
        if done == n_blocks:
            break
        done += 1
        sub = np.ascontiguousarray(cod[:, int(off[b]):int(off[b + 1])])
        oph, _, entries = m.run_trace(ob.OMEGA, sub)
        o_seq = [(e.x, e.lpr + (ob.lib().po_lpr_rho(e.x) if e.kind == 2 else ob.lib().po_lpr_kappa(e.x))) for e in entries]
        r = subprocess.run([sys.executable, "-c", child, str(b)], capture_output=True, text=True, env=dict(os.environ, PCSF_TRACE_BLOCK="0"))
        g_seq = [(float(a), float(v)) for a, v in re.findall(r"TRACE omega stage \d+ x (\S+) lpr (\S+)", r.stderr)]
        gph = float(re.search(r"RESULT (\S+)", r.stdout).group(1))
        first, worst = None, 0.0
        for i, ((ox, ov), (gx, gv)) in enumerate(zip(o_seq, g_seq)):
            if ox != gx:
                first = i
                break
            worst = max(worst, abs(ov - gv) / abs(ov))
        print("block %3d (%4d codons): %3d / %3d evaluations; first differing point: %s; worst objective difference before it %.2e; "
              "end results oracle %.6f gpu %.6f (d = %.2e)%s"
              % (b, sub.shape[1], len(o_seq), len(g_seq), "none" if first is None else "#%d (x %.9g vs %.9g)" % (first, o_seq[first][0], g_seq[first][0]),
                 worst, oph, gph, abs(oph - gph), "" if first is None else "  values at #%d: %.12g vs %.12g" % (first - 1, o_seq[first - 1][1], g_seq[first - 1][1])))


if __name__ == "__main__":
    main()
