import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import maf_blocks, oracle_binding as ob, phylocsfpp_amd as pa
G = os.path.join(ROOT, "tests/golden")
m = ob.OracleModel(G + "/models/100vertebrates")
cod, off = maf_blocks.block_codons(G + "/score-msa/chr22.50alignments.maf", m.labels, G + "/models/species_aliases.tsv")
blk = int(sys.argv[1]) if len(sys.argv) > 1 else 4
sub = np.ascontiguousarray(cod[:, int(off[blk]):int(off[blk + 1])])
so = np.array([0, sub.shape[1]], dtype=np.uint64)
if sys.argv[2] == "gpu":
    os.environ["PCSF_TRACE_BLOCK"] = "0"
    ctx = pa.ScoringContext(m.child1, m.child2, m.branch_len, m.S_c, m.pi_c, m.S_nc, m.pi_nc)
    print(ctx.score_blocks(pa.OMEGA, sub, so, compute_anc=False))
else:
    os.environ["PO_TRACE"] = "1"
    print(m.run(ob.OMEGA, sub, compute_anc=False))
