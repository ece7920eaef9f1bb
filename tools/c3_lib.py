"""score_blocks on the configs[2] MAF through the Python binding (PCSF_LIBRARY selects a build): python tools/c3_lib.py mle|omega"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import phylocsfpp_amd as pa
import maf_blocks
maf = "/tmp/config3_work/syn1mb.maf"
if not os.path.exists(maf):
    os.makedirs(os.path.dirname(maf), exist_ok=True)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_synthetic_maf as gen
    gen.generate(maf, bases=1_000_000, seed=20261002)
m = pa.load_model("58mammals")
npz = "/tmp/config3_work/blocks.npz"
if os.path.exists(npz):
    z = np.load(npz); cod, offs = z["cod"], z["offs"]
else:
    cod, offs = maf_blocks.block_codons(maf, m.labels, os.path.join(ROOT, "phylocsfpp_amd/models/species_aliases.tsv"))
    np.savez(npz, cod=cod, offs=offs)
ctx = m.context()
algo = pa.OMEGA if (len(sys.argv) > 1 and sys.argv[1] == "omega") else pa.MLE
ctx.score_blocks(algo, cod[:, :int(offs[4])], offs[:5], compute_anc=False)
for rep in range(2):
    t = time.time()
    ph, an = ctx.score_blocks(algo, cod, offs, compute_anc=(algo == pa.MLE))
    print("%s: %d blocks %.3f s" % (sys.argv[1] if len(sys.argv) > 1 else "mle", len(offs) - 1, time.time() - t), flush=True)
