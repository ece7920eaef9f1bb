"""score-msa block strategies on the GPU: python tools/bench_blocks.py [maf] [max_blocks] [replicas] [nocpu]
Times ScoringContext.score_blocks for FIXED / MLE / OMEGA over the blocks of a MAF (default: the reference's
516-block medium test file), and the oracle on the host CPUs over a sample of the same blocks."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")   # OMEGA drives four streams per device at once; before the HIP runtime starts
import phylocsfpp_amd as pa
import oracle_binding as ob
import maf_blocks

maf = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1] != "-" else os.path.join(ROOT, "tests/golden/score-msa/chr22.516alignments.maf.gz")
max_blocks = (int(sys.argv[2]) or None) if len(sys.argv) > 2 else None
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
with_cpu = not (len(sys.argv) > 4 and sys.argv[4] == "nocpu")   # the oracle sample costs ~2 minutes of one host core
m = ob.OracleModel(os.path.join(ROOT, "tests/golden/models/100vertebrates"))
cod, offs = maf_blocks.block_codons(maf, m.labels, os.path.join(ROOT, "tests/golden/models/species_aliases.tsv"), max_blocks)
if reps > 1:   # replicate the block set to see how the lock-step batch scales
    n0 = offs[-1]
    cod = np.ascontiguousarray(np.tile(cod, (1, reps)))
    offs = np.concatenate([[0]] + [offs[1:] + r * n0 for r in range(reps)]).astype(np.uint64)
nb = len(offs) - 1
print("%d blocks, %d codon columns (median %d per block)" % (nb, offs[-1], int(np.median(np.diff(offs)))))
ctx = pa.load_model(os.path.join(ROOT, "tests/golden/models/100vertebrates")).context()
ctx.prepare_block_scoring(pa.MLE if hasattr(pa, "MLE") else 0)   # the engines' device queues are created while the file is read
for name, algo in (("FIXED", pa.FIXED), ("MLE", pa.MLE), ("OMEGA", pa.OMEGA)):
    ctx.score_blocks(algo, cod[:, :int(offs[min(4, nb)])], offs[:min(4, nb) + 1], compute_anc=(algo != pa.OMEGA))  # warm-up
    t = time.time()
    ph, an = ctx.score_blocks(algo, cod, offs, compute_anc=(algo != pa.OMEGA))
    dt = time.time() - t
    print("GPU %-6s %8.3f s  %9.1f blocks/s  %10.0f codon columns/s   (finite %d)" % (name, dt, nb / dt, offs[-1] / dt, int(np.isfinite(ph).sum())), flush=True)
if not with_cpu:
    sys.exit(0)
# CPU oracle on a bounded sample
ns = min(nb, 32)
sub = cod[:, :int(offs[ns])]
for name, algo in (("FIXED", 1), ("MLE", 0), ("OMEGA", 2)):
    t = time.time()
    for b in range(ns if algo != 1 else ns):
        m.run(algo, np.ascontiguousarray(cod[:, int(offs[b]):int(offs[b + 1])]), compute_anc=(algo != 2))
    dt = time.time() - t
    print("CPU oracle %-6s (1 thread, %d blocks) %8.3f s  %9.2f blocks/s" % (name, ns, dt, ns / dt), flush=True)
