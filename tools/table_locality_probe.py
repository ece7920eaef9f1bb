"""On the GPU box: how much of the table form's time is row look-ups that miss the caches?  prune_fixed on 2 M codon-frames of (a) the bench's
evolved columns, (b) fully conserved columns (every species the same codon: every pair / triple row hot), (c) unrelated random codons (every
row cold), under the table form and the full program.  python tools/table_locality_probe.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import phylocsfpp_amd as pa
from phylocsfpp_amd import synth

m = pa.load_model("58mammals")
n = 2_000_000
rng = np.random.default_rng(1)
pool = synth.synthetic_codon_columns(m.child1, m.child2, m.branch_len, m.S_c, m.pi_c, m.S_nc, m.pi_nc, 131072, seed=20261002)
cols = {"evolved (the bench's pool)": pool[:, rng.integers(0, pool.shape[1], n)],
        "fully conserved": np.repeat(rng.integers(0, 61, n, dtype=np.uint8)[None, :], m.nl, axis=0),
        "unrelated random codons": rng.integers(0, 64, size=(m.nl, n), dtype=np.uint8)}
for form, env in (("subtree tables", None), ("full program", "0")):
    if env is None:
        os.environ.pop("PCSF_CHERRY_TABLES", None)
    else:
        os.environ["PCSF_CHERRY_TABLES"] = env
    c = m.context(device=0)
    for name, cod in cols.items():
        d = torch.from_numpy(np.ascontiguousarray(cod)).cuda()
        out = torch.empty(n, dtype=torch.float64, device="cuda")
        c.score_sites_fixed_torch(d, out=out)
        c.set_timing(True)
        ms = []
        for _ in range(5):
            c.score_sites_fixed_torch(d, out=out)
            ms.append(c.last_kernel_ms())
        c.set_timing(False)
        print("%-16s %-28s %.3f ms  %.1f M codon-frames/s" % (form, name, np.mean(ms), n / np.mean(ms) / 1e3))
        del d, out
    c.close()
