"""On the GPU box: subtree tables one level further (PCSF_TABLE_LEAVES=4, nodes over two cherries or over a triple and a leaf: 65^4 rows each):
context creation time, memory, bit-identity with the default form, kernel time on the bench's columns.  python tools/quad_probe.py [model]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import phylocsfpp_amd as pa
from phylocsfpp_amd import synth

name = sys.argv[1] if len(sys.argv) > 1 else "58mammals"
m = pa.load_model(name)
n = 2_000_000
rng = np.random.default_rng(1)
pool = synth.synthetic_codon_columns(m.child1, m.child2, m.branch_len, m.S_c, m.pi_c, m.S_nc, m.pi_nc, 131072, seed=20261002)
cod = np.ascontiguousarray(pool[:, rng.integers(0, pool.shape[1], n)])
cod[:, 1000:1100] = 64
cod[::3, 2000:4000] = 64
d = torch.from_numpy(cod).cuda()
res = {}
for tag, env in (("triples (default)", {}), ("quads", {"PCSF_TABLE_LEAVES": "4", "PCSF_TABLE_BYTES": str(200 << 30)})):
    for k in ("PCSF_TABLE_LEAVES", "PCSF_TABLE_BYTES"):
        os.environ.pop(k, None)
    os.environ.update(env)
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    t0 = time.time()
    c = m.context(device=0)
    t1 = time.time()
    info = c.tracks_program_info()
    out = torch.empty(n, dtype=torch.float64, device="cuda")
    c.score_sites_fixed_torch(d, out=out)
    c.set_timing(True)
    ms = []
    for _ in range(5):
        c.score_sites_fixed_torch(d, out=out)
        ms.append(c.last_kernel_ms())
    c.set_timing(False)
    res[tag] = out.cpu().numpy()
    print("%-18s context %.3f s, %.1f GB of device memory, %s; kernel %.3f ms = %.1f M codon-frames/s" %
          (tag, t1 - t0, (free0 - torch.cuda.mem_get_info()[0]) / 1e9, info, np.mean(ms), n / np.mean(ms) / 1e3), flush=True)
    c.close()
    del out
a, b = res["triples (default)"], res["quads"]
print("bit-identical:", bool(np.array_equal(a, b, equal_nan=True)), " finite:", int(np.isfinite(a).sum()), "of", n)
