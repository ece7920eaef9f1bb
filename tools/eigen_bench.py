#!/usr/bin/env python3
"""eigen_omega_kernel alone: cold starts, warm starts from a neighbouring kappa, warm starts from anywhere (run under
rocprofv3 --kernel-trace to get the kernel durations; needs a GPU)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import phylocsfpp_amd as pa
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ctx = pa.load_model(os.path.join(ROOT, "tests/golden/models/100vertebrates")).context()
rng = np.random.default_rng(3)
qs = np.empty((n, 12))
qs[:, 0] = rng.uniform(1.0, 10.0, n)
qs[:, 1] = rng.choice([1.0, 0.2], n)
qs[:, 2] = rng.choice([1.0, 0.01], n)
qs[:, 3:] = np.exp(rng.uniform(np.log(0.3), np.log(3.0), (n, 9)))
ctx.debug_eigen_omega(qs[:4])                                   # warm-up
ctx.debug_eigen_omega(qs)                                       # launch 1 (after the warm-up's): n cold starts
for name, scale in (("1 %", 0.01), ("10 %", 0.10)):
    ws = qs.copy()
    ws[:, 0] = np.clip(ws[:, 0] * (1 + scale * rng.choice([-1, 1], n)), 1.0, 10.0)
    ctx.debug_eigen_omega(qs, warm_settings=ws)                 # cold solve of ws, then warm solve of qs from it
ws = qs.copy()
ws[:, 0] = rng.uniform(1.0, 10.0, n)
ctx.debug_eigen_omega(qs, warm_settings=ws)                     # warm start from anywhere in [1, 10]
print("launch order: warm-up(4 jobs), cold, [cold, warm 1 %], [cold, warm 10 %], [cold, warm anywhere]")
