"""The smoothing recurrences on ONE long run (a contiguous chromosome): how many chunks have to be evaluated again at the default chunk
length / burn-in, and the time of the call.  python tools/hmm_long_run.py [values] [constant]   (runs on the GPU box)"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import phylocsfpp_amd as pa
import hmm_oracle as ho
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000000
prm = json.load(open(os.path.join(ROOT, "tests/golden/build-tracks/hmm_params_backsolved.json")))["hmm_params"]
init, trans = ho.make_hmm(prm)
rng = np.random.default_rng(5)
scores = np.zeros(n) if len(sys.argv) > 2 else np.round(rng.normal(-5.0, 8.0, size=n), 3)
emit = np.power(10.0, scores / 10.0)
ctx = pa.load_model("53birds").context()
offs = np.array([0, n], dtype=np.uint64)
ctx.hmm_smooth_runs(init, trans, emit[:1000], np.array([0, 1000], dtype=np.uint64))
for env in ({}, {"PCSF_HMM_CHUNK": "4096", "PCSF_HMM_BURN_IN": "4096"}, {"PCSF_HMM_CHUNK": str(n), "PCSF_HMM_BURN_IN": "1"}):
    for k in ("PCSF_HMM_CHUNK", "PCSF_HMM_BURN_IN"):
        os.environ.pop(k, None)
    os.environ.update(env)
    t = time.time()
    post, back, last, st = ctx.hmm_smooth_runs(init, trans, emit, offs)
    dt = time.time() - t
    print("%s: %d values, %d chunks; evaluated again: forward %d (%d passes), backward %d (%d), viterbi %d (%d); %.3f s; checksum %.17g" %
          (env or "default (chunk 32768, burn-in 32768)", n, st[0], st[1], st[2], st[3], st[4], st[5], st[6], dt, float(post.sum())), flush=True)
