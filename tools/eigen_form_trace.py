#!/usr/bin/env python3
"""The OMEGA searches of ONE block under the eigen-solver's forms, evaluation by evaluation (PCSF_BLOCK_ENGINE=host PCSF_TRACE_BLOCK):
python3 tools/eigen_form_trace.py [block] [kernel] -- the 700-site edge block of tests/test_gpu_optimizer.py by default.  Prints where the
two traces part and the size of the objective's difference before that."""
import os, sys, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import phylocsfpp_amd as pa
    import oracle_binding as ob
    from phylocsfpp_amd import synth
    GOLD = os.path.join(ROOT, "tests", "golden")
    m = ob.OracleModel(os.path.join(GOLD, "models", "53birds"))
    lens = [0, 1, 17, 0, 700, 5, 40]
    cols = synth.synthetic_codon_columns(m.child1, m.child2, m.branch_len, m.S_c, m.pi_c, m.S_nc, m.pi_nc, sum(lens), seed=5, dropout=0.2, gap_rate=0.05)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    cols[:, int(off[5]):int(off[6])] = 17
    ctx = pa.load_model(os.path.join(GOLD, "models", "53birds")).context()
    ctx.set_block_kernel(int(sys.argv[2]))
    ph = ctx.score_blocks(pa.OMEGA, cols, off, compute_anc=False)[0]
    print("RESULT", " ".join("%.6f" % x for x in ph))
    sys.exit(0)
block = sys.argv[1] if len(sys.argv) > 1 else "4"
kernel = sys.argv[2] if len(sys.argv) > 2 else "2"
traces = {}
for form in ("1", "2"):
    env = dict(os.environ, PCSF_EIGEN_FORM=form, PCSF_BLOCK_ENGINE="host", PCSF_TRACE_BLOCK=block)
    r = subprocess.run([sys.executable, __file__, "child", kernel], env=env, capture_output=True, text=True)
    rows = [l.split() for l in r.stderr.splitlines() if l.startswith("TRACE omega stage")]
    traces[form] = [(int(t[3]), int(t[5]), float(t[7]), float(t[9])) for t in rows]
    best = {}
    for t in rows:
        key, val = (int(t[3]), int(t[5])), float(t[9]) + float(t[11])
        if key not in best or val > best[key][0]:
            best[key] = (val, float(t[7]))
        best[key] = best[key] + ()
    count = {}
    for t in rows:
        count[(int(t[3]), int(t[5]))] = count.get((int(t[3]), int(t[5])), 0) + 1
    for key in sorted(best):
        print("   stage %d kind %d: %3d evaluations, best ln z + prior %.9f at x = %.12g" % (key[0], key[1], count[key], best[key][0], best[key][1]))
    print("form", form, len(traces[form]), "evaluations;", [l for l in r.stdout.splitlines() if l.startswith("RESULT")])
if os.environ.get("TRACE_STAGES"):
    for form in ("1", "2"):
        for st in [int(x) for x in os.environ["TRACE_STAGES"].split(",")]:
            print("form %s stage %d, in order:" % (form, st))
            for t in traces[form]:
                if t[0] == st:
                    print("     kind %d x %.15g ln z %.9f" % (t[1], t[2], t[3]))
a, b = traces["1"], traces["2"]
worst = 0.0
for i, (p, q) in enumerate(zip(a, b)):
    if p[:3] != q[:3]:
        print("the traces part at evaluation %d: form 1 stage %d kind %d x %.17g lnz %.17g | form 2 stage %d kind %d x %.17g lnz %.17g" % (i, *p, *q))
        for k in range(max(0, i - 6), min(len(a), len(b), i + 3)):
            print("   ", k, a[k], b[k])
        break
    worst = max(worst, abs(p[3] - q[3]) / max(1.0, abs(p[3])))
print("largest relative difference of ln z at a shared point before that: %.3g" % worst)
