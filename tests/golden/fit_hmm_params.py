#!/usr/bin/env python3
"""Back-solve the HMM parameters behind the reference's smoothed golden tracks.

The reference's build-tracks test smooths with parameters estimated from example/galGal6_coding_exons.txt, a file
the checkout does not contain, so tests/golden/build-tracks/PhyloCSF{+,-}{1,2,3}.wig.gz cannot be regenerated
from first principles.  This script fits the eight hmm_parameter values (coding prior, mean coding length, three
non-coding weights and mean lengths) by Levenberg-Marquardt so that smoothing the golden RAW tracks reproduces the
golden SMOOTHED tracks, using the product's own smoothing pass (`oracle_cli smooth-tracks`, host code only) as the
model, and writes hmm_params_backsolved.json.

Result (committed): rms 0.0065, max 0.10 log-odds units over the 174,648 values; an UNCONSTRAINED 4-state HMM
(15 free parameters) fits no better (rms 0.00645) and the residual is a slow random walk along each run that
does not correlate with the raw score, i.e. the goldens were smoothed from raw scores that differ by a few 1e-3
from the golden raw tracks -- not from a different model.  Parity with these golden FILES is therefore pinned only to
that tolerance; the smoothing pass and the estimation themselves are pinned byte for byte on the reference's own code
(oracle/ref_harness.cpp, tests/test_reference_vectors.py), which turns the golden raw tracks + these parameters into
exactly the bytes this repository's pass writes.

  python tests/golden/fit_hmm_params.py          # ~2 minutes
"""
import gzip, json, os, subprocess, sys, tempfile

import numpy as np
from scipy.optimize import least_squares

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CLI = os.path.join(ROOT, "oracle", "_build", "oracle_cli")
TAGS = [s + str(f) for s in "+-" for f in (1, 2, 3)]


def values(path):
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rt") as f:
        return np.array([float(x) for x in f if x[0] != "f"])


def main():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])
    work = tempfile.mkdtemp()
    for t in TAGS:
        with gzip.open(os.path.join(HERE, "build-tracks", "PhyloCSFRaw%s.wig.gz" % t), "rb") as fi, open(os.path.join(work, "PhyloCSFRaw%s.wig" % t), "wb") as fo:
            fo.write(fi.read())
    gold = np.concatenate([values(os.path.join(HERE, "build-tracks", "PhyloCSF%s.wig.gz" % t)) for t in TAGS])
    sel = np.abs(gold) < 14.9                     # saturated values carry no gradient

    def params(theta):
        p = 1 / (1 + np.exp(-theta[0]))
        w = np.exp(np.array([theta[2], theta[3], 0.0]))
        w /= w.sum()
        return [p, np.exp(theta[1])] + w.tolist() + np.exp(theta[4:7]).tolist()

    def run(theta):
        subprocess.check_call([CLI, "smooth-tracks", "--hmm-params", ",".join("%.17g" % x for x in params(theta)), work])
        return np.concatenate([values(os.path.join(work, "PhyloCSF%s.wig" % t)) for t in TAGS])

    # start: the best of several earlier runs (from typical vertebrate values -- 0.3 % coding per frame, 60-codon
    # exons, gaps of 1 k / 27 k / 33 codons -- LM stops in a slightly worse optimum, rms 0.0079)
    p0, l0, w0, n0 = 0.0054925759, 58.493163, [0.50641728, 0.45164511, 0.04193761], [570.86305, 21970.849, 105.30739]
    theta = np.array([np.log(p0 / (1 - p0)), np.log(l0), np.log(w0[0] / w0[2]), np.log(w0[1] / w0[2])] + [np.log(x) for x in n0])
    for step in (1e-3, 1e-4, 1e-5):
        theta = least_squares(lambda t: (run(t) - gold)[sel], theta, method="lm", xtol=1e-14, ftol=1e-14, gtol=1e-14, max_nfev=400, diff_step=step).x
        out = run(theta)
        r = (out - gold)[sel]
        print("rms %.6f  max %.4f  identical text %.4f" % (np.sqrt((r ** 2).mean()), np.abs(out - gold).max(), (out == gold).mean()), flush=True)
    prm = params(theta)
    json.dump({"hmm_params": prm, "order": ["coding_prior", "coding_length_in_codons", "non_coding_weights[3]", "non_coding_lengths_in_codons[3]"],
               "rms": float(np.sqrt((r ** 2).mean())), "max_abs": float(np.abs(out - gold).max()), "values": int(gold.size)},
              open(os.path.join(HERE, "build-tracks", "hmm_params_backsolved.json"), "w"), indent=1)
    print(prm)


if __name__ == "__main__":
    main()
