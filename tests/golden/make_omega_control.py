#!/usr/bin/env python3
"""The CONTROL behind DESIGN.md section 5's statement about OMEGA end results: the CPU oracle against ITSELF with a few-ulp perturbation
of every objective value (po_set_perturbation: the same sums, terms added in another order), on the two block lists the GPU suite
compares with the oracle -- the dense blocks of BASELINE.json configs[2] (the first ~200 blocks of the bench's synthetic MAF, 58mammals;
tests/test_gpu_config3_blocks.py) and the reference's 50 + 516 real blocks (100vertebrates; tests/test_gpu_parity_blocks.py).

If the procedure (twelve chained Brent searches, each returning its last probe) amplifies such a perturbation to the differences seen
between the HIP path and the oracle, those differences are the procedure's, not an implementation's; the rows where the control moves
are the rows where any two implementations may differ.  Writes

    tests/golden/oracle/omega_control.npz     base scores and the scores under each perturbation, per list and strategy
    profiles/r04/omega_control.txt            the statistics (rows within 1e-6, median, rows beyond the reference's tolerances)

    python tests/golden/make_omega_control.py      # ~25 min on 8 cores

tests/test_oracle_golden.py re-computes a sample of rows with the live oracle (stale file -> CPU suite fails)."""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, ROOT)
import maf_blocks
import oracle_binding as ob

ALIASES = os.path.join(HERE, "models", "species_aliases.tsv")
PERTURBATIONS = (1, 2, 4, 7)     # bits of po_set_perturbation: message sums reversed, root sum reversed, site sums reversed, all three
IMPLEMENTATION_SIZED = (8, 16)   # every site's ln z moved by <= 1.1e-13 relative: two independent draws (round 5, the whole-file MLE control)
PREFIX_BYTES = 3_600_000         # tests/test_gpu_config3_blocks.py


def config3_prefix(path):
    import gen_synthetic_maf as gen
    full = path + ".full"
    gen.generate(full, bases=1_000_000, seed=20261002)
    data = open(full, "rb").read(PREFIX_BYTES)
    os.remove(full)
    with open(path, "wb") as f:
        f.write(data[:data.rfind(b"\n\n") + 2])


def cli_scores(strategy, maf, out_dir, perturb):
    os.makedirs(out_dir, exist_ok=True)
    cli = os.path.join(ROOT, "oracle", "_build", "oracle_cli")
    env = dict(os.environ, PO_PERTURB=str(perturb))
    subprocess.check_call([cli, "score-msa", "--strategy", strategy, "--comp-anc", "1" if strategy == "mle" else "0", "--output", out_dir,
                           os.path.join(HERE, "models", "58mammals"), maf], env=env, stderr=subprocess.DEVNULL)
    rows = [l.rstrip("\n").split("\t") for l in open(os.path.join(out_dir, os.path.basename(maf) + ".scores")) if not l.startswith("#")][1:]
    return np.array([float(r[4]) for r in rows], dtype=np.float64)


def cli_scores_both(strategy, maf, out_dir, perturb):
    """Both score columns (PhyloCSF score, ancestral score) of every row."""
    os.makedirs(out_dir, exist_ok=True)
    cli = os.path.join(ROOT, "oracle", "_build", "oracle_cli")
    env = dict(os.environ, PO_PERTURB=str(perturb))
    subprocess.check_call([cli, "score-msa", "--strategy", strategy, "--comp-anc", "1", "--output", out_dir,
                           os.path.join(HERE, "models", "58mammals"), maf], env=env, stderr=subprocess.DEVNULL)
    rows = [l.rstrip("\n").split("\t") for l in open(os.path.join(out_dir, os.path.basename(maf) + ".scores")) if not l.startswith("#")][1:]
    return np.array([[float(r[4]), float(r[5])] for r in rows], dtype=np.float64)


def mle_full():
    """the MLE control on ALL blocks of the bench's MAF (BASELINE.json configs[2]: 5,088 blocks), both score
    columns -- the driver's bench line had shown a row at 0.040 > the reference's own MLE tolerance (test/tests.sh:41: d^2 <= 0.001)
    beyond the 197-block prefix the first control covered.  Writes tests/golden/oracle/mle_control_full.npz (float32 text values are
    exact in float64; ~400 KB) and profiles/r05/mle_control_full.txt."""
    import tempfile
    import gen_synthetic_maf as gen
    tol = 0.001 ** 0.5
    report = ["# tests/golden/make_omega_control.py --mle-full: oracle_cli score-msa --strategy mle --comp-anc 1 on the whole synthetic 1 Mb MAF",
              "# (tools/gen_synthetic_maf.py --bases 1000000 --seed 20261002, 58mammals), against itself under po_set_perturbation 1, 2, 4, 7",
              "# tolerance: the reference's own, test/tests.sh:41 (d^2 <= 0.001, i.e. |d| <= 0.0316)", ""]
    out_npz = os.path.join(HERE, "oracle", "mle_control_full.npz")
    res = dict(np.load(out_npz)) if os.path.exists(out_npz) and "--extend" in sys.argv else {}   # --extend: only the perturbations not in the file yet
    with tempfile.TemporaryDirectory() as tmp:
        maf = os.path.join(tmp, "synthetic_1mb.maf")
        gen.generate(maf, bases=1_000_000, seed=20261002)
        if "base" not in res:
            res["base"] = cli_scores_both("mle", maf, os.path.join(tmp, "mle_0"), 0)
        base = res["base"]
        print("base: %d rows" % len(base), flush=True)
        for b in PERTURBATIONS + IMPLEMENTATION_SIZED:
            if "p%d" % b not in res:
                res["p%d" % b] = cli_scores_both("mle", maf, os.path.join(tmp, "mle_%d" % b), b)
            print("perturbation %d done" % b, flush=True)
    for col, name in ((0, "PhyloCSF score"), (1, "ancestral score")):
        spread = describe("configs[2], all blocks, 58mammals, MLE, %s; perturbations of a few ulp (sums in another order)" % name, base[:, col],
                          {b: res["p%d" % b][:, col] for b in PERTURBATIONS}, tol, report)
        report.append("  rows whose spread is >= 0.01: %s" % [int(i) for i in np.flatnonzero(spread >= 0.01)])
        spread2 = describe("the same; perturbations of the size measured between the HIP path and the oracle (every site's ln z moved by <= 1.1e-13 relative)",
                           base[:, col], {b: res["p%d" % b][:, col] for b in IMPLEMENTATION_SIZED}, tol, report)
        report.append("  rows whose spread is >= 0.01: %s" % [int(i) for i in np.flatnonzero(spread2 >= 0.01)])
        report.append("")
    np.savez_compressed(out_npz, **res)
    os.makedirs(os.path.join(ROOT, "profiles", "r05"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "r05", "mle_control_full.txt"), "w") as f:
        f.write("\n".join(report) + "\n")
    print("\n".join(report))


def describe(name, base, pert, tol, out):
    ok = np.isfinite(base)
    out.append("%s: %d rows (%d finite)" % (name, len(base), int(ok.sum())))
    spread = np.zeros(len(base))
    for bits, p in pert.items():
        assert np.array_equal(np.isnan(p), np.isnan(base))
        d = np.abs(p[ok] - base[ok])
        rel = d / np.maximum(1e-6, np.abs(base[ok]))
        spread[ok] = np.maximum(spread[ok], d)
        out.append("  perturbation %d vs the oracle itself: within 1e-6 relative %4d (%.0f %%), median |d| %.3g, 90 %% %.3g, max %.3g, rows beyond %.3f: %d"
                   % (bits, int((rel <= 1e-6).sum()), 100.0 * (rel <= 1e-6).mean(), np.median(d), np.percentile(d, 90), d.max(), tol, int((d > tol).sum())))
    big = np.flatnonzero(spread > tol)
    out.append("  rows that moved beyond %.3f under at least one perturbation: %s" % (tol, list(big)))
    out.append("  rows that moved by more than 1e-6 relative under at least one: %d of %d" % (int((spread[ok] / np.maximum(1e-6, np.abs(base[ok])) > 1e-6).sum()), int(ok.sum())))
    return spread


def main():
    import tempfile
    res, report = {}, ["# tests/golden/make_omega_control.py: the oracle against itself under few-ulp perturbations of the objective (po_set_perturbation)",
                       "# tolerances: the reference's own, test/tests.sh:41 (MLE: d^2 <= 0.001) and :46 (OMEGA: d^2 <= 0.1)", ""]
    with tempfile.TemporaryDirectory() as tmp:
        maf = os.path.join(tmp, "prefix.maf")
        config3_prefix(maf)
        for strategy, tol in (("mle", 0.001 ** 0.5), ("omega", 0.1 ** 0.5)):
            base = cli_scores(strategy, maf, os.path.join(tmp, strategy + "_0"), 0)
            pert = {b: cli_scores(strategy, maf, os.path.join(tmp, "%s_%d" % (strategy, b)), b) for b in PERTURBATIONS}
            res["config3_%s_base" % strategy] = base
            for b, p in pert.items():
                res["config3_%s_p%d" % (strategy, b)] = p
            describe("configs[2] prefix, 58mammals, %s" % strategy.upper(), base, pert, tol, report)
            report.append("")
            print("\n".join(report[-8:]), flush=True)
    m = ob.OracleModel(os.path.join(HERE, "models", "100vertebrates"))
    for tag, name in (("small", "chr22.50alignments.maf"), ("medium", "chr22.516alignments.maf.gz")):
        cod, off = maf_blocks.block_codons(os.path.join(HERE, "score-msa", name), m.labels, ALIASES)
        base = np.load(os.path.join(HERE, "oracle", tag + "_omega.npy")).astype(np.float64)
        pert = {}
        for b in PERTURBATIONS:
            ob.lib().po_set_perturbation(b)
            pert[b] = m.run_blocks(ob.OMEGA, cod, off, compute_anc=False)[0].astype(np.float64)
            res["%s_omega_p%d" % (tag, b)] = pert[b]
        ob.lib().po_set_perturbation(0)
        describe("the reference's %d real blocks, 100vertebrates, OMEGA" % (len(off) - 1), base, pert, 0.1 ** 0.5, report)
        report.append("")
        print("\n".join(report[-8:]), flush=True)
    np.savez_compressed(os.path.join(HERE, "oracle", "omega_control.npz"), **res)
    os.makedirs(os.path.join(ROOT, "profiles", "r04"), exist_ok=True)
    with open(os.path.join(ROOT, "profiles", "r04", "omega_control.txt"), "w") as f:
        f.write("\n".join(report) + "\n")


if __name__ == "__main__":
    if "--mle-full" in sys.argv:
        mle_full()
    else:
        main()
