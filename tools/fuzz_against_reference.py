#!/usr/bin/env python3
"""Differential run of the host layer against the reference's own code (oracle/_ref/ref_harness; only where /root/reference
exists): random corner-case MAFs (tests/golden/make_ref_vectors.py's generator with other seeds) through the reference's
reader / power-track loop and through `oracle_cli dump-alignments` / `build-tracks`, compared byte for byte; the smoothing pass on random
tracks and parameters; and build-tracks --ranks 2 / 3 / 5 on the same MAFs against the single process.

  python tools/fuzz_against_reference.py [first_seed] [n_seeds]
"""
import os, subprocess, sys, tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_ref_vectors as mk

CLI = os.path.join(ROOT, "oracle", "_build", "oracle_cli")


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "all", "ref"])
    birds = os.path.join(mk.MODELS, "53birds")
    labels = mk.labels_of(birds)
    work = tempfile.mkdtemp()
    bad = 0
    for seed in range(first, first + n):
        rng = np.random.default_rng(seed)
        style = dict(blocks=int(rng.integers(60, 400)), cols_max=int(rng.choice([2, 4, 30, 300, 3000])), skip_prob=float(rng.choice([0.0, 0.02, 0.25, 0.9])),
                     absent_prob=float(rng.choice([0.2, 0.7, 0.97])), start=int(rng.choice([0, 5, 999_400, 999_990, 1_999_000, 2_999_999 - 4000])),
                     chrom_len=int(rng.choice([4_000_000, 250_000_000, 2**31 - 1])), ref_only_prob=float(rng.choice([0.0, 0.5, 1.0])))
        # (chromosome lengths stay within int: the reference parses the numeric MAF fields with atoi(), beyond that they wrap there)
        for concat in (True, False):
            path = os.path.join(work, "f.maf")
            open(path, "w").write(mk.corner_maf(rng, labels, concat, **style))
            want = mk.harness("alignments", birds, path, 1, int(concat))
            for jobs in ((1, 3, 11) if concat else (1,)):
                got = subprocess.check_output([CLI, "dump-alignments", "--jobs", str(jobs)] + ([] if concat else ["--blocks", "1"]) + [birds, path], text=True)
                if got != want:
                    bad += 1
                    keep = os.path.join("/tmp", "fuzz_seed%d_%d.maf" % (seed, concat))
                    open(keep, "w").write(open(path).read())
                    print("seed %d concat %d jobs %d: alignments differ (%s)" % (seed, concat, jobs, keep), flush=True)
            if concat:
                mk.harness("power", birds, path, 1, os.path.join(work, "p.wig"))
                subprocess.check_call([CLI, "build-tracks", "--output", os.path.join(work, "o"), "--output-raw-phylo", "0", birds, path], stderr=subprocess.DEVNULL)
                if open(os.path.join(work, "p.wig")).read() != open(os.path.join(work, "o", "PhyloCSFpower.wig")).read():
                    bad += 1
                    print("seed %d: power tracks differ" % seed, flush=True)
    # the smoothing pass and the regions with random parameters and tracks
    for seed in range(first, first + n):
        rng = np.random.default_rng(10_000 + seed)
        w = rng.dirichlet([2, 2, 1])
        prm = [float(10 ** rng.uniform(-4, -1)), float(rng.uniform(5, 200)), *w.tolist(), *[float(10 ** rng.uniform(1, 4.5)) for _ in range(3)]]
        ps = ",".join("%.17g" % x for x in prm)
        din, dref, dmine = (os.path.join(work, "h%d%s" % (seed, k)) for k in "irm")
        for d in (din, dref, dmine):
            os.makedirs(d)
        for t in mk.TAGS:
            lines, pos = [], int(rng.integers(1, 10**6))
            for r in range(int(rng.integers(1, 12))):
                m = int(rng.integers(1, 600))
                vals = rng.normal(rng.uniform(-30, 5), rng.uniform(1, 40), size=m)
                if rng.random() < 0.5:
                    a = int(rng.integers(0, m))
                    vals[a:a + int(rng.integers(1, 80))] += rng.uniform(10, 300)
                for k, v in enumerate(vals):
                    if k == 0 or rng.random() < 0.01:
                        lines.append("fixedStep chrom=chr%d start=%d step=3 span=3" % (seed % 3, pos + 3 * k))
                    x = ("%.3f" % float(np.float32(v))).rstrip("0")
                    lines.append(x + "0" if x.endswith(".") else x)
                pos += 3 * m + (0 if rng.random() < 0.1 else int(rng.integers(1, 99)))
            for d in (din, dmine):
                open(os.path.join(d, "PhyloCSFRaw%s.wig" % t), "w").write("\n".join(lines) + "\n")
        mk.harness("smooth", ps, din, dref, 1)
        subprocess.check_call([CLI, "smooth-tracks", "--output-phylo", "1", "--output-regions", "1", "--hmm-params", ps, dmine])
        for t in mk.TAGS:
            for e in (".wig", "Regions.bed"):
                if open(os.path.join(dref, "PhyloCSF%s%s" % (t, e))).read() != open(os.path.join(dmine, "PhyloCSF%s%s" % (t, e))).read():
                    bad += 1
                    print("seed %d: PhyloCSF%s%s differs (%s)" % (seed, t, e, work), flush=True)
    # one process per rank (build-tracks --ranks N, the test command line's file-based gather): the same corner-case MAFs cut into 2 / 3 / 5
    # byte ranges -- break points, discontinuities and near-empty ranks included -- against the single process
    env = dict(os.environ, PCSF_ORACLE_BACKEND_FAKE="1")
    for seed in range(first, first + n):
        rng = np.random.default_rng(seed)
        style = dict(blocks=int(rng.integers(60, 400)), cols_max=int(rng.choice([2, 4, 30, 300, 3000])), skip_prob=float(rng.choice([0.0, 0.02, 0.25, 0.9])),
                     absent_prob=float(rng.choice([0.2, 0.7, 0.97])), start=int(rng.choice([0, 5, 999_400, 999_990, 1_999_000, 2_999_999 - 4000])),
                     chrom_len=int(rng.choice([4_000_000, 250_000_000, 2**31 - 1])), ref_only_prob=float(rng.choice([0.0, 0.5, 1.0])))
        path = os.path.join(work, "r.maf")
        open(path, "w").write(mk.corner_maf(rng, labels, True, **style))
        subprocess.check_call([CLI, "build-tracks", "--output", os.path.join(work, "plain"), birds, path], stderr=subprocess.DEVNULL, env=env)
        for ranks in (2, 3, 5):
            out = os.path.join(work, "ranks%d" % ranks)
            extra = ["--gather-root-text", "1"] if ranks == 3 else []
            subprocess.check_call([CLI, "build-tracks", "--output", out, "--ranks", str(ranks), "--window-bytes", "70000", "--batch-sites", "3000"] + extra + [birds, path],
                                  stderr=subprocess.DEVNULL, env=env)
            for name in os.listdir(os.path.join(work, "plain")):
                if open(os.path.join(work, "plain", name), "rb").read() != open(os.path.join(out, name), "rb").read():
                    bad += 1
                    print("seed %d, %d ranks: %s differs from the single process's" % (seed, ranks, name), flush=True)
    print("%d seeds, %d differences" % (n, bad))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
