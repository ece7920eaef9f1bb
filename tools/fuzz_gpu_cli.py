#!/usr/bin/env python3
"""Differential run of `phylocsf-amd build-tracks` (the HIP path) against the oracle-backed driver (`oracle_cli build-tracks`, same host
code, CPU scoring) on random corner-case MAFs -- tests/golden/make_ref_vectors.py's generator (short blocks, reference-gap columns, missing
rows, unknown species, lower case, N, '.', two chromosomes, the 1 Mb break point) with other seeds and several trees -- in every form of
tracks mode: subtree tables (default: cherries + triples), cherries only, the full program.  The three GPU forms must write the same BYTES;
against the CPU driver the seven wig files are compared line by line (three decimals of text; lines that differ only in the sign of a
printed zero are counted apart).  Needs a GPU; nothing of /root/reference.

  python tools/fuzz_gpu_cli.py [first_seed] [n_seeds] [--tracks-only | --score-msa-only]
"""
import filecmp, os, subprocess, sys, tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import make_ref_vectors as mk

ORACLE = os.path.join(ROOT, "oracle", "_build", "oracle_cli")
HIP = os.path.join(ROOT, "phylocsfpp_amd", "host", "phylocsf-amd")
MODELS = os.path.join(ROOT, "phylocsfpp_amd", "models")
WIGS = ["PhyloCSFpower.wig"] + ["PhyloCSFRaw%s%d.wig" % (s, f) for s in "+-" for f in (1, 2, 3)]
FORMS = (("triples", {}), ("cherries", {"PCSF_TABLE_LEAVES": "2"}), ("full", {"PCSF_CHERRY_TABLES": "0"}))
TREES = ("58mammals", "53birds", "7yeast", "12flies", "29mammals", "100vertebrates", "26worms", "20flies", "21mosquitoes", "23flies", "49birds")


def lines_differing(a, b, show=3):
    """(lines that differ in value or in number, lines that differ only in the sign of a printed zero)"""
    la, lb = open(a, "rb").read().split(b"\n"), open(b, "rb").read().split(b"\n")
    diff = [(i, x, y) for i, (x, y) in enumerate(zip(la, lb)) if x != y]
    zeros = [d for d in diff if {d[1], d[2]} == {b"0.0", b"-0.0"}]
    other = [d for d in diff if {d[1], d[2]} != {b"0.0", b"-0.0"}]
    for i, x, y in other[:show]:
        print("    line %d: %s | %s" % (i + 1, x.decode(), y.decode()), flush=True)
    return abs(len(la) - len(lb)) + len(other), len(zeros)


def rows_of(path):
    return [l.rstrip("\n").split("\t") for l in open(path) if not l.startswith("#")][1:]


def score_msa(first, n, work):
    """score-msa on block-mode MAFs ('-'-strand reference rows, blocks of one column to thousands): FIXED on every seed (scores are floats printed
    with six decimals: compared as text, and where text differs by value), MLE and OMEGA on every fourth within the reference's own tolerance
    (squared difference 0.001 / 0.1, test/tests.sh:41,46)."""
    bad = rows = text_diff = 0
    worst = 0.0
    fields, beyond = {}, {}
    shown = 0
    for seed in range(first, first + n):
        rng = np.random.default_rng(88_000 + seed)
        tree = TREES[seed % len(TREES)]
        prefix = os.path.join(MODELS, tree)
        labels = mk.labels_of(prefix)
        style = dict(blocks=int(rng.integers(60, 300)), cols_max=int(rng.choice([4, 30, 300, 2000])), skip_prob=0.25,
                     absent_prob=float(rng.choice([0.05, 0.3, 0.7, 0.95])), start=int(rng.choice([0, 999_400])), chrom_len=4_000_000, ref_only_prob=float(rng.choice([0.0, 0.5])))
        text = mk.corner_maf(rng, labels, False, **style)
        # the optimising strategies on a smaller file of the same style (the CPU driver's share of the run: minutes per long block)
        small = mk.corner_maf(rng, labels, False, **dict(style, blocks=min(style["blocks"], 100), cols_max=min(style["cols_max"], 300)))
        for strategy, anc, tol in (("fixed", "1", None), ("mle", "1", 0.001), ("omega", "0", 0.1)):
            if strategy != "fixed" and seed % 4:
                continue
            got = []
            for cli, name in ((HIP, "gpu"), (ORACLE, "cpu")):
                maf = os.path.join(work, "s%d_%s_%s.maf" % (seed, strategy, name))
                open(maf, "w").write(text if strategy == "fixed" else small)
                subprocess.check_call([cli, "score-msa", "--strategy", strategy, "--comp-anc", anc, prefix, maf], stderr=subprocess.DEVNULL, stdout=subprocess.DEVNULL)
                got.append(rows_of(maf + ".scores"))
            g, w = got
            if len(g) != len(w):
                bad += 1
                print("seed %d %s %s: %d rows against %d" % (seed, tree, strategy, len(g), len(w)), flush=True)
                continue
            for a, b in zip(g, w):
                rows += 1
                if a[:4] != b[:4] or a[-1] != b[-1]:
                    bad += 1
                    print("seed %d %s %s: coordinates / bls differ: %s | %s" % (seed, tree, strategy, a, b), flush=True)
                    continue
                for x, y in zip(a[4:-1], b[4:-1]):
                    if x == y:
                        continue
                    fx, fy = float(x), float(y)
                    if tol is None:
                        text_diff += 1
                        rel = abs(fx - fy) / max(1.0, abs(fy))
                        worst = max(worst, rel)
                        if not rel <= 1e-6:      # six decimals of a float: the last digit may turn
                            bad += 1
                            print("seed %d %s fixed: %s | %s" % (seed, tree, a, b), flush=True)
                    elif y != "nan":
                        fields[strategy] = fields.get(strategy, 0) + 1
                        if not (fx - fy) ** 2 <= tol:    # reported, not failed: on unrelated random rows the objectives are flat and two optimisers
                            beyond[strategy] = beyond.get(strategy, 0) + 1      # that differ in the last bits may stop apart
                            lone = float(a[-1]) == 0.0     # no second species in the block: the likelihood does not depend on the tree's scale at all
                            if lone:
                                beyond[strategy + ", of which in blocks of one species"] = beyond.get(strategy + ", of which in blocks of one species", 0) + 1
                            shown += 0 if lone else 1
                            if beyond[strategy] <= 5 or (not lone and shown <= 30):
                                print("seed %d %s %s: %s | %s" % (seed, tree, strategy, a, b), flush=True)
        print("seed %d %s score-msa %s: compared" % (seed, tree, style), flush=True)
    print("score-msa: %d rows compared; FIXED fields whose text differs %d (largest relative difference %.2e); MLE / OMEGA fields whose text differs %s, "
          "of which beyond the reference's tolerance %s" % (rows, text_diff, worst, fields, beyond))
    return bad


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    work = tempfile.mkdtemp(prefix="pcsf_fuzz_")
    bad = lines = zero_signs = 0
    for seed in range(first, first + (0 if "--score-msa-only" in sys.argv else n)):
        rng = np.random.default_rng(77_000 + seed)
        tree = TREES[seed % len(TREES)]
        prefix = os.path.join(MODELS, tree)
        labels = mk.labels_of(prefix)
        style = dict(blocks=int(rng.integers(80, 500)), cols_max=int(rng.choice([4, 30, 300, 3000])), skip_prob=float(rng.choice([0.0, 0.02, 0.25])),
                     absent_prob=float(rng.choice([0.05, 0.3, 0.7, 0.95])), start=int(rng.choice([0, 5, 999_400, 1_999_000])),
                     chrom_len=int(rng.choice([4_000_000, 250_000_000])), ref_only_prob=float(rng.choice([0.0, 0.5])))
        maf = os.path.join(work, "f%d.maf" % seed)
        open(maf, "w").write(mk.corner_maf(rng, labels, True, **style))
        out_cpu = os.path.join(work, "cpu%d" % seed)
        subprocess.check_call([ORACLE, "build-tracks", "--output", out_cpu, prefix, maf], stderr=subprocess.DEVNULL)
        outs = {}
        for form, env in FORMS:
            outs[form] = os.path.join(work, "%s%d" % (form, seed))
            subprocess.check_call([HIP, "build-tracks", "--output", outs[form], prefix, maf], stderr=subprocess.DEVNULL, env=dict(os.environ, **env))
        for w in WIGS:
            lines += open(os.path.join(out_cpu, w), "rb").read().count(b"\n")
            for form in ("cherries", "full"):
                if not filecmp.cmp(os.path.join(outs["triples"], w), os.path.join(outs[form], w), shallow=False):
                    bad += 1
                    print("seed %d %s %s: the %s form writes other bytes than the default" % (seed, tree, w, form), flush=True)
            d, z = lines_differing(os.path.join(outs["triples"], w), os.path.join(out_cpu, w))
            zero_signs += z
            if d:
                bad += 1
                print("seed %d %s %s: %d lines differ from the CPU driver's" % (seed, tree, w, d), flush=True)
        print("seed %d %s %s: compared (%d zero-sign lines so far)" % (seed, tree, style, zero_signs), flush=True)
    if "--tracks-only" not in sys.argv:
        bad += score_msa(first, n, work)
    print("%d seeds, %d wig lines per form, %d files with differences; %d lines print -0.0 on one side and 0.0 on the other (a codon with no known base in any "
          "row: both likelihoods are 1 up to rounding, and the sign of the rounding is the text)" % (n, lines, bad, zero_signs))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
