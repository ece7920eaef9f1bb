"""CPU: the smoothing pass of build-tracks (host code, phylocsfpp_amd/host/hmm.hpp) against the plain-Python
restatement in tests/hmm_oracle.py, and against the reference's smoothed golden tracks with the back-solved HMM
parameters (tolerance only -- see tests/golden/fit_hmm_params.py for why they cannot be pinned exactly).
The byte-for-byte pins on the reference's own smoothing / estimation code are in tests/test_reference_vectors.py."""
import gzip
import json
import os
import re
import subprocess

import numpy as np
import pytest

import hmm_oracle as ho

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden", "build-tracks")
CLI = os.path.join(ROOT, "oracle", "_build", "oracle_cli")
TAGS = [s + str(f) for s in "+-" for f in (1, 2, 3)]
PARAMS = json.load(open(os.path.join(GOLD, "hmm_params_backsolved.json")))["hmm_params"]


@pytest.fixture(scope="module")
def cli():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return CLI


def _trim(v, digits=3):
    """my_fprintf: float narrowing, "%.3f", trailing zeros removed but one decimal kept"""
    s = "%.*f" % (digits, float(np.float32(v)))
    if "." in s:
        s = s.rstrip("0")
        if s.endswith("."):
            s += "0"
    return s


def _pstr(p):
    return ",".join("%.17g" % x for x in p)


def _write_random_tracks(d, rng, n_runs=12):
    """six raw wigs: runs of 1..400 values, some of them written as several contiguous fixedStep sections"""
    for t in TAGS:
        with open(os.path.join(d, "PhyloCSFRaw%s.wig" % t), "w") as f:
            pos = 1000
            for r in range(n_runs):
                n = int(rng.integers(1, 400)) if r else 1
                vals = rng.normal(-8, 12, size=n)
                if r % 3 == 0:
                    vals[n // 3: 2 * n // 3] += 25        # a coding-looking stretch
                if r == 2:
                    vals[:] = rng.normal(12, 3, size=n)   # all coding (path of zeros)
                cut = sorted(set([0] + [int(c) for c in rng.integers(0, n, size=2)])) if r % 2 else [0]
                for k, v in enumerate(vals):
                    if k in cut:
                        f.write("fixedStep chrom=chrT start=%d step=3 span=3\n" % (pos + 3 * k))
                    f.write(_trim(v) + "\n")
                pos += 3 * n + int(rng.integers(1, 50)) * 3 + int(rng.integers(0, 3))


def test_smoothing_and_regions_match_the_python_restatement(cli, tmp_path):
    rng = np.random.default_rng(5)
    _write_random_tracks(str(tmp_path), rng)
    p = [0.01, 40.0, 0.5, 0.3, 0.2, 300.0, 5000.0, 30.0]
    subprocess.check_call([cli, "smooth-tracks", "--output-phylo", "1", "--output-regions", "1", "--hmm-params", _pstr(p), str(tmp_path)])
    init, trans = ho.make_hmm(p)
    n_regions = 0
    for t in TAGS:
        want_wig, want_bed = [], []
        for chrom, start, vals in ho.wig_runs(str(tmp_path / ("PhyloCSFRaw%s.wig" % t))):
            post = ho.coding_posterior(init, trans, vals)
            want_wig.append("fixedStep chrom=%s start=%d step=3 span=3" % (chrom, start))
            want_wig += [_trim(ho.log_odds(q)) for q in post]
            for s, e, _prob in ho.regions(ho.viterbi(init, trans, vals), post, start):
                want_bed.append("%s\t%d\t%d\t%s:%d-%d\t0\t%s\t%d\t%d\t0,0,0" % (chrom, s, e, chrom, s + 1, e, t[0], s, e))
        assert open(tmp_path / ("PhyloCSF%s.wig" % t)).read().splitlines() == want_wig, t
        assert open(tmp_path / ("PhyloCSF%sRegions.bed" % t)).read().splitlines() == want_bed, t
        n_regions += len(want_bed)
        assert os.path.exists(tmp_path / ("PhyloCSFRaw%s.wig" % t))
    assert n_regions > 10


def test_smoothed_golden_tracks_within_backsolved_tolerance(cli, tmp_path):
    for t in TAGS:
        with gzip.open(os.path.join(GOLD, "PhyloCSFRaw%s.wig.gz" % t), "rb") as fi, open(tmp_path / ("PhyloCSFRaw%s.wig" % t), "wb") as fo:
            fo.write(fi.read())
    subprocess.check_call([cli, "smooth-tracks", "--hmm-params", _pstr(PARAMS), str(tmp_path)])
    sq, n, worst, same = 0.0, 0, 0.0, 0
    for t in TAGS:
        got = open(tmp_path / ("PhyloCSF%s.wig" % t)).read().splitlines()
        want = gzip.open(os.path.join(GOLD, "PhyloCSF%s.wig.gz" % t), "rt").read().splitlines()
        assert len(got) == len(want)
        for g, w in zip(got, want):
            if w[0] == "f":
                assert g == w                      # run boundaries and coordinates: exact
                continue
            d = float(g) - float(w)
            sq += d * d
            n += 1
            worst = max(worst, abs(d))
            same += g == w
    assert n == 174648
    assert np.sqrt(sq / n) < 0.007 and worst <= 0.1001, (np.sqrt(sq / n), worst)   # log-odds units; see fit_hmm_params.py
    assert same > 0.25 * n


def test_parameter_estimation_matches_the_python_restatement(cli, tmp_path):
    rng = np.random.default_rng(11)
    lines = []
    for chrom in ("chr1", "chr2"):
        for strand in "+-":
            for frame in "012":
                pos = 1000
                for _ in range(120):
                    pos += int(rng.choice([rng.exponential(300), rng.exponential(3000), rng.exponential(60000)])) + 1
                    ln = int(rng.integers(30, 600))
                    lines.append("%s\t%s\t%s\t%d\t%d" % (chrom, strand, frame, pos, pos + ln))
                    if rng.random() < 0.15:   # nested / overlapping exon: one of the two is dropped
                        lines.append("%s\t%s\t%s\t%d\t%d" % (chrom, strand, frame, pos + 5, pos + int(rng.integers(10, 900))))
                    pos += ln
    rng.shuffle(lines)
    exons = tmp_path / "exons.txt"
    exons.write_text("\n".join(lines) + "\n")
    genome = 123456789
    want = ho.estimate_params(lines, genome)
    d = tmp_path / "o"
    d.mkdir()
    (d / "PhyloCSFRaw+1.wig").write_text("fixedStep chrom=c start=1 step=3 span=3\n1.0\n-3.5\n")
    for t in TAGS[1:]:
        (d / ("PhyloCSFRaw%s.wig" % t)).write_text("")
    r = subprocess.run([cli, "smooth-tracks", "--genome-length", str(genome), "--coding-exons", str(exons), str(d)], capture_output=True, text=True, check=True)
    got = [float(x) for x in re.search(r"HMM parameters: (.*)", r.stderr).group(1).split(",")]
    np.testing.assert_allclose(got, want, rtol=1e-12)
    assert 0 < got[0] < 1 and abs(sum(got[2:5]) - 1) < 1e-9


def test_build_tracks_writes_smoothed_tracks_and_drops_unwanted_raw_ones(cli, tmp_path):
    maf = tmp_path / "g.maf"
    with gzip.open(os.path.join(GOLD, "galGal6_chr22_25_28_each_30k_bases.maf.gz"), "rt") as fi:
        blocks = fi.read().split("\n\n")
    maf.write_text("\n\n".join(blocks[:200]) + "\n\n")      # the first ~2 kb: seconds on the CPU oracle
    out = tmp_path / "out"
    subprocess.check_call([cli, "build-tracks", "--output", str(out), "--output-phylo", "1", "--output-raw-phylo", "0", "--hmm-params", _pstr(PARAMS),
                           os.path.join(ROOT, "tests", "golden", "models", "53birds"), str(maf)], stderr=subprocess.DEVNULL)
    names = sorted(os.listdir(out))
    assert names == sorted(["PhyloCSFpower.wig"] + ["PhyloCSF%s.wig" % t for t in TAGS])
    golden = gzip.open(os.path.join(GOLD, "PhyloCSF+1.wig.gz"), "rt").read().splitlines()
    got = open(out / "PhyloCSF+1.wig").read().splitlines()
    assert got[0] == golden[0] and len(got) > 100
    # without parameters the request is refused like in the reference
    r = subprocess.run([cli, "build-tracks", "--output", str(out), "--output-phylo", "1", os.path.join(ROOT, "tests", "golden", "models", "53birds"), str(maf)],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "--genome-length" in r.stderr


def test_score_msa_fixed_mean_matches_the_python_restatement(cli, tmp_path):
    """--strategy FIXED_MEAN (reference src/phylocsf++score_msa.hpp:133-212): per-codon FIXED scores of a block, smoothed as
    one run, float-accumulated mean of the log-odds."""
    import shutil
    import maf_blocks
    import oracle_binding as ob
    src = os.path.join(ROOT, "tests", "golden", "score-msa", "chr22.50alignments.maf")
    maf = tmp_path / "small.maf"
    shutil.copyfile(src, maf)
    prefix = os.path.join(ROOT, "tests", "golden", "models", "100vertebrates")
    p = [0.004, 55.0, 0.45, 0.45, 0.10, 800.0, 9000.0, 60.0]
    subprocess.check_call([cli, "score-msa", "--strategy", "fixed_mean", "--comp-bls", "0", "--hmm-params", _pstr(p), prefix, str(maf)], stderr=subprocess.DEVNULL)
    rows = [l.split("\t") for l in open(str(maf) + ".scores").read().splitlines() if not l.startswith("#")][1:]
    assert len(rows) == 50
    m = ob.OracleModel(prefix)
    cod, off = maf_blocks.block_codons(src, m.labels, os.path.join(ROOT, "tests", "golden", "models", "species_aliases.tsv"))
    init, trans = ho.make_hmm(p)
    n_checked = 0
    for b, row in enumerate(rows):
        sub = np.ascontiguousarray(cod[:, int(off[b]):int(off[b + 1])])
        if sub.shape[1] == 0:
            assert row[4] in ("nan", "-nan")
            continue
        post = ho.coding_posterior(init, trans, list(m.run_tracks(sub)))
        acc = np.float32(0.0)
        for q in post:
            acc = np.float32(float(acc) + ho.log_odds(q))
        want = np.float32(acc / np.float32(len(post)))
        assert abs(float(row[4]) - float(want)) <= 1e-6 * max(1.0, abs(float(want))), (b, row[4], want)
        n_checked += 1
    assert n_checked >= 40
