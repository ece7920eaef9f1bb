"""End to end on the GPU: the host driver (phylocsfpp_amd/host/phylocsf-amd) calling the HIP path
through the C-ABI must regenerate the reference's golden files (tests/golden/)."""
import gzip
import os
import shutil
import subprocess

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
CLI = os.path.join(ROOT, "phylocsfpp_amd", "host", "phylocsf-amd")


def _cli():
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(CLI)])   # no-op when the driver is up to date
    return CLI


# branch-length scores from the device kernel (default) / on the host; the files' text formatted on the device (default) / by the host;
# the MAF's rows parsed straight into the batch matrices (default) / through strings and a packing copy
@pytest.mark.parametrize("host_bls,device_wig,direct_parse", [("0", "1", "1"), ("0", "0", "1"), ("1", "1", "1"), ("0", "1", "0")])
def test_build_tracks_golden_wigs_on_gpu(tmp_path, host_bls, device_wig, direct_parse):
    maf = tmp_path / "galGal6.maf"
    with gzip.open(os.path.join(GOLD, "build-tracks", "galGal6_chr22_25_28_each_30k_bases.maf.gz"), "rb") as fi, open(maf, "wb") as fo:
        shutil.copyfileobj(fi, fo)
    out = tmp_path / "out"
    subprocess.check_call([_cli(), "build-tracks", "--output", str(out), "--host-bls", host_bls, os.path.join(GOLD, "models", "53birds"), str(maf)],
                          env=dict(os.environ, PCSF_DEVICE_WIG=device_wig, PCSF_DIRECT_PARSE=direct_parse))
    for n in ["PhyloCSFpower"] + ["PhyloCSFRaw%s%d" % (s, f) for s in "+-" for f in (1, 2, 3)]:
        want = gzip.open(os.path.join(GOLD, "build-tracks", n + ".wig.gz"), "rb").read()
        got = open(out / (n + ".wig"), "rb").read()
        assert got == want, n                      # the reference's own test is `diff -r`


def test_build_tracks_without_the_power_track_on_gpu(tmp_path):
    """--output-power 0: six files, formatted on the device without the power track's lines (want_power = 0 of the C-ABI call)"""
    maf = tmp_path / "galGal6.maf"
    with gzip.open(os.path.join(GOLD, "build-tracks", "galGal6_chr22_25_28_each_30k_bases.maf.gz"), "rb") as fi, open(maf, "wb") as fo:
        shutil.copyfileobj(fi, fo)
    out = tmp_path / "out"
    subprocess.check_call([_cli(), "build-tracks", "--output", str(out), "--output-power", "0", os.path.join(GOLD, "models", "53birds"), str(maf)])
    assert sorted(os.listdir(out)) == sorted("PhyloCSFRaw%s%d.wig" % (s, f) for s in "+-" for f in (1, 2, 3))
    for n in os.listdir(out):
        assert open(out / n, "rb").read() == gzip.open(os.path.join(GOLD, "build-tracks", n + ".gz"), "rb").read(), n


def _scores(path):
    return [l.rstrip("\n").split("\t") for l in open(path) if not l.startswith("#")][1:]


@pytest.mark.parametrize("maf,gold", [("chr22.50alignments.maf", "chr22.50alignments.fixed.scores"),
                                      ("chr22.516alignments.maf.gz", "chr22.516alignments.fixed.scores")])
def test_score_msa_fixed_golden_on_gpu(tmp_path, maf, gold):
    src = os.path.join(GOLD, "score-msa", maf)
    dst = tmp_path / maf.replace(".gz", "")
    if maf.endswith(".gz"):
        with gzip.open(src, "rb") as fi, open(dst, "wb") as fo:
            shutil.copyfileobj(fi, fo)
    else:
        shutil.copyfile(src, dst)
    subprocess.check_call([_cli(), "score-msa", "--strategy", "fixed", "--comp-anc", "1",
                           os.path.join(GOLD, "models", "100vertebrates"), str(dst)])
    assert _scores(str(dst) + ".scores") == _scores(os.path.join(GOLD, "score-msa", gold))


def test_score_msa_mle_and_omega_golden_on_gpu(tmp_path):
    src = os.path.join(GOLD, "score-msa", "chr22.50alignments.maf")
    for strategy, anc, tol in (("mle", "1", 0.001), ("omega", "0", 0.1)):
        dst = tmp_path / ("small_%s.maf" % strategy)
        shutil.copyfile(src, dst)
        subprocess.check_call([_cli(), "score-msa", "--strategy", strategy, "--comp-anc", anc,
                               os.path.join(GOLD, "models", "100vertebrates"), str(dst)])
        got = _scores(str(dst) + ".scores")
        want = _scores(os.path.join(GOLD, "score-msa", "chr22.50alignments.%s.scores" % strategy))
        assert len(got) == len(want) == 50
        for g, w in zip(got, want):
            assert g[:4] == w[:4] and g[-1] == w[-1]                  # coordinates, strand, bls: exact
            for gv, wv in zip(g[4:-1], w[4:-1]):
                if wv == "nan":
                    continue                                          # see test_gpu_blocks.test_omega_blocks
                assert (float(gv) - float(wv)) ** 2 <= tol            # reference test/tests.sh:41,46


def test_species_reduced_model_on_gpu_matches_oracle_driver(tmp_path):
    """--species (tree reduced to 20 of the 53 birds; kernel program, tables and leaf count change at run time):
    the GPU driver must write the same seven wig files as the oracle-backed driver on the same input."""
    oracle_cli = os.path.join(ROOT, "oracle", "_build", "oracle_cli")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    maf = tmp_path / "galGal6.maf"
    with gzip.open(os.path.join(GOLD, "build-tracks", "galGal6_chr22_25_28_each_30k_bases.maf.gz"), "rb") as fi, open(maf, "wb") as fo:
        shutil.copyfileobj(fi, fo)
    labels = [l.split("\t")[0] for l in open(os.path.join(GOLD, "models", "species_aliases.tsv")) if l.strip()]
    tree = open(os.path.join(GOLD, "models", "53birds.nh")).read().lower()
    birds = [l for l in labels if (l.lower() + ":") in tree][:40:2]
    assert len(birds) == 20
    outs = []
    for cli, name in ((_cli(), "gpu"), (oracle_cli, "cpu")):
        out = tmp_path / name
        subprocess.check_call([cli, "build-tracks", "--output", str(out), "--species", ",".join(birds + ["Chicken"]),
                               os.path.join(GOLD, "models", "53birds"), str(maf)])
        outs.append(out)
    for n in ["PhyloCSFpower"] + ["PhyloCSFRaw%s%d" % (s, f) for s in "+-" for f in (1, 2, 3)]:
        got, want = open(outs[0] / (n + ".wig"), "rb").read(), open(outs[1] / (n + ".wig"), "rb").read()
        assert len(want) > 10000 and got == want, n


def test_build_tracks_smoothed_tracks_on_gpu(tmp_path):
    """--output-phylo 1: raw tracks byte-identical to the goldens, smoothed tracks within the back-solved-parameter
    tolerance of the reference's (tests/test_hmm.py, tests/golden/fit_hmm_params.py)."""
    import json
    maf = tmp_path / "galGal6.maf"
    with gzip.open(os.path.join(GOLD, "build-tracks", "galGal6_chr22_25_28_each_30k_bases.maf.gz"), "rb") as fi, open(maf, "wb") as fo:
        shutil.copyfileobj(fi, fo)
    prm = json.load(open(os.path.join(GOLD, "build-tracks", "hmm_params_backsolved.json")))["hmm_params"]
    out = tmp_path / "out"
    subprocess.check_call([_cli(), "build-tracks", "--output", str(out), "--output-phylo", "1", "--hmm-params", ",".join("%.17g" % x for x in prm),
                           os.path.join(GOLD, "models", "53birds"), str(maf)])
    worst, n = 0.0, 0
    for s in "+-":
        for f in (1, 2, 3):
            tag = "%s%d" % (s, f)
            assert open(out / ("PhyloCSFRaw%s.wig" % tag), "rb").read() == gzip.open(os.path.join(GOLD, "build-tracks", "PhyloCSFRaw%s.wig.gz" % tag), "rb").read()
            got = open(out / ("PhyloCSF%s.wig" % tag)).read().splitlines()
            want = gzip.open(os.path.join(GOLD, "build-tracks", "PhyloCSF%s.wig.gz" % tag), "rt").read().splitlines()
            assert len(got) == len(want)
            for g, w in zip(got, want):
                if w[0] == "f":
                    assert g == w
                else:
                    worst = max(worst, abs(float(g) - float(w)))
                    n += 1
    assert n == 174648 and worst <= 0.1001


def test_synthetic_maf_across_the_break_point_matches_oracle_driver(tmp_path):
    """A generated 100-species alignment (discontinuities, species drop-out, gaps, lower case) that crosses the 1 Mb
    break point, 58mammals model (42 rows are skipped), several MAF windows and byte-range jobs: the GPU driver and the
    oracle-backed driver must write identical wigs; --strategy fixed_mean and --output-phylo ride along."""
    import sys
    oracle_cli = os.path.join(ROOT, "oracle", "_build", "oracle_cli")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    maf = tmp_path / "syn.maf"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_synthetic_maf.py"), str(maf), "--bases", "24000", "--start", "990000", "--seed", "3"],
                          stdout=subprocess.DEVNULL)
    hmm = "0.004,55,0.45,0.45,0.1,800,9000,60"
    outs = []
    for cli, name in ((_cli(), "gpu"), (oracle_cli, "cpu")):
        out = tmp_path / name
        subprocess.check_call([cli, "build-tracks", "--output", str(out), "--output-phylo", "1", "--hmm-params", hmm,
                               os.path.join(GOLD, "models", "58mammals"), str(maf)], stderr=subprocess.DEVNULL)
        outs.append(out)
    names = sorted(os.listdir(outs[1]))
    assert len(names) == 13
    for n in names:
        got, want = open(outs[0] / n, "rb").read(), open(outs[1] / n, "rb").read()
        assert len(want) > 1000
        if n.startswith("PhyloCSFRaw") or n == "PhyloCSFpower.wig":
            assert got == want, n
        else:  # smoothed: the raw text is identical, so the HMM input is, and so is its output
            assert got == want, n


def test_a_failure_of_the_device_smoothing_pass_half_way_loses_nothing(tmp_path):
    """--output-raw-phylo 0 (the raw tracks are removed once they are smoothed): the device pass is cut into groups of one track
    (PCSF_HMM_GROUP_VALUES) and made to throw in its third group (PCSF_HMM_FAIL_GROUP): the two finished tracks must be kept as they
    are, the other four come from the host pass, and the result equals an undisturbed run byte for byte -- no truncated file, no
    missing raw track under the host pass, raw tracks removed only at the very end."""
    import sys
    maf = tmp_path / "syn.maf"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_synthetic_maf.py"), str(maf), "--bases", "24000", "--seed", "5"], stdout=subprocess.DEVNULL)
    hmm = "0.004,55,0.45,0.45,0.1,800,9000,60"
    outs = {}
    for name, env in (("plain", {}), ("failing", {"PCSF_HMM_GROUP_VALUES": "1", "PCSF_HMM_FAIL_GROUP": "2"}), ("host", {"PCSF_HOST_HMM": "1"})):
        out = tmp_path / name
        r = subprocess.run([_cli(), "build-tracks", "--output", str(out), "--output-phylo", "1", "--output-regions", "1", "--output-raw-phylo", "0", "--hmm-params", hmm,
                            os.path.join(GOLD, "models", "58mammals"), str(maf)], env=dict(os.environ, **env), stderr=subprocess.PIPE, text=True)
        assert r.returncode == 0, r.stderr
        assert ("the smoothing pass on the device failed" in r.stderr) == (name == "failing"), r.stderr
        outs[name] = out
    names = sorted(os.listdir(outs["plain"]))
    assert len(names) == 13 and not any(n.startswith("PhyloCSFRaw") for n in names)      # power + 6 smoothed + 6 region files
    for other in ("failing", "host"):
        assert sorted(os.listdir(outs[other])) == names
        for n in names:
            a, b = open(outs["plain"] / n, "rb").read(), open(outs[other] / n, "rb").read()
            assert a == b and (len(a) > 100 or n.endswith("Regions.bed")), (other, n)      # (a frame may have no coding region)


def test_two_scoring_contexts_give_the_same_files(tmp_path):
    """--devices 0,0: two contexts (two host threads, two streams) share the batches of build-tracks and the block ranges
    of score-msa -- the multi-GPU code path of the driver, exercised on one GPU; outputs must not change."""
    maf = tmp_path / "galGal6.maf"
    with gzip.open(os.path.join(GOLD, "build-tracks", "galGal6_chr22_25_28_each_30k_bases.maf.gz"), "rb") as fi, open(maf, "wb") as fo:
        shutil.copyfileobj(fi, fo)
    out = tmp_path / "out"
    subprocess.check_call([_cli(), "build-tracks", "--output", str(out), "--devices", "0,0", "--batch-sites", "30000", "--window-bytes", "1000000",
                           os.path.join(GOLD, "models", "53birds"), str(maf)])
    for n in ["PhyloCSFpower"] + ["PhyloCSFRaw%s%d" % (s, f) for s in "+-" for f in (1, 2, 3)]:
        assert open(out / (n + ".wig"), "rb").read() == gzip.open(os.path.join(GOLD, "build-tracks", n + ".wig.gz"), "rb").read(), n
    res = []
    for i, extra in enumerate(([], ["--devices", "0,0"])):
        m = tmp_path / ("s%d.maf" % i)
        shutil.copyfile(os.path.join(GOLD, "score-msa", "chr22.50alignments.maf"), m)
        subprocess.check_call([_cli(), "score-msa", "--strategy", "omega"] + extra + [os.path.join(GOLD, "models", "100vertebrates"), str(m)])
        res.append(_scores(str(m) + ".scores"))
    assert res[0] == res[1] and len(res[0]) == 50


def test_score_msa_scores_do_not_depend_on_how_the_file_is_cut_across_devices(tmp_path):
    """src/phylocsf++score_msa.hpp:115 "make sure that parallelization does not affect results".  A MIXED file -- dense synthetic blocks
    (every species present: on their own, prune_tiles' kind of list) followed by the reference's 516 real blocks (4 of 100 species:
    prune_factored's) -- scored with one context and cut over two and three: the halves of the cut would pick different pruning kernels
    from their own lists (the kernels round differently), score_msa decides ONCE for the file and pins it (pcsf_set_block_kernel), so
    the three .scores files are the same bytes, MLE and OMEGA."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_synthetic_maf as gen
    full = tmp_path / "dense.maf"
    gen.generate(str(full), bases=60_000, seed=11, tree=os.path.join(GOLD, "models", "100vertebrates"))
    mixed = tmp_path / "mixed.maf"
    with open(mixed, "wb") as fo:
        fo.write(open(full, "rb").read())
        fo.write(b"\n")
        with gzip.open(os.path.join(GOLD, "score-msa", "chr22.516alignments.maf.gz"), "rb") as fi:
            body = fi.read()
        fo.write(body[body.index(b"\na ") + 1:] if not body.startswith(b"a ") else body)
    env = dict(os.environ, PCSF_TIME_BLOCKS="1")
    for strategy, anc in (("mle", "1"), ("omega", "0")):
        outs, kernels = [], []
        for i, devs in enumerate((["--device", "0"], ["--devices", "0,0"], ["--devices", "0,0,0"])):
            out = tmp_path / ("%s_%d" % (strategy, i))
            os.makedirs(out)
            p = subprocess.run([_cli(), "score-msa", "--strategy", strategy, "--comp-anc", anc, "--output", str(out)] + devs +
                               [os.path.join(GOLD, "models", "100vertebrates"), str(mixed)], capture_output=True, text=True, env=env)
            assert p.returncode == 0, p.stderr[-2000:]
            outs.append(open(out / "mixed.maf.scores", "rb").read())
            kernels.append(sorted(set(l.split(": ")[1] for l in p.stderr.splitlines() if l.startswith("pruning kernel of this list"))))
        rows = outs[0].count(b"\n") - 2
        assert rows > 550, rows                                         # the dense blocks and the 516 real ones
        assert all(len(k) == 1 for k in kernels) and kernels[0] == kernels[1] == kernels[2], kernels    # one kernel per FILE
        assert outs[0] == outs[1] == outs[2], strategy
    # what the pin is for: left to themselves, the two halves of this file choose different kernels
    import numpy as np
    import maf_blocks
    import oracle_binding as ob
    import phylocsfpp_amd as pa
    pm = pa.load_model(os.path.join(GOLD, "models", "100vertebrates"))
    cod, off = maf_blocks.block_codons(str(mixed), pm.labels, os.path.join(GOLD, "models", "species_aliases.tsv"))
    ctx = pm.context()
    try:
        half = int(np.searchsorted(off, off[-1] // 2))
        whole = ctx.choose_block_kernel(cod, off)
        first, second = ctx.choose_block_kernel(cod, off[:half + 1]), ctx.choose_block_kernel(cod, off[half:])
        assert first != second and whole in (first, second), (whole, first, second)
        # the same through the library: a list scored whole and cut into two calls, kernel pinned -> the same floats
        ctx.set_block_kernel(whole)
        p_all, a_all = ctx.score_blocks(pa.MLE, cod, off)
        p_a, a_a = ctx.score_blocks(pa.MLE, cod, off[:half + 1])
        p_b, a_b = ctx.score_blocks(pa.MLE, cod, off[half:])
        assert p_all.tobytes() == np.concatenate([p_a, p_b]).tobytes() and a_all.tobytes() == np.concatenate([a_a, a_b]).tobytes()
        ctx.set_block_kernel(pa.ScoringContext.BLOCK_KERNEL_AUTO)
    finally:
        ctx.close()


def test_score_msa_mle_and_omega_medium_file_on_gpu(tmp_path):
    """The reference's 516-block file (its goldens are shipped but not part of its own test script), MLE and OMEGA.
    PhyloCSF scores: every MLE value within the reference's MLE tolerance (test/tests.sh:41), every OMEGA value within
    its OMEGA tolerance (:46); golden NaN rows (GSL's eigen-solver degrading on short blocks) are skipped.
    The MLE ancestral score is a first-order function of the fitted tree scale, which Brent only brackets to 1 %
    and reports as its LAST probe: on long blocks a different last probe moves it by up to ~0.7 (the CPU oracle
    gives the same values as the GPU there) -- 95 % of the values meet the tolerance, all stay within 1.0."""
    src = os.path.join(GOLD, "score-msa", "chr22.516alignments.maf.gz")
    for strategy, anc in (("mle", "1"), ("omega", "0")):
        dst = tmp_path / ("medium_%s.maf" % strategy)
        with gzip.open(src, "rb") as fi, open(dst, "wb") as fo:
            shutil.copyfileobj(fi, fo)
        subprocess.check_call([_cli(), "score-msa", "--strategy", strategy, "--comp-anc", anc, os.path.join(GOLD, "models", "100vertebrates"), str(dst)])
        got = _scores(str(dst) + ".scores")
        want = _scores(os.path.join(GOLD, "score-msa", "chr22.516alignments.%s.scores" % strategy))
        assert len(got) == len(want) == 516
        phylo_checked = anc_close = 0
        for g, w in zip(got, want):
            assert g[:4] == w[:4] and g[-1] == w[-1]                  # coordinates, strand, bls: exact
            if w[4] not in ("nan", "-nan"):
                assert (float(g[4]) - float(w[4])) ** 2 <= (0.001 if strategy == "mle" else 0.1), (strategy, g, w)
                phylo_checked += 1
            if strategy == "mle":
                d = abs(float(g[5]) - float(w[5]))
                assert d <= 1.0, (g, w)
                anc_close += d * d <= 0.001
        if strategy == "mle":
            assert phylo_checked == 516 and anc_close >= 0.95 * 516
        else:
            assert phylo_checked >= 400


def test_build_tracks_rejects_other_characters_on_gpu(tmp_path):
    """the device validates the characters when it computes the branch-length scores (the reference exits on
    anything but ACGTNacgtn-., src/translation.hpp:29-88)"""
    maf = tmp_path / "bad.maf"
    maf.write_text("a score=0\ns chicken.chr1 10 4 + 1000 ACXT\ns turkey.chr1 10 4 + 1000 ACGT\n")
    r = subprocess.run([_cli(), "build-tracks", "--output", str(tmp_path / "o"), os.path.join(GOLD, "models", "53birds"), str(maf)],
                       capture_output=True, text=True)
    assert r.returncode != 0 and "unexpected character 'X'" in r.stderr


def test_config5_100vertebrates_all_rows_hmm_estimated_from_exons(tmp_path):
    """BASELINE.json configs[4] end to end at test size: build-tracks with the 100-leaf model on a generated MAF whose 100
    rows are ALL mapped, --output-phylo 1 --output-regions 1 with the HMM parameters ESTIMATED from the generator's exon
    list (--genome-length / --coding-exons: estimate_hmm_params_for_genome, src/estimate_hmm_parameter.hpp:243-340, then
    src/phylocsf++build_tracks.hpp:262-364,421-424).  GPU driver vs the oracle-backed driver: all 19 files identical;
    the estimated parameters equal the independent Python restatement's (tests/hmm_oracle.py)."""
    import re
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import hmm_oracle as ho
    oracle_cli = os.path.join(ROOT, "oracle", "_build", "oracle_cli")
    maf, exons = tmp_path / "c5.maf", tmp_path / "c5.exons"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_synthetic_maf.py"), str(maf), "--bases", "24000", "--seed", "5",
                           "--exons", str(exons)], stdout=subprocess.DEVNULL)
    genome_length = 2048000
    outs, params = [], []
    for cli, name in ((_cli(), "gpu"), (oracle_cli, "cpu")):
        out = tmp_path / name
        r = subprocess.run([cli, "build-tracks", "--output", str(out), "--output-phylo", "1", "--output-regions", "1", "--genome-length", str(genome_length),
                            "--coding-exons", str(exons), os.path.join(GOLD, "models", "100vertebrates"), str(maf)], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        params.append([float(x) for x in re.search(r"HMM parameters: (\S+)", r.stderr).group(1).split(",")])
        assert re.search(r"(\d+) reference bases", r.stderr).group(1) == "24000"
        outs.append(out)
    want = ho.estimate_params(open(exons).read().splitlines(), genome_length)
    flat = [want["coding_prior"], want["coding_length"]] + list(want["weights"]) + list(want["lengths"]) if isinstance(want, dict) else list(want)
    assert params[0] == params[1]
    assert all(abs(a - b) <= 1e-9 * max(1.0, abs(b)) for a, b in zip(params[0], flat)), (params[0], flat)
    names = sorted(os.listdir(outs[1]))
    assert len(names) == 19                                            # power, 6 raw, 6 smoothed, 6 region files
    n_bytes = 0
    for n in names:
        got, want_b = open(outs[0] / n, "rb").read(), open(outs[1] / n, "rb").read()
        assert got == want_b, n
        n_bytes += len(got)
    assert n_bytes > 500000


def test_packed_alignment_file_on_gpu(tmp_path):
    """pack-maf + build-tracks on the packed file through the GPU driver: same seven files as from the MAF text, twice
    (a second run must reproduce the first bit for bit -- the 4 Mb run that exposed the LDS-DMA race differed from itself)."""
    import sys
    maf = tmp_path / "syn.maf"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_synthetic_maf.py"), str(maf), "--bases", "150000", "--seed", "11"],
                          stdout=subprocess.DEVNULL)
    packed = tmp_path / "syn.pcsfbin"
    model = os.path.join(GOLD, "models", "58mammals")
    subprocess.check_call([_cli(), "pack-maf", model, str(maf), str(packed)], stderr=subprocess.DEVNULL)
    outs = []
    for src, extra in ((maf, []), (maf, ["--window-bytes", "3000000"]), (packed, []), (packed, ["--window-bytes", "2000000", "--batch-sites", "50000"])):
        out = tmp_path / ("o%d" % len(outs))
        subprocess.check_call([_cli(), "build-tracks", "--output", str(out)] + extra + [model, str(src)], stderr=subprocess.DEVNULL)
        outs.append(out)
    names = sorted(os.listdir(outs[0]))
    assert len(names) == 7
    for n in names:
        want = open(outs[0] / n, "rb").read()
        assert len(want) > 100000
        for o in outs[1:]:
            assert open(o / n, "rb").read() == want, (n, str(o))
