"""CPU: the C++ host layer (phylocsfpp_amd/host/: MAF reader, frames, BLS, formatting) on crafted inputs,
through the oracle-backed test CLI.  The golden-file tests pin the common path; these pin the corner
cases the goldens do not reach: block concatenation rules, N padding, reference
gap removal, the 1 Mb break point, '-' strand frames, trimming of printed numbers."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "oracle", "_build", "oracle_cli")
MODEL = os.path.join(ROOT, "tests", "golden", "models", "53birds")   # first leaves: chicken, japanese_quail, turkey


@pytest.fixture(scope="module")
def cli():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return CLI


def _dump(cli, path, blocks=False):
    out = subprocess.check_output([cli, "dump-alignments"] + (["--blocks", "1"] if blocks else []) + [MODEL, path], text=True)
    # the same alignments measured and then parsed straight into a matrix (MafReader::measure / fill: what build-tracks does with text input)
    assert subprocess.check_output([cli, "dump-alignments", "--matrix", "1"] + (["--blocks", "1"] if blocks else []) + [MODEL, path], text=True) == out
    rows = []
    for line in out.splitlines():
        f = line.split("\t")
        if f[0] == "name":          # the row-name map follows the nodes
            continue
        rows.append(dict(chrom=f[0], start=int(f[1]), length=int(f[2]), strand=f[3], chrom_len=int(f[4]), seqs=f[5:]))
    return rows


def _write(tmp_path, text, name="t.maf"):
    p = tmp_path / name
    p.write_text(text)
    return str(p)


def test_concatenation_padding_and_reference_gaps(cli, tmp_path):
    maf = """##maf version=1
a score=1
s chicken.chr1 99 6 + 1000 AC-GTA-A
s turkey.chrX 5 8 + 500 ACTGTACA
s unknownSpecies.chr1 0 8 + 9 AAAAAAAA

a score=2
s chicken.chr1 105 3 + 1000 GGG
s japanese_quail.c 7 3 - 50 ccc

a score=3
s chicken.chr1 200 2 + 1000 TT
"""
    rows = _dump(cli, _write(tmp_path, maf))
    assert len(rows) == 2                                   # blocks 1+2 are contiguous (99+6 == 105), block 3 is not
    a = rows[0]
    assert (a["chrom"], a["start"], a["length"], a["strand"], a["chrom_len"]) == ("chr1", 100, 9, "+", 1000)
    assert a["seqs"][0] == "ACGTAAGGG"                      # chicken: gap columns removed
    assert a["seqs"][2] == "ACGTAANNN"                      # turkey: reference-gap columns dropped, missing block -> N
    assert a["seqs"][1] == "NNNNNNccc"                      # quail: absent from block 1 -> N padding
    assert all(s == "N" * 9 for s in a["seqs"][3:])
    assert rows[1]["start"] == 201 and rows[1]["seqs"][0] == "TT"
    # score-msa mode: every block alone
    rows = _dump(cli, _write(tmp_path, maf), blocks=True)
    assert [r["length"] for r in rows] == [6, 3, 2]


def test_break_point_at_one_megabase(cli, tmp_path):
    # block B crosses 1,000,000 (1-based 999,996..1,000,003); C and D supply the two extra bases; E would follow
    def s(start, seq):
        return "a score=0\ns chicken.chr2 %d %d + 5000000 %s\n\n" % (start, len(seq), seq)
    maf = "##maf\n" + s(999990, "AAAAA") + s(999995, "CCCCCCCC") + s(1000003, "G") + s(1000004, "TTT") + s(1000007, "ACAC")
    rows = _dump(cli, _write(tmp_path, maf))
    # first alignment: A + B (break-point block) + exactly two more reference bases, truncated there
    assert rows[0]["start"] == 999991 and rows[0]["seqs"][0] == "AAAAA" + "CCCCCCCC" + "G" + "T"
    # the next alignment restarts at the block AFTER the break-point block (blocks C, D are read twice)
    assert rows[1]["start"] == 1000004 and rows[1]["seqs"][0] == "G" + "TTT" + "ACAC"
    assert len(rows) == 2


def test_minus_strand_reference_is_rejected_in_build_tracks_mode(cli, tmp_path):
    maf = "a score=0\ns chicken.chr1 10 4 - 1000 ACGT\n"
    path = _write(tmp_path, maf)
    r = subprocess.run([cli, "dump-alignments", MODEL, path], capture_output=True, text=True)
    assert r.returncode != 0 and "+ strand" in r.stderr
    assert _dump(cli, path, blocks=True)[0]["strand"] == "-"


def test_malformed_blocks_are_refused_by_both_forms_of_the_reader(cli, tmp_path):
    """A row longer than its block's reference row and a second row of a species in one block (the reference species included, also when
    its first row is all gaps) are errors -- in next() (strings) and in measure() + fill() (the direct parse build-tracks uses) alike; a
    SHORTER row is padded with N by both (the two forms had diverged on malformed input)."""
    ok = "a score=0\ns chicken.chr1 10 4 + 1000 ACGT\ns turkey.chr1 10 2 + 1000 AC\n"
    rows = _dump(cli, _write(tmp_path, ok, "ok.maf"))                       # _dump compares the two forms
    assert rows[0]["seqs"][:2] == ["ACGT", "ACNN"] or "ACNN" in rows[0]["seqs"]
    cases = {"long": ("a score=0\ns chicken.chr1 10 4 + 1000 ACGT\ns turkey.chr1 10 6 + 1000 ACGTAC\n", "longer than"),
             "twice": ("a score=0\ns chicken.chr1 10 4 + 1000 ACGT\ns turkey.chr1 10 4 + 1000 ACGT\ns turkey.chr1 50 4 + 1000 ACGT\n", "two rows"),
             "ref_twice": ("a score=0\ns chicken.chr1 10 4 + 1000 ACGT\ns chicken.chr1 20 4 + 1000 ACGT\n", "two rows"),
             "gap_ref_twice": ("a score=0\ns chicken.chr1 10 4 + 1000 ACGT\n\na score=0\ns chicken.chr1 14 0 + 1000 ----\ns chicken.chr1 14 0 + 1000 ----\ns turkey.chr1 1 4 + 1000 ACGT\n", "two rows")}
    for name, (maf, message) in cases.items():
        path = _write(tmp_path, maf, name + ".maf")
        for extra in ([], ["--matrix", "1"]):
            r = subprocess.run([cli, "dump-alignments"] + extra + [MODEL, path], capture_output=True, text=True)
            assert r.returncode != 0 and message in r.stderr, (name, extra, r.returncode, r.stderr)


def test_frames_and_wig_text_on_a_tiny_alignment(cli, tmp_path):
    # 14 columns at 1-based position 101 on a chromosome of length 1000, three species with bases everywhere
    seq = "ATGGCCAAGTTTGA"
    maf = "a score=0\n" + "".join("s %s.chrZ 100 14 + 1000 %s\n" % (sp, seq) for sp in ("chicken", "turkey", "japanese_quail", "barn_owl"))
    out = tmp_path / "out"
    subprocess.check_call([cli, "build-tracks", "--power-threshold", "0", "--output", str(out), MODEL, _write(tmp_path, maf)],
                          stderr=subprocess.DEVNULL)
    def starts(name):
        lines = open(out / name).read().splitlines()
        return [l for l in lines if l.startswith("fixedStep")], [l for l in lines if not l.startswith("fixedStep")]
    # '+' frame f: skip = (f - 101) mod 3 -> f=1: 2, f=2: 0, f=3: 1 ; positions 101+skip
    for f, skip in ((1, 2), (2, 0), (3, 1)):
        hdr, vals = starts("PhyloCSFRaw+%d.wig" % f)
        assert hdr == ["fixedStep chrom=chrZ start=%d step=3 span=3" % (101 + skip)]
        assert len(vals) == (14 - skip) // 3
    # '-' frame f: skip = (f - (1000 - (101+14) + 2)) mod 3 = (f - 887) mod 3 -> f=1: 2, f=2: 0, f=3: 1 ; start = 101 + (14-skip)%3
    for f, skip in ((1, 2), (2, 0), (3, 1)):
        hdr, vals = starts("PhyloCSFRaw-%d.wig" % f)
        assert hdr == ["fixedStep chrom=chrZ start=%d step=3 span=3" % (101 + (14 - skip) % 3)]
        assert len(vals) == (14 - skip) // 3
    hdr, vals = starts("PhyloCSFpower.wig")
    assert hdr == ["fixedStep chrom=chrZ start=%d step=3 span=3" % (101 + (3 - 101) % 3)]
    # number formatting: at least one decimal, no trailing zeros beyond it
    for v in vals:
        assert "." in v and not (v.endswith("0") and not v.endswith(".0"))


def test_invalid_character_is_an_error(cli, tmp_path):
    maf = "a score=0\ns chicken.chr1 10 4 + 1000 ACXT\ns turkey.chr1 10 4 + 1000 ACGT\n"
    r = subprocess.run([cli, "build-tracks", "--output", str(tmp_path / "o"), MODEL, _write(tmp_path, maf)], capture_output=True, text=True)
    assert r.returncode != 0 and "unexpected character" in r.stderr


# ---- --species: tree reduction and name mapping (reference src/newick.hpp:286-363, src/models.hpp:1780-1857) ----
def _model_dir(tmp_path, newick):
    import shutil
    for kind in ("coding", "noncoding"):
        shutil.copy(MODEL + "_%s.ECM" % kind, tmp_path / ("t_%s.ECM" % kind))
    (tmp_path / "t.nh").write_text(newick)
    return str(tmp_path / "t")


def _dump_model(cli, prefix, species=None, mapping=None):
    cmd = [cli, "dump-model"]
    if species:
        cmd += ["--species", species]
    if mapping:
        cmd += ["--mapping", mapping]
    out = subprocess.check_output(cmd + [prefix], text=True)
    rows = []
    for line in out.splitlines():
        f = line.split("\t")
        if f[0] == "name":          # the row-name map follows the nodes
            continue
        rows.append((int(f[0]), int(f[1]), int(f[2]), int(f[3]), float(f[4]), f[6] if len(f) > 6 else ""))
    return rows


def test_species_reduction_merges_branches_like_the_reference(cli, tmp_path):
    prefix = _model_dir(tmp_path, "((A:1,b:2):3,(c:4,(d:5,e:6):7):8);")
    full = _dump_model(cli, prefix)
    assert [r[5] for r in full[:5]] == ["a", "b", "c", "d", "e"] and len(full) == 9
    # (a,b):3 loses b -> leaf a with 3+1; (c,(d,e):7):8 loses c -> (d,e) with 8+7
    red = _dump_model(cli, prefix, "a,D,e")
    assert red == [(0, -1, -1, 4, 4.0, "a"), (1, -1, -1, 3, 5.0, "d"), (2, -1, -1, 3, 6.0, "e"), (3, 1, 2, 4, 15.0, ""), (4, 0, 3, -1, 0.0, "")]
    # the root never accumulates a length: it becomes the (d,e) node
    assert _dump_model(cli, prefix, "d,e") == [(0, -1, -1, 2, 5.0, "d"), (1, -1, -1, 2, 6.0, "e"), (2, 0, 1, -1, 0.0, "")]
    # selecting everything changes nothing
    assert _dump_model(cli, prefix, "a,b,c,d,e") == full
    # unknown names are an error that lists them; fewer than two species too
    r = subprocess.run([cli, "dump-model", "--species", "a,zzz", prefix], capture_output=True, text=True)
    assert r.returncode != 0 and "zzz" in r.stderr
    assert subprocess.run([cli, "dump-model", "--species", "a", prefix], capture_output=True, text=True).returncode != 0


def test_species_accepts_other_names_and_mapping_file(cli, tmp_path):
    prefix100 = os.path.join(ROOT, "tests", "golden", "models", "100vertebrates")
    red = _dump_model(cli, prefix100, "hg38,Mouse,galGal6")        # assembly names resolve through the alias table
    assert [r[5] for r in red[:3]] == ["human", "mouse", "chicken"] and len(red) == 5
    # sum of the merged path lengths is preserved: human's branch in the 3-species tree = path to the human/mouse ancestor
    prefix = _model_dir(tmp_path, "((A:1,b:2):3,(c:4,(d:5,e:6):7):8);")
    mapping = tmp_path / "map.tsv"
    mapping.write_text("A\tasm1\nE\tasm5\n")
    red = _dump_model(cli, prefix, "asm1,asm5,d", str(mapping))
    assert [r[5] for r in red[:3]] == ["a", "d", "e"]


def test_reduced_model_scores_like_the_full_model_on_the_same_rows(cli, tmp_path):
    """Removing a species from the tree == marginalising it: FIXED scores of the reduced model equal those of the
    full model when the alignment only holds the selected species (up to the float narrowing of merged branches)."""
    import gzip
    prefix100 = os.path.join(ROOT, "tests", "golden", "models", "100vertebrates")
    keep = ["hg19", "pantro4", "ponabe2", "turtru2", "mondom5", "anocar2", "gasacu1"]   # 7 of the 11 species in the file
    src = os.path.join(ROOT, "tests", "golden", "score-msa", "chr22.50alignments.maf")
    lines = []
    opener = gzip.open if src.endswith(".gz") else open
    with opener(src, "rt") as f:
        for line in f:
            if line.startswith("s "):
                if line.split()[1].split(".")[0].lower() not in keep:
                    continue
            lines.append(line)
    maf = tmp_path / "sub.maf"
    maf.write_text("".join(lines))
    outs = []
    for i, extra in enumerate(([], ["--species", ",".join(keep)])):
        d = tmp_path / ("o%d" % i)
        d.mkdir()
        subprocess.check_call([cli, "score-msa", "--strategy", "fixed", "--comp-anc", "1", "--comp-bls", "0", "--output", str(d)] + extra +
                              [prefix100, str(maf)], stderr=subprocess.DEVNULL)
        rows = [l.split("\t") for l in open(d / "sub.maf.scores").read().splitlines() if not l.startswith("#")][1:]  # header row
        outs.append(np.array([[float(x) for x in r[-2:]] for r in rows]))
    assert outs[0].shape == outs[1].shape and outs[0].shape[0] == 50
    # the likelihood ratio is root-invariant; the ancestral-composition score is a statement about the ROOT, which
    # moves to the selected species' last common ancestor, so it legitimately changes
    np.testing.assert_allclose(outs[1][:, 0], outs[0][:, 0], rtol=2e-5, atol=2e-5)
    assert np.abs(outs[0][:, 0]).max() > 1.0 and np.abs(outs[1][:, 1] - outs[0][:, 1]).max() > 1e-3


# ---- parallel ingest: byte-range jobs (reference src/parallel_file_reader.hpp:283-350,396-425) and fast text ----
def _dump_text(cli, model, path, jobs):
    out = subprocess.check_output([cli, "dump-alignments", "--jobs", str(jobs), model, path], text=True)
    assert subprocess.check_output([cli, "dump-alignments", "--matrix", "1", "--jobs", str(jobs), model, path], text=True) == out   # measure() + fill()
    return out


def test_byte_range_jobs_do_not_change_the_alignments(cli, tmp_path):
    import gzip
    import sys
    # (a) the reference's example (10 097 blocks, 9 alignments), cut into 1..97 byte ranges
    maf = tmp_path / "g.maf"
    with gzip.open(os.path.join(ROOT, "tests", "golden", "build-tracks", "galGal6_chr22_25_28_each_30k_bases.maf.gz"), "rb") as fi:
        maf.write_bytes(fi.read())
    one = _dump_text(cli, MODEL, str(maf), 1)
    assert one.count("\n") == 9
    for jobs in (2, 3, 16, 97):
        assert _dump_text(cli, MODEL, str(maf), jobs) == one, jobs
    # (b) a synthetic 30 kb alignment that crosses the 1 Mb break point, discontinuities included
    syn = tmp_path / "syn.maf"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_synthetic_maf.py"), str(syn), "--bases", "30000", "--start", "985000",
                           "--seed", "7"], stdout=subprocess.DEVNULL)
    model100 = os.path.join(ROOT, "tests", "golden", "models", "100vertebrates")
    one = _dump_text(cli, model100, str(syn), 1)
    assert one.count("\n") >= 3
    for jobs in (2, 5, 33, 200):
        assert _dump_text(cli, model100, str(syn), jobs) == one, jobs


def test_fast_number_formatting_matches_printf(cli):
    out = subprocess.check_output([cli, "selftest-format", "600000"], text=True)
    lines = out.strip().splitlines()
    assert lines[0].endswith("0 mismatches") and " 0 mismatches" in lines[1], out   # numbers; AVX2 vs scalar character classes


def test_printed_value_bins_of_the_smoothing_pass_equal_the_printing_code(cli):
    """The device smoothing pass returns, per position, the step of the printed log-odds its posterior falls on (thresholds found by
    bisection with this host's libm): the text from the step must be the text from the posterior, on both sides of all 30,001 steps
    and on random posteriors."""
    out = subprocess.check_output([cli, "selftest-printed-codes", "1500000"], text=True)
    assert "ascending: yes" in out and out.strip().endswith(" 0 mismatches"), out


def test_rows_parsed_straight_into_the_batch_matrices_give_the_same_tracks(cli, tmp_path):
    """The driver's stage 1 in the form it takes with a backend that computes the branch-length scores itself (the GPU): text input is
    measured first (MafReader::measure: reference rows only) and then parsed straight into the batch matrices (fill) -- no strings, no
    packing copy.  Run here on the CPU: the test backend computes the branch-length scores from the matrix rows
    (PCSF_ORACLE_BACKEND_BLS=1) and replaces the scores by a checksum of every codon column (PCSF_ORACLE_BACKEND_FAKE=1: every byte of
    every matrix still decides the seven files).  A generated alignment across the 1 Mb break point, windows of 150 kB and batches of
    ~3,000 columns on three backends against the strings-and-copy form (PCSF_DIRECT_PARSE=0) and the default one-window run."""
    import sys
    syn = tmp_path / "syn.maf"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_synthetic_maf.py"), str(syn), "--bases", "60000", "--start", "970000",
                           "--seed", "13"], stdout=subprocess.DEVNULL)
    model = os.path.join(ROOT, "tests", "golden", "models", "58mammals")
    env = dict(os.environ, PCSF_ORACLE_BACKEND_FAKE="1")
    outs = {}
    for name, extra, more in (("plain", [], {}),
                              ("direct", ["--devices", "0,0,0", "--batch-sites", "6000", "--window-bytes", "150000", "--threads", "5"], {"PCSF_ORACLE_BACKEND_BLS": "1"}),
                              ("strings", ["--devices", "0,0,0", "--batch-sites", "6000", "--window-bytes", "150000", "--threads", "5"],
                               {"PCSF_ORACLE_BACKEND_BLS": "1", "PCSF_DIRECT_PARSE": "0"})):
        out = tmp_path / name
        subprocess.check_call([cli, "build-tracks", "--output", str(out)] + extra + [model, str(syn)], stderr=subprocess.DEVNULL, env=dict(env, **more))
        outs[name] = out
    names = sorted(os.listdir(outs["plain"]))
    assert len(names) == 7
    for n in names:
        want = open(outs["plain"] / n, "rb").read()
        assert len(want) > 10000
        assert open(outs["direct"] / n, "rb").read() == want, n
        assert open(outs["strings"] / n, "rb").read() == want, n


def test_several_backends_small_batches_and_windows_do_not_change_the_output(cli, tmp_path):
    """build-tracks with three scoring backends (--devices; one per GPU in production), batches of ~2,000 columns and
    100 kB windows -- dozens of batches retired in order behind concurrent backends -- against the one-batch run; and
    score-msa with its block list cut into three ranges."""
    import gzip
    import shutil
    maf = tmp_path / "g.maf"
    with gzip.open(os.path.join(ROOT, "tests", "golden", "build-tracks", "galGal6_chr22_25_28_each_30k_bases.maf.gz"), "rt") as fi:
        blocks = fi.read().split("\n\n")
    maf.write_text("\n\n".join(blocks[:1500]) + "\n\n")
    outs = []
    for extra in ([], ["--devices", "0,0,0", "--batch-sites", "4000", "--window-bytes", "100000", "--threads", "3"]):
        out = tmp_path / ("o%d" % len(outs))
        # (the order of the pipeline's stages is what is tested: the test backend's checksum "scores" -- every byte of every batch matrix
        # still decides the files -- make the two runs take seconds instead of a minute)
        subprocess.check_call([cli, "build-tracks", "--output", str(out)] + extra + [MODEL, str(maf)], stderr=subprocess.DEVNULL,
                              env=dict(os.environ, PCSF_ORACLE_BACKEND_FAKE="1"))
        outs.append(out)
    names = sorted(os.listdir(outs[0]))
    assert len(names) == 7
    for n in names:
        assert open(outs[0] / n, "rb").read() == open(outs[1] / n, "rb").read(), n
    src = os.path.join(ROOT, "tests", "golden", "score-msa", "chr22.50alignments.maf")
    res = []
    for i, extra in enumerate(([], ["--devices", "0,0,0"])):
        m = tmp_path / ("s%d.maf" % i)
        shutil.copyfile(src, m)
        # (--strategy fixed: the cut of the block list is what is tested here; MLE / OMEGA lists cut across devices: tests/test_gpu_cli.py)
        subprocess.check_call([cli, "score-msa", "--strategy", "fixed", "--comp-anc", "1"] + extra +
                              [os.path.join(ROOT, "tests", "golden", "models", "100vertebrates"), str(m)], stderr=subprocess.DEVNULL)
        res.append(open(str(m) + ".scores").read().splitlines()[1:])
    assert res[0] == res[1] and len(res[0]) == 51


def test_packed_alignment_file_gives_the_same_tracks(cli, tmp_path):
    """pack-maf stores what the MAF reader yields in build-tracks mode (packed_reader.hpp); build-tracks on the packed file
    -- whole, and cut into small windows and byte-range jobs -- must write the same seven files as on the text, and a
    packed file of another model or a truncated one must be refused."""
    import gzip
    maf = tmp_path / "g.maf"
    with gzip.open(os.path.join(ROOT, "tests", "golden", "build-tracks", "galGal6_chr22_25_28_each_30k_bases.maf.gz"), "rt") as fi:
        blocks = fi.read().split("\n\n")
    maf.write_text("\n\n".join(blocks[:1200]) + "\n\n")
    packed = tmp_path / "g.pcsfbin"
    r = subprocess.run([cli, "pack-maf", MODEL, str(maf), str(packed)], capture_output=True, text=True)
    assert r.returncode == 0 and "alignments" in r.stderr
    assert open(packed, "rb").read(8) == b"PCSFBIN1" and os.path.getsize(packed) < os.path.getsize(maf)
    outs = []
    for src, extra in ((maf, []), (packed, []), (packed, ["--window-bytes", "70000", "--batch-sites", "3000", "--threads", "3", "--devices", "0,0"])):
        out = tmp_path / ("o%d" % len(outs))
        subprocess.check_call([cli, "build-tracks", "--output", str(out)] + extra + [MODEL, str(src)], stderr=subprocess.DEVNULL)
        outs.append(out)
    names = sorted(os.listdir(outs[0]))
    assert len(names) == 7
    for n in names:
        want = open(outs[0] / n, "rb").read()
        assert len(want) > 1000 and open(outs[1] / n, "rb").read() == want and open(outs[2] / n, "rb").read() == want, n
    other = subprocess.run([cli, "build-tracks", "--output", str(tmp_path / "x"), os.path.join(ROOT, "tests", "golden", "models", "58mammals"), str(packed)],
                           capture_output=True, text=True)
    assert other.returncode != 0 and "species" in other.stderr
    cut = tmp_path / "cut.pcsfbin"
    cut.write_bytes(open(packed, "rb").read()[:-100])
    bad = subprocess.run([cli, "build-tracks", "--output", str(tmp_path / "y"), MODEL, str(cut)], capture_output=True, text=True)
    assert bad.returncode != 0 and "truncated" in bad.stderr
    # the same tree reduced to other species of the same count, or another --mapping: rows would land elsewhere -- refused by the header's
    # row-map hash instead of silently mis-scored
    labels = [l.split("\t")[0] for l in open(os.path.join(ROOT, "tests", "golden", "models", "species_aliases.tsv")) if l.strip()]
    import phylocsfpp_amd as pa
    leaves = pa.load_model(MODEL).labels[:pa.load_model(MODEL).nl]
    sel_a, sel_b = ",".join(leaves[:20]), ",".join(leaves[1:21])
    pa_a = tmp_path / "a.pcsfbin"
    assert subprocess.run([cli, "pack-maf", "--species", sel_a, MODEL, str(maf), str(pa_a)], capture_output=True).returncode == 0
    ok = subprocess.run([cli, "build-tracks", "--output", str(tmp_path / "za"), "--species", sel_a, MODEL, str(pa_a)], capture_output=True, text=True)
    assert ok.returncode == 0, ok.stderr
    wrong = subprocess.run([cli, "build-tracks", "--output", str(tmp_path / "zb"), "--species", sel_b, MODEL, str(pa_a)], capture_output=True, text=True)
    assert wrong.returncode != 0 and "another model" in wrong.stderr
    # a corrupted index or record is an error, never a read outside the file
    raw = bytearray(open(packed, "rb").read())
    import struct
    n_rec, index_at = struct.unpack_from("<QQ", raw, len(raw) - 24)
    assert n_rec >= 2
    for what, patch in (("offset beyond the data", lambda b: struct.pack_into("<Q", b, index_at + 8, index_at + 4096)),
                        ("unaligned offset", lambda b: struct.pack_into("<Q", b, index_at + 8, struct.unpack_from("<Q", b, index_at + 8)[0] + 3)),
                        ("chromosome length", lambda b: struct.pack_into("<I", b, struct.unpack_from("<Q", b, index_at)[0] + 4, 0x7fffffff)),
                        ("row count", lambda b: struct.pack_into("<I", b, struct.unpack_from("<Q", b, index_at)[0] + 28, 7)),
                        ("record length", lambda b: struct.pack_into("<I", b, struct.unpack_from("<Q", b, index_at)[0] + 24, 0x7fffffff))):
        b = bytearray(raw)
        patch(b)
        f = tmp_path / "corrupt.pcsfbin"
        f.write_bytes(bytes(b))
        r = subprocess.run([cli, "build-tracks", "--output", str(tmp_path / "c"), MODEL, str(f)], capture_output=True, text=True)
        assert r.returncode == 1 and "packed alignment file" in r.stderr, (what, r.returncode, r.stderr[-300:])


def test_one_process_per_rank_writes_the_files_of_a_single_process(cli, tmp_path):
    """build-tracks --ranks N: the launcher starts N copies of the command line, rank r scores the r-th of N byte ranges of the input (the
    reference's byte-range jobs, src/phylocsf++build_tracks.hpp:88), rank 0 owns the files and appends the others' text in rank order
    (merge_job_output_files, :27-53).  The product's gather is RCCL between GPUs (tests/test_gpu_ranks.py); here the test command line's
    file-based gather carries the same driver code: byte ranges, batch order keys, the merge.  Small batches and windows so that every
    rank keeps dozens of batches; both forms of stage 1; rank 0's own text through the gather as well (--gather-root-text 1)."""
    import gzip
    import sys
    maf = tmp_path / "g.maf"
    with gzip.open(os.path.join(ROOT, "tests", "golden", "build-tracks", "galGal6_chr22_25_28_each_30k_bases.maf.gz"), "rt") as fi:
        blocks = fi.read().split("\n\n")
    maf.write_text("\n\n".join(blocks[:1500]) + "\n\n")
    env = dict(os.environ, PCSF_ORACLE_BACKEND_FAKE="1")
    plain = tmp_path / "plain"
    subprocess.check_call([cli, "build-tracks", "--output", str(plain), MODEL, str(maf)], stderr=subprocess.DEVNULL, env=env)
    names = sorted(os.listdir(plain))
    assert len(names) == 7
    small = ["--batch-sites", "4000", "--window-bytes", "100000", "--threads", "3"]
    for tag, extra, more in (("r2", ["--ranks", "2"] + small, {}),
                             ("r3", ["--ranks", "3"] + small, {"PCSF_ORACLE_BACKEND_BLS": "1"}),
                             ("r7", ["--ranks", "7"], {}),
                             ("r1root", ["--ranks", "1", "--gather-root-text", "1"] + small, {}),
                             ("r3root", ["--ranks", "3", "--gather-root-text", "1"] + small, {"PCSF_ORACLE_BACKEND_BLS": "1"})):
        out = tmp_path / tag
        err = subprocess.run([cli, "build-tracks", "--output", str(out)] + extra + [MODEL, str(maf)], stderr=subprocess.PIPE, env=dict(env, **more), check=True, text=True).stderr
        assert sorted(os.listdir(out)) == names, tag
        for n in names:
            assert open(out / n, "rb").read() == open(plain / n, "rb").read(), (tag, n)
        n_ranks = int(extra[1])
        assert all("[rank %d of %d" % (r, n_ranks) in err for r in range(n_ranks)), err
    # two input files: one gather per file, the second appended behind the first
    syn = tmp_path / "syn.maf"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_synthetic_maf.py"), str(syn), "--bases", "30000", "--start", "985000", "--seed", "5"],
                          stdout=subprocess.DEVNULL)
    model58 = os.path.join(ROOT, "tests", "golden", "models", "58mammals")
    outs = []
    for extra in ([], ["--ranks", "3", "--batch-sites", "6000", "--window-bytes", "150000"]):
        out = tmp_path / ("two%d" % len(outs))
        subprocess.check_call([cli, "build-tracks", "--output", str(out)] + extra + [model58, str(syn), str(syn)], stderr=subprocess.DEVNULL, env=env)
        outs.append(out)
    for n in names:
        a = open(outs[0] / n, "rb").read()
        assert a == open(outs[1] / n, "rb").read() and len(a) > 1000, n
    # a packed alignment file cut into the ranks' ranges
    packed = tmp_path / "syn.pcsfbin"
    subprocess.check_call([cli, "pack-maf", model58, str(syn), str(packed)], stderr=subprocess.DEVNULL)
    one = tmp_path / "one"
    subprocess.check_call([cli, "build-tracks", "--output", str(one), model58, str(syn)], stderr=subprocess.DEVNULL, env=env)
    out = tmp_path / "packed_r3"
    subprocess.check_call([cli, "build-tracks", "--output", str(out), "--ranks", "3", "--batch-sites", "6000", "--window-bytes", "100000", model58, str(packed)],
                          stderr=subprocess.DEVNULL, env=env)
    for n in names:
        assert open(out / n, "rb").read() == open(one / n, "rb").read(), n


def test_a_failing_rank_ends_the_run(cli, tmp_path):
    """A rank that fails (here: a character the reference rejects, in the last third of the file) would leave the others waiting in the
    gather for ever: the launcher ends them and reports the failure; rank 0's unfinished files are removed."""
    import gzip
    import time
    maf = tmp_path / "g.maf"
    with gzip.open(os.path.join(ROOT, "tests", "golden", "build-tracks", "galGal6_chr22_25_28_each_30k_bases.maf.gz"), "rt") as fi:
        blocks = fi.read().split("\n\n")[:900]
    lines = blocks[800].split("\n")
    row = lines[1].split()
    row[-1] = row[-1][:5] + "!" + row[-1][6:]
    lines[1] = " ".join(row)
    blocks[800] = "\n".join(lines)
    maf.write_text("\n\n".join(blocks) + "\n\n")
    out = tmp_path / "out"
    t0 = time.time()
    r = subprocess.run([cli, "build-tracks", "--output", str(out), "--ranks", "3", MODEL, str(maf)], stderr=subprocess.PIPE, text=True,
                       env=dict(os.environ, PCSF_ORACLE_BACKEND_FAKE="1"))
    assert r.returncode != 0 and "error" in r.stderr, r.stderr
    assert time.time() - t0 < 60
