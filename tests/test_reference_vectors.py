"""The product's host code (and, on the GPU box, the device front end) against fixtures made by running the REFERENCE itself:
tests/golden/ref/ holds what oracle/ref_harness.cpp -- the reference's own GSL-free headers, compiled where they lie --
printed for generated inputs (tests/golden/make_ref_vectors.py).  Everything here is integer / text work: byte for byte.

  MAF reader / concatenation   src/parallel_file_reader.hpp:283-350,396-685      host/maf_reader.hpp
  six frames                   src/parallel_file_reader.hpp:61-113               csrc/frames.hpp, kernels.hip prologue (GPU)
  power track                  src/additional_scores.hpp:5-84, build_tracks.hpp:139-158   host BLS, frontend.hip bls_kernel (GPU)
  trees, --species, --mapping  src/newick.hpp:170-363, src/models.hpp:1780-1860  host/model_io.hpp
  HMM                          src/create_tracks.hpp:90-314, src/estimate_hmm_parameter.hpp:243-340   host/hmm.hpp
"""
import gzip
import hashlib
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import hmm_oracle as ho

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "tests", "golden", "ref")
GOLD = os.path.join(ROOT, "tests", "golden")
MODELS = os.path.join(GOLD, "models")
CLI = os.path.join(ROOT, "oracle", "_build", "oracle_cli")        # the product's host code over the CPU oracle (CPU tests)
GPU_CLI = os.path.join(ROOT, "phylocsfpp_amd", "host", "phylocsf-amd")
TAGS = [s + str(f) for s in "+-" for f in (1, 2, 3)]


def _ref(name):
    return gzip.open(os.path.join(REF, name), "rt").read()


@pytest.fixture(scope="module")
def cli():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return CLI


@pytest.fixture(scope="module")
def inputs(tmp_path_factory):
    """the MAF files the fixtures were made from, unpacked / regenerated"""
    d = tmp_path_factory.mktemp("refvec")
    paths = {}
    for name, src in (("corner", os.path.join(REF, "corner.maf.gz")), ("corner_blocks", os.path.join(REF, "corner_blocks.maf.gz")),
                      ("gal", os.path.join(GOLD, "build-tracks", "galGal6_chr22_25_28_each_30k_bases.maf.gz")),
                      ("blocks50", os.path.join(GOLD, "score-msa", "chr22.50alignments.maf")),
                      ("blocks516", os.path.join(GOLD, "score-msa", "chr22.516alignments.maf.gz"))):
        paths[name] = str(d / (name + ".maf"))
        with (gzip.open if src.endswith(".gz") else open)(src, "rb") as fi, open(paths[name], "wb") as fo:
            fo.write(fi.read())
    man = json.load(open(os.path.join(REF, "MANIFEST.json")))["syn"]
    paths["syn"] = str(d / "syn.maf")
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_synthetic_maf.py"), paths["syn"]] + man["args"], stdout=subprocess.DEVNULL)
    assert hashlib.sha256(open(paths["syn"], "rb").read()).hexdigest() == man["maf_sha256"], "the generator no longer writes the file the fixture was made from"
    return paths


CASES = {"corner": ("53birds", True), "corner_blocks": ("53birds", False), "gal": ("53birds", True), "syn": ("100vertebrates", True),
         "blocks50": ("100vertebrates", False), "blocks516": ("100vertebrates", False)}


def _dump(case, path, jobs=1):
    model, concat = CASES[case]
    tail = ([] if concat else ["--blocks", "1"]) + [os.path.join(MODELS, model), path]
    out = subprocess.check_output([CLI, "dump-alignments", "--jobs", str(jobs)] + tail, text=True)
    # the reader's second form (alignments measured, then parsed straight into a matrix: build-tracks with text input) gives the same rows
    assert subprocess.check_output([CLI, "dump-alignments", "--matrix", "1", "--jobs", str(jobs)] + tail, text=True) == out
    return out


def _hashed(text, plain):
    return "".join("\t".join(l.split("\t")[:plain] + [hashlib.sha256(l.encode()).hexdigest()]) + "\n" for l in text.splitlines())


# ---- MAF reader ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", sorted(CASES))
def test_alignments_equal_the_references(case, inputs, cli):
    got = _dump(case, inputs[case])
    if case.startswith("corner"):
        want = _ref(case + ".alignments.tsv.gz")
        assert len(want.splitlines()) > (20 if case == "corner" else 200)
        assert got == want
    else:
        assert _hashed(got, 5) == _ref(case + ".alignments.sha.tsv.gz")
    if CASES[case][1]:            # byte-range jobs (the reference: page-aligned ranges) yield the same alignments
        for jobs in (2, 5, 13):
            assert _dump(case, inputs[case], jobs) == got, jobs


# ---- six frames -----------------------------------------------------------------------------------------------------------------
DNA = np.full(256, 4, np.uint8)
for _k, _v in (("A", 0), ("C", 1), ("G", 2), ("T", 3)):
    DNA[ord(_k)] = DNA[ord(_k.lower())] = _v
COMP = str.maketrans("ACGTacgt", "TGCAtgca")


def six_frames(rows, start_pos, chrom_len):
    """Plain restatement of update_seqs / translate / the reverse complement between the strands; yields per frame
    (strand, frame, skip, position of the first codon as the raw track prints it, codons uint8 [rows, n])."""
    L = len(rows[0])
    for strand in "+-":
        b = np.stack([np.frombuffer(r.encode(), np.uint8) for r in rows])
        for frame in (1, 2, 3):
            skip = (frame - start_pos) % 3 if strand == "+" else (frame - (chrom_len - (start_pos + L) + 2)) % 3
            skip = min(skip, L)
            n = (L - skip) // 3
            d = DNA[b[:, skip:skip + 3 * n]].reshape(len(rows), n, 3).astype(np.int64)
            cod = (16 * d[:, :, 0] + 4 * d[:, :, 1] + d[:, :, 2]).astype(np.uint8)
            cod[(d == 4).any(axis=2)] = 64
            yield strand, frame, skip, (start_pos + skip if strand == "+" else start_pos + (L - skip) % 3), cod
        rows = [r[::-1].translate(COMP) for r in rows]


@pytest.mark.parametrize("case", ["corner", "gal", "syn"])
def test_six_frame_restatement_equals_the_references_translation(case, inputs, cli):
    want = _ref(case + ".frames.sha.tsv.gz").splitlines()
    k = 0
    for line in _dump(case, inputs[case]).splitlines():
        f = line.split("\t")
        for strand, frame, skip, first, cod in six_frames(f[5:], int(f[1]), int(f[4])):
            mine = "\t".join([f[0], f[1], strand, str(frame), str(skip), str(first), str(cod.shape[1])] + [row.tobytes().hex() for row in cod])
            assert "\t".join(mine.split("\t")[:7] + [hashlib.sha256(mine.encode()).hexdigest()]) == want[k], (case, f[1], strand, frame)
            k += 1
    assert k == len(want) and k >= 36


# ---- power track (branch-length score) ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("case", ["corner", "syn"])
def test_power_track_equals_the_references(case, inputs, cli, tmp_path):
    for threads in (1, 3):
        out = tmp_path / ("o%d" % threads)
        subprocess.check_call([CLI, "build-tracks", "--threads", str(threads), "--output", str(out), "--output-raw-phylo", "0", os.path.join(MODELS, CASES[case][0]), inputs[case]],
                              stderr=subprocess.DEVNULL)
        assert open(out / "PhyloCSFpower.wig").read() == _ref(case + ".power.wig.gz")


# ---- trees ------------------------------------------------------------------------------------------------------------------------
def test_flattened_trees_species_reduction_and_name_map_equal_the_references(cli):
    cases = json.loads(_ref("models.json.gz"))
    assert len(cases) >= 24
    for c in cases:
        cmd = [CLI, "dump-model"]
        if c["species"]:
            cmd += ["--species", c["species"]]
        if c["mapping"]:
            cmd += ["--mapping", os.path.join(REF, c["mapping"])]
        got = subprocess.check_output(cmd + [os.path.join(MODELS, c["model"])], text=True).splitlines()
        want = c["dump"].splitlines()
        nodes = ["\t".join(f[:4] + f[5:]) for f in (l.split("\t") for l in got) if f[0] != "name"]   # drop the double-precision column the reference does not keep
        assert nodes == [l for l in want if not l.startswith("name\t")], (c["model"], c["species"])
        names = {l for l in got if l.startswith("name\t")}
        want_names = {l for l in want if l.startswith("name\t")}
        assert names == want_names, (c["model"], c["species"], sorted(want_names ^ names)[:5])   # the row names a MAF may use -> leaf
        assert len(want_names) >= 2


# ---- HMM ----------------------------------------------------------------------------------------------------------------------------
def test_hmm_matrices_of_the_python_restatement_equal_the_references():
    for m in json.loads(_ref("hmm.json.gz"))["matrices"]:
        init, trans = ho.make_hmm([float(x) for x in m["params"].split(",")])
        rows = [[float(x) for x in l.split("\t")] for l in m["init_and_transitions"].splitlines()]
        assert list(init) == rows[0]
        assert [list(r) for r in trans] == rows[1:]


def test_hmm_parameter_estimation_equals_the_references(cli, tmp_path):
    cases = json.loads(_ref("hmm.json.gz"))["estimates"]
    assert len(cases) >= 4
    d = tmp_path / "o"
    d.mkdir()
    (d / "PhyloCSFRaw+1.wig").write_text("fixedStep chrom=c start=1 step=3 span=3\n1.0\n-3.5\n")
    for t in TAGS[1:]:
        (d / ("PhyloCSFRaw%s.wig" % t)).write_text("")
    for c in cases:
        exons = tmp_path / "exons.txt"
        exons.write_text(_ref(c["exons"]))
        r = subprocess.run([CLI, "smooth-tracks", "--genome-length", str(c["genome_length"]), "--coding-exons", str(exons), str(d)], capture_output=True, text=True, check=True)
        got = re.search(r"HMM parameters: (.*)", r.stderr).group(1).strip()
        assert got == c["params"], c["exons"]                     # %.17g text: the same doubles
        # and the Python restatement the other HMM tests lean on
        np.testing.assert_allclose(ho.estimate_params(_ref(c["exons"]).splitlines(), c["genome_length"]), [float(x) for x in c["params"].split(",")], rtol=1e-12)


def test_smoothed_tracks_and_regions_equal_the_references(cli, tmp_path):
    sm = json.loads(_ref("hmm.json.gz"))["smoothing"]
    # (a) random tracks with the edge cases (single value, saturation, coding only at an end, sections that continue a run)
    for t in TAGS:
        (tmp_path / ("PhyloCSFRaw%s.wig" % t)).write_text(_ref("smooth_in.PhyloCSFRaw%s.wig.gz" % t))
    subprocess.check_call([CLI, "smooth-tracks", "--output-phylo", "1", "--output-regions", "1", "--hmm-params", sm["random_tracks"]["params"], str(tmp_path)])
    n_regions = 0
    for t in TAGS:
        assert open(tmp_path / ("PhyloCSF%s.wig" % t)).read() == _ref("smooth_out.PhyloCSF%s.wig.gz" % t), t
        bed = _ref("smooth_out.PhyloCSF%sRegions.bed.gz" % t)
        assert open(tmp_path / ("PhyloCSF%sRegions.bed" % t)).read() == bed, t
        n_regions += len(bed.splitlines())
    assert n_regions >= 12
    # (b) the reference's golden raw tracks (174,648 values) with the back-solved parameters: what the reference's own pass makes of them
    d = tmp_path / "golden"
    d.mkdir()
    for t in TAGS:
        with gzip.open(os.path.join(GOLD, "build-tracks", "PhyloCSFRaw%s.wig.gz" % t), "rb") as fi, open(d / ("PhyloCSFRaw%s.wig" % t), "wb") as fo:
            fo.write(fi.read())
    subprocess.check_call([CLI, "smooth-tracks", "--output-phylo", "1", "--output-regions", "1", "--hmm-params", sm["golden_raw_tracks"]["params"], str(d)])
    for name, digest in sm["golden_raw_tracks"]["sha256"].items():
        assert hashlib.sha256(open(d / name, "rb").read()).hexdigest() == digest, name


# ---- the device front end against the same fixtures ---------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("case", ["corner", "syn"])
def test_gpu_power_track_and_frame_positions_equal_the_references(case, inputs, tmp_path):
    """build-tracks on the GPU: the power track comes from bls_kernel, the raw tracks' fixedStep positions and value counts
    from the fused six-frame prologue; with --power-threshold 0 every codon is printed, so each frame's positions are the reference's."""
    out = tmp_path / "o"
    subprocess.check_call([GPU_CLI, "build-tracks", "--output", str(out), "--power-threshold", "0", os.path.join(MODELS, CASES[case][0]), inputs[case]], stderr=subprocess.DEVNULL)
    assert open(out / "PhyloCSFpower.wig").read() == _ref(case + ".power.wig.gz")
    want = {t: [] for t in TAGS}
    for l in _ref(case + ".frames.sha.tsv.gz").splitlines():
        f = l.split("\t")
        if int(f[6]):
            want[f[2] + f[3]].append((f[0], int(f[5]), int(f[6])))
    for t in TAGS:
        got, n = [], 0
        for l in open(out / ("PhyloCSFRaw%s.wig" % t)):
            if l[0] == "f":
                m = re.match(r"fixedStep chrom=(\S+) start=(\d+) step=3 span=3", l)
                got.append([m.group(1), int(m.group(2)), 0])
            else:
                got[-1][2] += 1
        assert [tuple(g) for g in got] == want[t], (case, t)
