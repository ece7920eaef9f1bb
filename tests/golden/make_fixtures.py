#!/usr/bin/env python3
"""Regenerate tests/golden/ from the read-only reference checkout (this container only).

Everything written here is DATA the reference ships (model parameter files, tree strings,
species-alias table, test MAFs and the expected outputs of its own test-suite,
/root/reference/test/tests.sh) -- no reference source text is copied.  The GPU box never has
/root/reference; it uses the committed files.

  python tests/golden/make_fixtures.py [/root/reference]
"""
import gzip, os, re, shutil, sys

REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def copy(rel, dst=None):
    dst = os.path.join(OUT, dst or os.path.basename(rel))
    os.makedirs(os.path.dirname(dst), exist_ok=True)
    shutil.copyfile(os.path.join(REF, rel), dst)
    os.chmod(dst, 0o644)


# (1) file-based model used by the reference's build-tracks test (test/tests.sh:15-17)
for f in ("53birds_coding.ECM", "53birds_noncoding.ECM", "53birds.nh"):
    copy("test/" + f, "models/" + f)

# (2) built-in models: tree strings (src/models.hpp:533,793,1053) and a check that their rate
#     tables equal the 53birds ECM files (the five vertebrate models share them)
src = open(os.path.join(REF, "src/models.hpp")).read()


def literal(name):
    m = re.search(r"double\s+%s\[[^\]]*\]\s*=\s*\{([^}]*)\}" % name, src)
    return [float(x) for x in m.group(1).replace("\n", " ").split(",") if x.strip()]


def ecm_file(path):
    lines = open(path).read().split("\n")
    tri = [float(x) for l in lines[:63] for x in l.split()]
    freq = [float(x) for x in lines[64].split()]
    return tri, freq


ctri, cfreq = ecm_file(os.path.join(REF, "test/53birds_coding.ECM"))
ntri, nfreq = ecm_file(os.path.join(REF, "test/53birds_noncoding.ECM"))
for model in ("58mammals", "100vertebrates"):
    tree = re.search(r'std::string g_%s_tree\s*=\s*"([^"]*)"' % model, src).group(1)
    with open(os.path.join(OUT, "models", model + ".nh"), "w") as f:
        f.write(tree + "\n")
    assert literal("g_%s_cmatrix" % model) == ctri, model
    assert literal("g_%s_ncmatrix" % model) == ntri, model
    assert literal("g_%s_cfreq" % model) == cfreq, model
    assert literal("g_%s_ncfreq" % model) == nfreq, model
    # the ECM files are shared: 58mammals_coding.ECM etc. are the 53birds files under another name
    for kind in ("coding", "noncoding"):
        shutil.copyfile(os.path.join(OUT, "models", "53birds_%s.ECM" % kind),
                        os.path.join(OUT, "models", "%s_%s.ECM" % (model, kind)))

# (3) species alias table (src/models.hpp:1468-1706), restricted to labels of the trees we ship:
#     "<tree label, lower case>\t<alias> <alias> ..."
labels = set()
for model in ("53birds", "58mammals", "100vertebrates"):
    t = open(os.path.join(OUT, "models", model + ".nh")).read()
    labels |= {x.lower() for x in re.findall(r"[(,]\s*([^:,()\s]+)\s*:", t)}
block = src[src.index("sequence_name_mapping = {"):]
block = block[:block.index("};")]
rows = re.findall(r'\{\s*"([^"]+)"\s*,\s*\{([^}]*)\}\s*\}', block)
with open(os.path.join(OUT, "models", "species_aliases.tsv"), "w") as f:
    for common, aliases in sorted(rows):
        if common in labels:
            f.write(common + "\t" + " ".join(re.findall(r'"([^"]+)"', aliases)) + "\n")

# (4) score-msa fixtures (test/tests.sh:35-47) and the medium set
copy("test/maf-file-small/chr22.50alignments.maf", "score-msa/chr22.50alignments.maf")
for s in ("fixed", "mle", "omega"):
    copy("test/maf-file-small/PhyloCSFpp-results/chr22.50alignments.%s.scores" % s,
         "score-msa/chr22.50alignments.%s.scores" % s)
    copy("test/maf-file-medium/chr22.516alignments.maf.%s.scores" % s,
         "score-msa/chr22.516alignments.%s.scores" % s)
with open(os.path.join(REF, "test/maf-file-medium/chr22.516alignments.maf"), "rb") as fi, \
        gzip.GzipFile(os.path.join(OUT, "score-msa/chr22.516alignments.maf.gz"), "wb", mtime=0) as fo:
    fo.write(fi.read())

# (5) build-tracks fixtures (test/tests.sh:15-19): input MAF (kept gzipped) + the 7 raw/power wigs + the 6 smoothed
# wigs (their HMM parameters came from example/galGal6_coding_exons.txt, which the checkout does not hold:
# tests/golden/build-tracks/hmm_params_backsolved.json holds parameters fitted to these files, see fit_hmm_params.py)
copy("example/galGal6_chr22_25_28_each_30k_bases.maf.gz", "build-tracks/galGal6_chr22_25_28_each_30k_bases.maf.gz")
for w in ["PhyloCSFpower.wig"] + ["PhyloCSFRaw%s%d.wig" % (s, f) for s in "+-" for f in (1, 2, 3)] + \
        ["PhyloCSF%s%d.wig" % (s, f) for s in "+-" for f in (1, 2, 3)]:
    with open(os.path.join(REF, "test/expected_results/build-tracks", w), "rb") as fi, \
            gzip.GzipFile(os.path.join(OUT, "build-tracks", w + ".gz"), "wb", mtime=0) as fo:
        fo.write(fi.read())
print("fixtures written to", OUT)
