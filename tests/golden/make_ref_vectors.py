#!/usr/bin/env python3
"""Fixtures made by running the reference ITSELF (tests/golden/ref/).

The reference's likelihood core needs GSL and cannot be built in this image, but the code either side of it does not:
MAF reading and block concatenation (src/parallel_file_reader.hpp), six-frame translation (:61-113, src/translation.hpp),
the branch-length score and power track (src/additional_scores.hpp, src/phylocsf++build_tracks.hpp:139-158), newick
flattening / --species reduction / name map (src/newick.hpp, src/models.hpp:1780-1860), the HMM smoothing pass
(src/create_tracks.hpp) and the HMM parameter estimation (src/estimate_hmm_parameter.hpp).  oracle/ref_harness.cpp is a
command line around exactly those headers, compiled from /root/reference/src where they lie (`make -C oracle ref`).
This script feeds it generated inputs and stores inputs + what the reference printed; tests/test_reference_vectors.py
holds the product's host code (and, on the GPU box, the device front end) to them byte for byte.

  python tests/golden/make_ref_vectors.py        # needs /root/reference; about a minute

Large dumps are stored as one sha256 per line (`*.sha.tsv.gz`: the leading plain columns, then the digest of the whole
line); small ones in full.
"""
import gzip, hashlib, json, os, subprocess, sys, tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT = os.path.join(HERE, "ref")
HARNESS = os.path.join(ROOT, "oracle", "_ref", "ref_harness")
MODELS = os.path.join(HERE, "models")
TAGS = [s + str(f) for s in "+-" for f in (1, 2, 3)]


def harness(*args):
    """data the harness dumped (the reference's own warnings go to stdout and are dropped)"""
    with tempfile.NamedTemporaryFile() as t:
        subprocess.check_call([HARNESS] + [str(a) for a in args], env=dict(os.environ, REF_HARNESS_OUT=t.name), stdout=subprocess.DEVNULL,
                              stderr=subprocess.DEVNULL)
        return open(t.name).read()


def put(name, text):
    with gzip.GzipFile(os.path.join(OUT, name), "wb", mtime=0) as f:
        f.write(text.encode())


def hashed(text, plain_columns):
    out = []
    for line in text.splitlines():
        out.append("\t".join(line.split("\t")[:plain_columns] + [hashlib.sha256(line.encode()).hexdigest()]))
    return "\n".join(out) + "\n"


def labels_of(prefix):
    import re
    return re.findall(r"[(,]([^(),:]+):", re.sub(r"\s+", "", open(prefix + ".nh").read()))


def corner_maf(rng, labels, concat, blocks=220, cols_max=30, skip_prob=0.25, absent_prob=0.7, start=999_400, chrom_len=2_500_000, ref_only_prob=0.0):
    """Short blocks around the 1 Mb break point with everything the reader has a branch for: contiguous and skipping
    blocks, a second chromosome, reference-gap columns, missing rows, rows of species the model does not know, i/e/q
    lines and comments, lower case, N and '.', and (block mode only) reference rows on the '-' strand."""
    out = ["##maf version=1 scoring=corner", "# a comment"]
    chrom, pos = "chrA", start
    alphabet = np.frombuffer(b"ACGTacgtNn-.", dtype=np.uint8)
    prob = np.array([.2, .2, .2, .2, .03, .03, .03, .03, .01, .01, .05, .01])
    prob /= prob.sum()
    for b in range(blocks):
        if b == blocks - 50:
            chrom, pos = "chrB", 1_999_700 + int(rng.integers(0, 60))   # ... and the 2 Mb one on another chromosome
        elif rng.random() < skip_prob and not (999_800 < pos % 1_000_000 or pos % 1_000_000 < 200):   # contiguous across the break points
            pos += int(rng.integers(1, 40))                  # not contiguous: a new alignment starts
        cols = int(rng.integers(1, cols_max))
        refrow = rng.choice(np.frombuffer(b"ACGTacgtNn", dtype=np.uint8), size=cols, p=[.22, .22, .22, .22, .02, .02, .02, .02, .02, .02])
        if cols > 3 and rng.random() < 0.5:
            refrow[rng.integers(0, cols, size=int(rng.integers(1, 4)))] = ord("-")
        size = int((refrow != ord("-")).sum())
        if size == 0:
            refrow[0] = ord("A")
            size = 1
        out.append("a score=%d.0" % b)
        strand = "-" if (not concat and rng.random() < 0.3) else "+"
        start = pos if strand == "+" else chrom_len - (pos + size)
        out.append("s %s.%s %d %d %s %d %s" % (labels[0], chrom, start, size, strand, chrom_len, refrow.tobytes().decode()))
        for l in range(1, len(labels)):
            if rng.random() < absent_prob or (ref_only_prob and b % 7 == 3 and rng.random() < ref_only_prob):
                continue
            row = rng.choice(alphabet, size=cols, p=prob).tobytes().decode()
            name = labels[l] if rng.random() < 0.9 else labels[l].upper()
            out.append("s %s.scaf%d %d %d %s %d %s" % (name, l, int(rng.integers(0, 10**6)), cols - row.count("-"), "+-"[int(rng.integers(0, 2))], 10**7, row))
            if rng.random() < 0.1:
                out.append("i %s.scaf%d C 0 I 12" % (name, l))
            if rng.random() < 0.05:
                out.append("q %s.scaf%d %s" % (name, l, "9" * cols))
        if rng.random() < 0.2:
            out.append("s unknown_species_%d.chrZ 5 %d + 1000 %s" % (b % 3, cols, "A" * cols))
        if rng.random() < 0.1:
            out.append("e %s.scaf9 100 20 + 5000 I" % labels[-1])
        out.append("")
        pos += size
    return "\n".join(out) + "\n"


def main():
    if not os.path.isdir("/root/reference/src"):
        sys.exit("needs /root/reference")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"])
    os.makedirs(OUT, exist_ok=True)
    work = tempfile.mkdtemp()
    birds = os.path.join(MODELS, "53birds")
    manifest = {"made_by": "tests/golden/make_ref_vectors.py", "harness": "oracle/ref_harness.cpp over /root/reference/src"}

    # ---- 1. MAF reader, six frames, power track -------------------------------------------------------------------------
    rng = np.random.default_rng(20261003)
    labels = labels_of(birds)
    for name, concat in (("corner", True), ("corner_blocks", False)):
        text = corner_maf(rng, labels, concat)
        put(name + ".maf.gz", text)
        path = os.path.join(work, name + ".maf")
        open(path, "w").write(text)
        put(name + ".alignments.tsv.gz", harness("alignments", birds, path, 1, int(concat)))
        if concat:
            put(name + ".frames.sha.tsv.gz", hashed(harness("frames", birds, path), 7))
            harness("power", birds, path, 1, os.path.join(work, "p.wig"))
            put(name + ".power.wig.gz", open(os.path.join(work, "p.wig")).read())
    # the reference's own example alignment (53 birds; its power track is already a golden file)
    gal = os.path.join(work, "gal.maf")
    with gzip.open(os.path.join(HERE, "build-tracks", "galGal6_chr22_25_28_each_30k_bases.maf.gz"), "rb") as fi, open(gal, "wb") as fo:
        fo.write(fi.read())
    put("gal.alignments.sha.tsv.gz", hashed(harness("alignments", birds, gal, 1, 1), 5))
    put("gal.frames.sha.tsv.gz", hashed(harness("frames", birds, gal), 7))
    # the bench generator's output across the 1 Mb break point, 100 rows
    syn = os.path.join(work, "syn.maf")
    manifest["syn"] = {"generator": "tools/gen_synthetic_maf.py", "args": ["--bases", "60000", "--start", "970000", "--seed", "7"], "model": "100vertebrates"}
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_synthetic_maf.py"), syn] + manifest["syn"]["args"], stdout=subprocess.DEVNULL)
    manifest["syn"]["maf_sha256"] = hashlib.sha256(open(syn, "rb").read()).hexdigest()
    put("syn.alignments.sha.tsv.gz", hashed(harness("alignments", "100vertebrates", syn, 1, 1), 5))
    put("syn.frames.sha.tsv.gz", hashed(harness("frames", "100vertebrates", syn), 7))
    harness("power", "100vertebrates", syn, 1, os.path.join(work, "p.wig"))
    put("syn.power.wig.gz", open(os.path.join(work, "p.wig")).read())
    # the score-msa test files in block mode
    for name, src in (("blocks50", os.path.join(HERE, "score-msa", "chr22.50alignments.maf")), ("blocks516", os.path.join(HERE, "score-msa", "chr22.516alignments.maf.gz"))):
        path = os.path.join(work, name + ".maf")
        with (gzip.open if src.endswith(".gz") else open)(src, "rb") as fi, open(path, "wb") as fo:
            fo.write(fi.read())
        put(name + ".alignments.sha.tsv.gz", hashed(harness("alignments", "100vertebrates", path, 1, 0), 5))

    # ---- 2. trees: flattening, --species reduction, --mapping, the row-name map --------------------------------------------
    rng = np.random.default_rng(20261004)
    cases = [("53birds", birds, "", ""), ("58mammals", "58mammals", "", ""), ("100vertebrates", "100vertebrates", "", "")]
    l100 = labels_of(os.path.join(MODELS, "100vertebrates"))
    cases += [("100vertebrates", "100vertebrates", s, "") for s in ("hg38,Mouse,galGal6", "human,chimp", "Dog,Cat,Cow,Horse,Rat,mm10", "danRer11,Human")]
    for k in (2, 3, 5, 10, 25, 50, 80, 99, 100):
        cases.append(("100vertebrates", "100vertebrates", ",".join(rng.choice(l100, size=k, replace=False)), ""))
    l58 = labels_of(os.path.join(MODELS, "58mammals"))
    for k in (2, 7, 30, 57):
        cases.append(("58mammals", "58mammals", ",".join(rng.choice(l58, size=k, replace=False)), ""))
    for k in (2, 9, 40):
        cases.append(("53birds", birds, ",".join(rng.choice(labels, size=k, replace=False)), ""))
    mapping = os.path.join(OUT, "mapping.tsv")
    # (the reference looks mapping-file names up as written: lower case is what works there; the product also accepts other case)
    open(mapping, "w").write("human\tmyhuman1\nchicken\tbigbird\nmouse\tmm999\nhuman\thg38\n")
    cases.append(("100vertebrates", "100vertebrates", "myHuman1,bigbird,mm999,Rat", mapping))
    models = []
    for mine, theirs, species, mp in cases:
        models.append({"model": mine, "species": species, "mapping": os.path.basename(mp) if mp else "", "dump": harness("model", theirs, species, mp)})
    put("models.json.gz", json.dumps(models, indent=0))

    # ---- 3. HMM: matrices, parameter estimation, smoothing + regions ----------------------------------------------------------
    rng = np.random.default_rng(20261005)
    hmm = {"matrices": [], "estimates": [], "smoothing": {}}
    backsolved = json.load(open(os.path.join(HERE, "build-tracks", "hmm_params_backsolved.json")))["hmm_params"]
    plist = [backsolved, [0.01, 40.0, 0.5, 0.3, 0.2, 300.0, 5000.0, 30.0], [0.004, 55.0, 0.45, 0.45, 0.10, 800.0, 9000.0, 60.0]]
    for p in plist:
        ps = ",".join("%.17g" % x for x in p)
        hmm["matrices"].append({"params": ps, "init_and_transitions": harness("hmm", ps)})
    for n_exons, genome in ((40, 3_000_000), (120, 123_456_789), (3000, 1_065_365_434), (7000, 3_100_000_000)):
        lines = []
        for chrom in ("chr1", "chr2", "chrUn_x"):
            for strand in "+-":
                for frame in "012":
                    pos = 1000
                    for _ in range(max(2, n_exons // 18)):
                        pos += int(rng.choice([rng.exponential(300), rng.exponential(3000), rng.exponential(60000)])) + 1
                        ln = int(rng.integers(30, 600))
                        lines.append("%s\t%s\t%s\t%d\t%d" % (chrom, strand, frame, pos, pos + ln))
                        if rng.random() < 0.15:       # nested / overlapping exons: one of the two is dropped
                            lines.append("%s\t%s\t%s\t%d\t%d" % (chrom, strand, frame, pos + 5, pos + int(rng.integers(10, 900))))
                        if rng.random() < 0.05:       # touching exons: no gap is recorded
                            lines.append("%s\t%s\t%s\t%d\t%d" % (chrom, strand, frame, pos + ln + 1, pos + ln + 40))
                        pos += ln
        rng.shuffle(lines)
        text = "\n".join(lines) + "\n"
        path = os.path.join(work, "exons.txt")
        open(path, "w").write(text)
        name = "exons_%d.txt.gz" % n_exons
        put(name, text)
        hmm["estimates"].append({"exons": name, "genome_length": genome, "params": harness("estimate", path, genome).strip()})
    # smoothing and regions: (a) the golden raw tracks with the back-solved parameters (digests), (b) random tracks with edge cases (in full)
    din, dout = os.path.join(work, "in"), os.path.join(work, "out")
    os.makedirs(din), os.makedirs(dout)
    for t in TAGS:
        with gzip.open(os.path.join(HERE, "build-tracks", "PhyloCSFRaw%s.wig.gz" % t), "rb") as fi, open(os.path.join(din, "PhyloCSFRaw%s.wig" % t), "wb") as fo:
            fo.write(fi.read())
    ps = ",".join("%.17g" % x for x in backsolved)
    harness("smooth", ps, din, dout, 1)
    hmm["smoothing"]["golden_raw_tracks"] = {"params": ps, "sha256": {("PhyloCSF%s%s" % (t, e)): hashlib.sha256(open(os.path.join(dout, "PhyloCSF%s%s" % (t, e)), "rb").read()).hexdigest()
                                                                    for t in TAGS for e in (".wig", "Regions.bed")}}
    ps = ",".join("%.17g" % x for x in plist[1])
    hmm["smoothing"]["random_tracks"] = {"params": ps}
    for t in TAGS:
        lines, pos = [], 1000
        for r in range(14):
            n = int(rng.integers(1, 400)) if r else 1            # the first run is a single value
            vals = rng.normal(-8, 12, size=n)
            if r % 3 == 0:
                vals[n // 3: 2 * n // 3] += 25                   # a coding-looking stretch
            if r == 2:
                vals[:] = rng.normal(12, 3, size=n)              # coding from end to end
            if r == 5:
                vals[:] = rng.normal(200, 30, size=n)            # saturates the posterior
            if r == 7:
                vals[-1] += 60                                   # coding only at the last position
            if r == 9:
                n = 2
                vals = np.array([30.0, -30.0])
            cut = sorted(set([0] + [int(c) for c in rng.integers(0, n, size=2)])) if r % 2 else [0]   # contiguous sections are one run
            for k, v in enumerate(vals):
                if k in cut:
                    lines.append("fixedStep chrom=chrT start=%d step=3 span=3" % (pos + 3 * k))
                s = "%.3f" % float(np.float32(v))
                s = s.rstrip("0")
                lines.append(s + "0" if s.endswith(".") else s)
            pos += 3 * n + int(rng.integers(1, 50)) * 3 + int(rng.integers(0, 3))
        text = "\n".join(lines) + "\n"
        open(os.path.join(din, "PhyloCSFRaw%s.wig" % t), "w").write(text)
        put("smooth_in.PhyloCSFRaw%s.wig.gz" % t, text)
    harness("smooth", ps, din, dout, 1)
    for t in TAGS:
        put("smooth_out.PhyloCSF%s.wig.gz" % t, open(os.path.join(dout, "PhyloCSF%s.wig" % t)).read())
        put("smooth_out.PhyloCSF%sRegions.bed.gz" % t, open(os.path.join(dout, "PhyloCSF%sRegions.bed" % t)).read())
    put("hmm.json.gz", json.dumps(hmm, indent=0))
    json.dump(manifest, open(os.path.join(OUT, "MANIFEST.json"), "w"), indent=1)
    print("wrote", OUT, sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT)), "bytes")


if __name__ == "__main__":
    main()
