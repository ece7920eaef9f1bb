import gzip, os, shutil, subprocess, sys
ROOT = "/root/repo" if os.path.exists("/root/repo/tests") else os.getcwd()
GOLD = os.path.join(ROOT, "tests", "golden")
CLI = os.path.join(ROOT, "phylocsfpp_amd", "host", "phylocsf-amd")
def scores(path): return [l.rstrip("\n").split("\t") for l in open(path) if not l.startswith("#")][1:]
for strategy, anc in (("mle", "1"), ("omega", "0")):
    dst = "/tmp/medium_%s.maf" % strategy
    with gzip.open(os.path.join(GOLD, "score-msa", "chr22.516alignments.maf.gz"), "rb") as fi, open(dst, "wb") as fo: shutil.copyfileobj(fi, fo)
    subprocess.check_call([CLI, "score-msa", "--strategy", strategy, "--comp-anc", anc, os.path.join(GOLD, "models", "100vertebrates"), dst], stderr=subprocess.DEVNULL)
    got = scores(dst + ".scores"); want = scores(os.path.join(GOLD, "score-msa", "chr22.516alignments.%s.scores" % strategy))
    import collections
    for col, name in ((4, "phylo"),) + (((5, "anc"),) if strategy == "mle" else ()):
        d = []
        nan_g = nan_w = 0
        for g, w in zip(got, want):
            if w[col] in ("nan", "-nan"): nan_w += 1; continue
            if g[col] in ("nan", "-nan"): nan_g += 1; continue
            d.append((abs(float(g[col]) - float(w[col])), int(w[2]) - int(w[1]) + 1, g[col] == w[col]))
        d.sort()
        n = len(d)
        print(strategy, name, "n", n, "golden nan", nan_w, "ours nan (golden finite)", nan_g, "identical", sum(x[2] for x in d),
              "<=0.0316:", sum(x[0] <= 0.0316 for x in d), "<=0.316:", sum(x[0] <= 0.316 for x in d), "max", d[-1][:2], "worst5", [(round(x[0], 3), x[1]) for x in d[-5:]])
