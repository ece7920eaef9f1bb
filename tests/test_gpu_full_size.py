"""GPU: BASELINE.json configs[3] and configs[4] at the size ONE GPU of an eight-GPU node sees (100 Mb / 8 = 12.5 Mb of reference
sequence), every output line tied to the oracle through a size-independent property: the input is ONE generated 200 kb segment
repeated along the chromosome at a spacing that is a multiple of three and keeps every copy clear of the reader's 1 Mb break points,
so every copy must produce the text of the first one (positions moved by the spacing) -- and the first copy's text must be what the
oracle-backed driver (the same host code on the CPU oracle) writes for the segment alone.  build-tracks end to end through the
command line: parser windows, batches cut inside copies, two scoring contexts, the device-side text and the device smoothing pass."""
import os
import re
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
CLI = os.path.join(ROOT, "phylocsfpp_amd", "host", "phylocsf-amd")
ORACLE_CLI = os.path.join(ROOT, "oracle", "_build", "oracle_cli")
SEGMENT, START, SHIFT = 200_000, 10_000, 249_999        # 249,999 = 3 x 83,333; copy k covers [10,000 + k SHIFT, 210,000 + k SHIFT): no multiple of 1,000,000 inside for k <= 63


def _copies_of(text, n_copies, kind):
    """The file cut into its copies, positions moved back by k x SHIFT: list of n_copies byte strings."""
    if kind == "wig":
        pat = re.compile(rb"^fixedStep chrom=\S+ start=(\d+)", re.M)
        spans = [(m.start(1), m.end(1), m.start(), int(m.group(1))) for m in pat.finditer(text)]
    else:   # bed: chrom, start, end, chrom:start-end, 0, strand, start, end, colour
        pat = re.compile(rb"^(\S+)\t(\d+)\t(\d+)\t\S+:(\d+)-(\d+)\t0\t.\t(\d+)\t(\d+)\t", re.M)
        spans = []
        for m in pat.finditer(text):
            for g in (2, 3, 4, 5, 6, 7):
                spans.append((m.start(g), m.end(g), m.start(), int(m.group(2))))
    out, pieces, at, cur = [], [], 0, 0
    for a, b, line0, key in spans:
        k = (key - START + 6) // SHIFT        # (+6: a region may start a codon or two before its copy's first scored position)
        assert 0 <= k < n_copies and k >= cur, (k, cur, key)
        if k != cur:
            pieces.append(text[at:line0])
            out.append(b"".join(pieces))
            out += [b""] * (k - cur - 1)
            pieces, at, cur = [], line0, k
        pieces.append(text[at:a])
        pieces.append(str(int(text[a:b]) - k * SHIFT).encode())
        at = b
    pieces.append(text[at:])
    out.append(b"".join(pieces))
    out += [b""] * (n_copies - len(out))
    return out


def _run(cli, args, model, maf, out, env=None):
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(CLI)])
    r = subprocess.run([cli, "build-tracks", "--output", str(out)] + args + [os.path.join(GOLD, "models", model), str(maf)], capture_output=True, text=True, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    return r.stderr


@pytest.mark.parametrize("model,extra,n_files,copies", [
    ("58mammals", [], 7, 63),                                                                   # configs[3]'s share: 12.6 Mb
    ("100vertebrates", ["--output-phylo", "1", "--output-regions", "1", "--hmm-params", "0.004,55,0.45,0.45,0.1,800,9000,60"], 19, 63),   # configs[4]'s
])
def test_one_gpus_share_of_the_node_run_equals_the_oracle_copy_by_copy(tmp_path, model, extra, n_files, copies):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import replicate_maf
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    seg = tmp_path / "segment.maf"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_synthetic_maf.py"), str(seg), "--bases", str(SEGMENT), "--start", str(START), "--seed", "11"],
                          stdout=subprocess.DEVNULL)
    chrom_len = 2 * (START + SEGMENT + copies * SHIFT)
    one, big = tmp_path / "one.maf", tmp_path / "big.maf"
    replicate_maf.tile(str(seg), str(one), 1, SHIFT, chrom_len)
    replicate_maf.tile(str(seg), str(big), copies, SHIFT, chrom_len)
    os.remove(seg)
    log = _run(CLI, extra + ["--devices", "0,0"], model, big, tmp_path / "gpu")
    os.remove(big)
    m = re.search(r"(\d+) alignments, (\d+) reference bases, (\d+) codon-frames", log)
    assert int(m.group(2)) >= copies * SEGMENT * 0.99 and int(m.group(3)) >= 2 * copies * SEGMENT * 0.98, log
    _run(ORACLE_CLI, extra, model, one, tmp_path / "cpu")
    names = sorted(os.listdir(tmp_path / "cpu"))
    assert len(names) == n_files and sorted(os.listdir(tmp_path / "gpu")) == names
    total = 0
    for n in names:
        want = open(tmp_path / "cpu" / n, "rb").read()
        got = open(tmp_path / "gpu" / n, "rb").read()
        assert len(want) > 1000, n
        parts = _copies_of(got, copies, "bed" if n.endswith(".bed") else "wig")
        ref = _copies_of(want, 1, "bed" if n.endswith(".bed") else "wig")[0]
        for k, part in enumerate(parts):
            if part != ref and os.environ.get("PCSF_TEST_DUMP"):
                os.makedirs(os.environ["PCSF_TEST_DUMP"], exist_ok=True)
                open(os.path.join(os.environ["PCSF_TEST_DUMP"], "%s.copy%d.got" % (n, k)), "wb").write(part)
                open(os.path.join(os.environ["PCSF_TEST_DUMP"], "%s.ref" % n), "wb").write(ref)
                open(os.path.join(os.environ["PCSF_TEST_DUMP"], "log.txt"), "w").write(log)
            assert part == ref, (n, k, len(part), len(ref))
        total += len(got)
    assert total > copies * 1_000_000
