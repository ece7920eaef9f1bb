"""The smoothing pass on the GPU (pcsf_hmm_smooth_runs, csrc/hmm.hip): posterior, Viterbi path and output files must equal the
sequential host algorithm BIT FOR BIT -- the chunked evaluation repeats every chunk whose entry state did not meet its neighbour's
exit state, so no choice of chunk length or burn-in may change a value.  The checker is tests/hmm_oracle.py (plain Python, pinned
in tests/test_hmm.py) and the C++ host pass (phylocsfpp_amd/host/hmm.hpp, byte-identical to the reference's own process_scores:
tests/test_reference_vectors.py)."""
import gzip
import json
import os
import shutil
import subprocess

import numpy as np
import pytest

import hmm_oracle as ho

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
CLI = os.path.join(ROOT, "phylocsfpp_amd", "host", "phylocsf-amd")


def _params():
    return json.load(open(os.path.join(GOLD, "build-tracks", "hmm_params_backsolved.json")))["hmm_params"]


def _runs(rng):
    """runs of raw scores as the tracks hold them (multiples of 0.001): short ones, one long enough for many chunks, one that says
    nothing (constant 0: the filter forgets slowly), one with extreme scores"""
    runs = [np.round(rng.normal(-5.0, 8.0, size=n), 3) for n in (1, 2, 3, 17, 1000, 6000)]
    runs.append(np.zeros(3000))
    runs.append(np.round(rng.choice([-95.0, 80.0, 0.001, -0.0], size=500), 3))
    return runs


@pytest.mark.parametrize("chunk,burn", [(None, None), ("64", "8"), ("7", "1"), ("100000", "5")])
def test_device_recurrences_equal_the_sequential_algorithm_bit_for_bit(chunk, burn, monkeypatch):
    import phylocsfpp_amd as pa
    if chunk:
        monkeypatch.setenv("PCSF_HMM_CHUNK", chunk)
        monkeypatch.setenv("PCSF_HMM_BURN_IN", burn)
    init, trans = ho.make_hmm(_params())
    rng = np.random.default_rng(11)
    runs = _runs(rng)
    offs = np.concatenate([[0], np.cumsum([len(r) for r in runs])]).astype(np.uint64)
    scores = np.concatenate(runs)
    emit = np.array([ho.emit(0, float(s)) for s in scores])
    ctx = pa.load_model("53birds").context()
    post, back, last, stats = ctx.hmm_smooth_runs(init, trans, emit, offs)
    ctx.close()
    if chunk == "64":
        assert stats[1] + stats[3] + stats[5] > 0, "a burn-in of 8 positions should leave chunks to evaluate again"
    if chunk == "100000":
        assert stats[0] == len(runs) and stats[1] + stats[3] + stats[5] == 0   # one chunk per run: nothing to verify
    for r, obs in enumerate(runs):
        b, e = int(offs[r]), int(offs[r + 1])
        want = np.array(ho.coding_posterior(init, trans, [float(x) for x in obs]))
        assert np.array_equal(post[b:e].view(np.uint64), want.view(np.uint64)), "posterior of run %d" % r
        # the path from the device's back pointers and last vector, by hmm_viterbi's rules
        best = 0
        for i in range(1, 4):
            if last[r, i] > last[r, best]:
                best = i
        if not last[r, best] > 0.0:
            best = 0
        path = [best]
        for pos in range(e - 1, b, -1):
            path.append((int(back[pos]) >> (2 * path[-1])) & 3)
        path.reverse()
        assert path == ho.viterbi(init, trans, [float(x) for x in obs]), "Viterbi path of run %d" % r


def _cli():
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(CLI)])
    return CLI


@pytest.mark.parametrize("env", [{}, {"PCSF_HMM_CHUNK": "50", "PCSF_HMM_BURN_IN": "5", "PCSF_WIG_PIECE": "100"}, {"PCSF_HMM_HOST_RUN": "400"}])   # default; many chunks per run and text pieces inside runs; long runs on host threads beside the device call
def test_build_tracks_smooths_on_the_device_and_writes_the_host_pass_files(tmp_path, env):
    """--output-phylo 1 --output-regions 1: the twelve smoothed files from the device recurrences (default) equal those of the host
    pass (PCSF_HOST_HMM=1) byte for byte."""
    maf = tmp_path / "galGal6.maf"
    with gzip.open(os.path.join(GOLD, "build-tracks", "galGal6_chr22_25_28_each_30k_bases.maf.gz"), "rb") as fi, open(maf, "wb") as fo:
        shutil.copyfileobj(fi, fo)
    prm = ",".join("%.17g" % x for x in _params())
    outs = {}
    for name, extra in (("device", env), ("host", {"PCSF_HOST_HMM": "1"})):
        out = tmp_path / name
        e = dict(os.environ, PCSF_HMM_STATS="1", **extra)
        r = subprocess.run([_cli(), "build-tracks", "--output", str(out), "--output-phylo", "1", "--output-regions", "1", "--hmm-params", prm,
                            os.path.join(GOLD, "models", "53birds"), str(maf)], env=e, stderr=subprocess.PIPE, text=True)
        assert r.returncode == 0, r.stderr
        assert ("hmm_smooth_runs:" in r.stderr) == (name == "device"), r.stderr
        outs[name] = out
    names = sorted(os.listdir(outs["host"]))
    assert names == sorted(os.listdir(outs["device"])) and len([n for n in names if n.endswith("Regions.bed")]) == 6
    for n in names:
        assert open(outs["device"] / n, "rb").read() == open(outs["host"] / n, "rb").read(), n


def _write_raw_tracks(d, rng):
    """six raw tracks with runs the wig reader merges or not: short runs, a header that continues the stretch before it, a header
    right behind a header, a second chromosome, a constant stretch, and one run of 300,000 values (ten chunks at the default length)"""
    def fmt(v):
        s = "%.3f" % v
        s = s.rstrip("0")
        return s + "0" if s.endswith(".") else s
    for strand in "+-":
        for f in (1, 2, 3):
            lines, pos = [], 1000 + f
            def run(chrom, n, start, values=None):
                lines.append("fixedStep chrom=%s start=%d step=3 span=3" % (chrom, start))
                vals = values if values is not None else np.round(rng.normal(-5.0, 8.0, size=n), 3)
                lines.extend(fmt(float(v)) for v in vals)
                return start + 3 * n
            pos = run("chrA", 1, pos)
            pos = run("chrA", 2, pos + 30)
            pos = run("chrA", 50, pos + 3)            # a gap of one codon: a new run
            pos = run("chrA", 5000, pos)              # continues the stretch before it: the same run
            lines.append("fixedStep chrom=chrA start=%d step=3 span=3" % (pos + 300))   # a header without values ...
            pos = run("chrA", 300000, pos + 600)      # ... right before the long run
            pos = run("chrA", 70000, pos + 9, values=np.zeros(70000))
            pos = run("chrB", 1200, 7)
            lines.extend(["-0.0", "3.0", "12.5", "-7.25"])   # fewer decimals than three
            lines.extend(["250.5", "-350.125", "1234.5", "1.23456", "1e-3", "-2.5E1", "0.0005"])   # beyond the emission table, more than three decimals, exponents
            pos = run("chrC", 40, 100)
            lines.append("nan")                              # what a NaN score looks like in a raw track: the whole run has no posterior
            lines.extend(fmt(float(v)) for v in np.round(rng.normal(-5.0, 8.0, size=30), 3))
            open(os.path.join(d, "PhyloCSFRaw%s%d.wig" % (strand, f)), "w").write("\n".join(lines) + "\n")


def test_smooth_tracks_on_long_runs_device_against_host(tmp_path):
    """smooth-tracks over generated raw tracks, as host code and with the device recurrences (--context-model): runs of 300,000 values
    are cut into ten chunks at the default chunk length and burn-in -- some meet their neighbours' states, the others are evaluated
    again -- and the twelve files must not differ in a byte."""
    d = tmp_path / "tracks"
    os.makedirs(d)
    _write_raw_tracks(str(d), np.random.default_rng(3))
    prm = ",".join("%.17g" % x for x in _params())
    results = {}
    for name, extra in (("host", []), ("device", ["--context-model", "53birds"])):
        r = subprocess.run([_cli(), "smooth-tracks", "--output-phylo", "1", "--output-regions", "1", "--hmm-params", prm] + extra + [str(d)],
                           env=dict(os.environ, PCSF_HMM_STATS="1"), stderr=subprocess.PIPE, text=True)
        assert r.returncode == 0, r.stderr
        assert ("hmm_smooth_runs:" in r.stderr) == (name == "device"), r.stderr
        out = tmp_path / name
        os.makedirs(out)
        for n in os.listdir(d):
            if not n.startswith("PhyloCSFRaw"):
                shutil.move(str(d / n), str(out / n))
        results[name] = (out, r.stderr)
    names = sorted(os.listdir(results["host"][0]))
    assert len(names) == 12 and names == sorted(os.listdir(results["device"][0]))
    total = 0
    for n in names:
        a, b = open(results["host"][0] / n, "rb").read(), open(results["device"][0] / n, "rb").read()
        assert a == b, n
        total += len(a)
    assert total > 10_000_000
    import re
    m = re.search(r"(\d+) runs, (\d+) values in (\d+) chunks", results["device"][1])
    assert m and int(m.group(3)) > int(m.group(1)) + 40, results["device"][1]   # the long runs were cut into chunks
