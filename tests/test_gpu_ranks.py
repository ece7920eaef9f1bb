"""GPU: build-tracks as one process per GPU -- the text kept on the device and gathered over RCCL (csrc/comm.cpp; the reference's job
files and their merge, src/phylocsf++build_tracks.hpp:88, :27-53, :246-259).  The pool has one GPU per box, so the communicator here has
one rank whose text travels through RCCL TO ITSELF (include_root): the same send / receive / fetch code a sender and the root run,
the same command line a node with eight GPUs would be given with --ranks 8.  The byte ranges, order keys and the merge at world sizes 2-7
are pinned on the CPU (tests/test_host_layer.py)."""
import gzip
import os
import shutil
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
CLI = os.path.join(ROOT, "phylocsfpp_amd", "host", "phylocsf-amd")


def test_kept_text_comes_back_through_rccl_in_key_order(oracle_models):
    import phylocsfpp_amd as pa
    from test_gpu_wig import _rows_with_thin_stretches
    m = oracle_models("53birds")
    ctx = pa.load_model(os.path.join(GOLD, "models", "53birds")).context(device=0)   # a context of its own: it gets a communicator
    try:
        ctx.set_bls_branch_lengths(m.branch_len_d)
        rng = np.random.default_rng(7)
        batches = []
        for b in range(4):
            alns, chroms, cols = [], [], 0
            for i, L in enumerate([5, 64, 193, 1000 + 37 * b]):
                alns.append((cols, L, int(rng.integers(1, 10**6)), int(rng.integers(2 * 10**6, 3 * 10**6))))
                chroms.append("chr%d" % (b + 1))
                cols += L + int(rng.integers(0, 4))
            batches.append((_rows_with_thin_stretches(rng, m.nl, cols), alns, chroms))
        ctx.comm_create(0, 1, pa.ScoringContext.comm_unique_id())
        want = [b""] * 7
        texts = []
        for nuc, alns, chroms in batches:
            got, _ = ctx.score_alignments_6frames_wig(nuc, alns, chroms, 0.1)
            texts.append(got)
        # kept out of order, one of them handed in as host text: the gather sorts by key
        for b in (2, 0, 3):
            ctx.score_alignments_6frames_wig(batches[b][0], batches[b][1], batches[b][2], 0.1, text_capacity=0, retain_order=b)
        ctx.wig_retain_host(1, texts[1])
        for f in range(7):
            want[f] = b"".join(texts[b][f] for b in range(4))
        assert ctx.wig_retained_bytes() == [len(w) for w in want]
        sizes = ctx.comm_gather_wig(root=0, include_root=True)
        assert sizes == [[len(w) for w in want]]
        for f in range(7):
            assert ctx.comm_fetch_gathered(0, f, sizes[0][f]) == want[f], f
        assert ctx.wig_retained_bytes() == [0] * 7
        # a second gather on the same communicator (the next input file); nothing kept by a root that does not send
        ctx.score_alignments_6frames_wig(batches[0][0], batches[0][1], batches[0][2], 0.1, text_capacity=0, retain_order=0)
        assert ctx.comm_gather_wig(root=0, include_root=False) == [[0] * 7]
        ctx.score_alignments_6frames_wig(batches[1][0], batches[1][1], batches[1][2], 0.1, text_capacity=0, retain_order=5)
        sizes = ctx.comm_gather_wig(root=0, include_root=True)
        assert [ctx.comm_fetch_gathered(0, f, sizes[0][f]) for f in range(7)] == texts[1]
    finally:
        ctx.close()


@pytest.mark.parametrize("device_wig", ["1", "0"])
def test_ranks_command_line_writes_the_golden_files(tmp_path, device_wig):
    """phylocsf-amd build-tracks --ranks 1 --gather-root-text 1: the launcher, the rendezvous file, the communicator set up beside the
    scoring, every batch's text kept on the device (device_wig 1) or handed to the device by the host formatter (0), one RCCL gather,
    the files written from what came back -- the reference's golden tracks, byte for byte."""
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(CLI)])
    maf = tmp_path / "galGal6.maf"
    with gzip.open(os.path.join(GOLD, "build-tracks", "galGal6_chr22_25_28_each_30k_bases.maf.gz"), "rb") as fi, open(maf, "wb") as fo:
        shutil.copyfileobj(fi, fo)
    out = tmp_path / "out"
    r = subprocess.run([CLI, "build-tracks", "--output", str(out), "--ranks", "1", "--gather-root-text", "1", "--batch-sites", "20000", "--window-bytes", "400000",
                        os.path.join(GOLD, "models", "53birds"), str(maf)], env=dict(os.environ, PCSF_DEVICE_WIG=device_wig), stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr
    assert "[rank 0 of 1" in r.stderr
    for n in ["PhyloCSFpower"] + ["PhyloCSFRaw%s%d" % (s, f) for s in "+-" for f in (1, 2, 3)]:
        want = gzip.open(os.path.join(GOLD, "build-tracks", n + ".wig.gz"), "rb").read()
        assert open(out / (n + ".wig"), "rb").read() == want, n


def test_ranks_command_line_with_the_smoothing_pass(tmp_path):
    """--ranks with --output-phylo / --output-regions: rank 0 runs the smoothing pass on the gathered raw tracks; same files as one process"""
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(CLI)])
    import sys
    syn = tmp_path / "syn.maf"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_synthetic_maf.py"), str(syn), "--bases", "120000", "--start", "940000", "--seed", "3"],
                          stdout=subprocess.DEVNULL)
    import json
    hmm = ",".join("%.17g" % x for x in json.load(open(os.path.join(GOLD, "build-tracks", "hmm_params_backsolved.json")))["hmm_params"])
    outs = []
    for extra in ([], ["--ranks", "1", "--gather-root-text", "1"]):
        out = tmp_path / ("o%d" % len(outs))
        subprocess.check_call([CLI, "build-tracks", "--output", str(out), "--output-phylo", "1", "--output-regions", "1", "--hmm-params", hmm] + extra +
                              ["58mammals", str(syn)], stderr=subprocess.DEVNULL)
        outs.append(out)
    names = sorted(os.listdir(outs[0]))
    assert len(names) == 19
    for n in names:
        assert open(outs[0] / n, "rb").read() == open(outs[1] / n, "rb").read(), n
