"""GPU: the text of build-tracks' seven wig files formatted on the device (csrc/wig.hip) against a plain Python restatement of the
reference's output loop (src/phylocsf++build_tracks.hpp:139-156 power track, :183-217 raw tracks; print_trimmed src/common.hpp:48-68)
fed with the scores and per-base branch-length scores of the SAME call's download variant.  Byte work: identical, not close."""
import os
import struct

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _trimmed(v32, decimals):
    """my_fprintf / print_trimmed: "%.Nf" of the float, trailing zeros removed, one decimal kept"""
    s = ("%%.%df" % decimals) % float(np.float32(v32))
    if s in ("nan", "-nan", "inf", "-inf"):
        return s
    s = s.rstrip("0")
    return s + "0" if s.endswith(".") else s


def _frame_positions(start_pos, chrom_len, L, f):
    """first codon's genome position, number of codons, first column -- frame f (0..5 = +1 +2 +3 -1 -2 -3), forward genome order
    (build_tracks.hpp:162-217: the '-' strand's scores are reversed before they are written)"""
    frame = f % 3 + 1
    if f < 3:
        skip = min((frame - start_pos) % 3, L)
        n = (L - skip) // 3
        return start_pos + skip, n, skip
    skip = min((frame - (chrom_len - (start_pos + L) + 2)) % 3, L)
    n = (L - skip) // 3
    first_col = L - skip - 3 * n          # the reverse strand's last codon is the forward order's first
    return start_pos + first_col, n, first_col


def _expected_text(alns, chroms, res, bls, threshold, want_power=True, want_raw=True):
    files = [[] for _ in range(7)]
    thr3 = np.float32(threshold) * np.float32(3)
    for k, (c0, L, sp, cl) in enumerate(alns):
        b = bls[c0:c0 + L]
        if want_power:
            skip = (3 - sp) % 3
            if skip + 2 < L:
                files[0].append("fixedStep chrom=%s start=%d step=3 span=3\n" % (chroms[k], sp + skip))
                for pos in range(skip, L - 2, 3):
                    files[0].append(_trimmed(np.float32((b[pos] + b[pos + 1] + b[pos + 2]) / 3.0), 4) + "\n")
        if not want_raw:
            continue
        for f in range(6):
            pos0, n, col = _frame_positions(sp, cl, L, f)
            assert n == len(res[k][f])
            prev = -4
            for xx in range(n):
                s = np.float32(b[col + 3 * xx] + b[col + 3 * xx + 1] + b[col + 3 * xx + 2])
                if s < thr3:
                    continue
                pos = pos0 + 3 * xx
                if prev + 3 != pos:
                    files[1 + f].append("fixedStep chrom=%s start=%d step=3 span=3\n" % (chroms[k], pos))
                prev = pos
                files[1 + f].append(_trimmed(np.float32(res[k][f][xx]), 3) + "\n")
    return ["".join(x).encode() for x in files]


def _rows_with_thin_stretches(rng, nl, n_cols):
    """species present over long stretches only: codons on both sides of the branch-length threshold, runs that break and resume"""
    nuc = np.frombuffer(b"ACGTacgt", dtype=np.uint8)[rng.integers(0, 8, size=(nl, n_cols))]
    for l in range(nl):
        for _ in range(int(rng.integers(1, 6))):
            a = int(rng.integers(0, n_cols))
            e = min(n_cols, a + int(rng.integers(1, max(2, n_cols // 2))))
            nuc[l, a:e] = ord("-") if rng.random() < 0.6 else ord("N")
    for _ in range(max(1, n_cols // 150)):                     # stretches where nearly every species is absent
        a = int(rng.integers(0, n_cols))
        e = min(n_cols, a + int(rng.integers(1, 40)))
        keep = rng.choice(nl, size=int(rng.integers(0, 4)), replace=False)
        rows = np.setdiff1d(np.arange(nl), keep)
        nuc[rows, a:e] = ord("-")
    return nuc


@pytest.mark.parametrize("model", ["53birds", "58mammals"])
def test_wig_text_equals_the_output_loop_on_the_downloaded_scores(model, oracle_models, gpu_contexts):
    m = oracle_models(model)
    ctx = gpu_contexts(model)
    ctx.set_bls_branch_lengths(m.branch_len_d)
    rng = np.random.default_rng(31)
    alns, chroms, cols = [], [], 0
    lens = [1, 2, 3, 4, 5, 6, 17, 64, 191, 192, 193, 1000, 4097]
    for i, L in enumerate(lens):
        sp = int(rng.integers(1, 10**6)) if i != 3 else 1      # start positions in all three residues
        alns.append((cols, L, sp + i, int(rng.integers(2 * 10**6, 3 * 10**6))))
        chroms.append(["chr1", "chrUn_KI270742v1", "Z", "scaffold_%d" % i][i % 4])
        cols += L + int(rng.integers(0, 5))
    nuc = _rows_with_thin_stretches(rng, m.nl, cols)
    res, _, bls = ctx.score_alignments_6frames(nuc, alns, with_bls=True)
    for threshold in (0.0, 0.1, 0.35, 0.9, 2.0):
        want = _expected_text(alns, chroms, res, bls, threshold)
        got, frames = ctx.score_alignments_6frames_wig(nuc, alns, chroms, threshold)
        assert frames == sum(len(res[k][f]) for k in range(len(alns)) for f in range(6))
        for f in range(7):
            assert got[f] == want[f], (model, threshold, f, len(got[f]), len(want[f]))
    # runs were in fact broken and resumed inside alignments, and some codons were filtered
    mid, everything = _expected_text(alns, chroms, res, bls, 0.35), _expected_text(alns, chroms, res, bls, 0.0)
    assert mid[1].count(b"fixedStep") > everything[1].count(b"fixedStep") + 5
    assert mid[1].count(b"\n") - mid[1].count(b"fixedStep") < everything[1].count(b"\n") - everything[1].count(b"fixedStep") - 20
    # the caller's buffer: none, too small (the text is fetched afterwards), exactly enough
    total = sum(len(t) for t in _expected_text(alns, chroms, res, bls, 0.1))
    for cap in (0, 100, total - 1, total):
        got, _ = ctx.score_alignments_6frames_wig(nuc, alns, chroms, 0.1, text_capacity=cap)
        assert got == _expected_text(alns, chroms, res, bls, 0.1), cap
    # one family of files only
    got, _ = ctx.score_alignments_6frames_wig(nuc, alns, chroms, 0.1, want_power=False)
    assert got[0] == b"" and got[1:] == _expected_text(alns, chroms, res, bls, 0.1)[1:]
    got, _ = ctx.score_alignments_6frames_wig(nuc, alns, chroms, 0.1, want_raw=False)
    assert got[0] == _expected_text(alns, chroms, res, bls, 0.1)[0] and all(t == b"" for t in got[1:])


def test_wig_text_edge_cases(oracle_models, gpu_contexts):
    m = oracle_models("53birds")
    ctx = gpu_contexts("53birds")
    ctx.set_bls_branch_lengths(m.branch_len_d)
    rng = np.random.default_rng(4)
    # no alignments; only alignments shorter than a codon
    got, frames = ctx.score_alignments_6frames_wig(np.zeros((m.nl, 0), np.uint8), [], [], 0.1)
    assert frames == 0 and all(t == b"" for t in got)
    two = _rows_with_thin_stretches(rng, m.nl, 2)
    got, frames = ctx.score_alignments_6frames_wig(two, [(0, 2, 7, 1000), (0, 1, 9, 1000)], ["a", "b"], 0.0)
    assert frames == 0 and all(t == b"" for t in got)
    # a three-column alignment: one codon in one frame per strand, power track iff it starts a codon of frame 3
    three = np.full((m.nl, 3), ord("A"), np.uint8)
    for sp in (3, 4, 5):
        res, _, bls = ctx.score_alignments_6frames(three, [(0, 3, sp, 1000)], with_bls=True)
        got, frames = ctx.score_alignments_6frames_wig(three, [(0, 3, sp, 1000)], ["chrM"], 0.1)
        assert frames == 2 and got == _expected_text([(0, 3, sp, 1000)], ["chrM"], res, bls, 0.1)
    # a character the reference exits on fails this call like the download variant
    bad = _rows_with_thin_stretches(rng, m.nl, 300)
    bad[5, 77] = ord("#")
    with pytest.raises(RuntimeError, match="unexpected character"):
        ctx.score_alignments_6frames_wig(bad, [(0, 300, 1, 1000)], ["c"], 0.1)
    bad[5, 77] = ord("a")
    got, frames = ctx.score_alignments_6frames_wig(bad, [(0, 300, 1, 1000)], ["c"], 0.1)     # the context stays usable
    assert frames == 596 and got[0].startswith(b"fixedStep chrom=c start=3 step=3 span=3\n")


@pytest.mark.parametrize("decimals", [3, 4])
def test_device_number_text_equals_printf(decimals, oracle_models, gpu_contexts):
    """format_value (csrc/wig.hip) against "%.3f" / "%.4f" + trimming on floats of every magnitude a track can hold and beyond:
    exact ties of the decimal rounding (printf rounds the exact binary value to nearest, ties to even), negative zero, denormals,
    infinities, NaN; a finite |value| >= 9e14 is left to the host."""
    ctx = gpu_contexts("53birds")
    rng = np.random.default_rng(12)
    vals = [0.0, -0.0, 0.5, -0.5, 0.0005, -0.0005, 0.0015, 0.0025, 0.00005, 0.00015, 0.00025, 0.125, 0.375, 2.5e-4, 1.0, -1.0, 9.9995, 9.99951,
            999.9995, 0.9995, 0.99949997, 0.99950004, 1e-10, -1e-10, 1e-45, 16777216.0, 123456.789, -123456.789, 8.9e14, -8.9e14,
            float("inf"), float("-inf"), float("nan"), 3.4e38, 9.1e14, 1e15]
    # exact ties: k / 2^j with few bits
    vals += [k / 16.0 for k in range(-40, 41)] + [k / 2048.0 for k in range(0, 4096, 7)] + [(2 * k + 1) / 16384.0 for k in range(0, 4000, 3)]
    v = np.concatenate([np.array(vals, dtype=np.float32),
                        rng.uniform(-300, 300, 200_000).astype(np.float32),                       # decibans per codon
                        rng.uniform(0, 1, 100_000).astype(np.float32),                            # branch-length scores
                        (rng.integers(-3_000_000, 3_000_000, 100_000) / 10_000.0 + 0.00005).astype(np.float32),   # near the 4-decimal ties
                        (rng.integers(-300_000, 300_000, 100_000) / 1_000.0 + 0.0005).astype(np.float32),         # near the 3-decimal ties
                        rng.integers(0, 2**32, 300_000, dtype=np.uint64).astype(np.uint32).view(np.float32)])     # any bit pattern
    got = ctx.debug_wig_format(v, decimals)
    n_big = 0
    for x, g in zip(v, got):
        if np.isfinite(x) and abs(float(x)) >= 9e14:
            assert g is None
            n_big += 1
            continue
        if np.isnan(x):
            want = "-nan" if struct.unpack("<I", struct.pack("<f", x))[0] >> 31 else "nan"      # glibc prints the sign bit
        else:
            want = _trimmed(x, decimals)
        assert g == want.encode(), (float(x), g, want)
    assert n_big > 1000


def test_text_buffer_estimate_too_small_writes_again(golden_dir):
    """The text buffer is sized before the sizes are known; with an estimate of one byte per line the first write pass leaves lines
    out and the call must notice and write again.  (The estimate is read once per process and a context's buffer only grows: one
    process per value.)"""
    import subprocess
    import sys
    code = r"""
import hashlib, os, sys
import numpy as np
sys.path.insert(0, sys.argv[1])
import phylocsfpp_amd as pa
m = pa.load_model(os.path.join(sys.argv[2], "models", "53birds"))
rng = np.random.default_rng(3)
nuc = np.frombuffer(b"ACGTacgt", dtype=np.uint8)[rng.integers(0, 8, size=(m.nl, 3000))]
ctx = m.context(device=0, with_bls=True)
text, frames = ctx.score_alignments_6frames_wig(nuc, [(0, 1000, 17, 10**6), (1000, 2000, 5000, 10**6)], ["chrA", "chrB"], 0.1)
print("result", hashlib.sha256(b"|".join(text)).hexdigest(), frames, sum(map(len, text)))
"""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    seen = []
    for est in ("1", "12"):
        p = subprocess.run([sys.executable, "-c", code, root, golden_dir], capture_output=True, text=True, env=dict(os.environ, PCSF_WIG_TEXT_ESTIMATE=est))
        assert p.returncode == 0, p.stderr[-2000:]
        seen.append([l for l in p.stdout.splitlines() if l.startswith("result")][0])
    assert seen[0] == seen[1] and int(seen[0].split()[3]) > 6 * 3000, seen


def test_build_tracks_device_text_equals_host_text_on_a_synthetic_file(tmp_path):
    """The driver end to end, PCSF_DEVICE_WIG=1 (default) against =0 (the host formats): many batches and windows, alignments cut by
    batch boundaries, 42 species the model does not know, both sides of the 1 Mb coordinate break."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cli = os.path.join(root, "phylocsfpp_amd", "host", "phylocsf-amd")
    subprocess.check_call(["make", "-s", "-C", os.path.dirname(cli)])
    maf = tmp_path / "syn.maf"
    subprocess.check_call([sys.executable, os.path.join(root, "tools", "gen_synthetic_maf.py"), str(maf), "--bases", "60000", "--start", "980000", "--seed", "8"],
                          stdout=subprocess.DEVNULL)
    outs = {}
    for mode in ("1", "0"):
        out = tmp_path / ("out" + mode)
        env = dict(os.environ, PCSF_DEVICE_WIG=mode)
        log = subprocess.run([cli, "build-tracks", "--output", str(out), "--batch-sites", "20000", "--window-bytes", "400000",
                              os.path.join(root, "tests", "golden", "models", "58mammals"), str(maf)], env=env, check=True, stderr=subprocess.PIPE).stderr
        outs[mode] = out
    names = sorted(os.listdir(outs["0"]))
    assert len(names) == 7
    for n in names:
        a, b = open(outs["1"] / n, "rb").read(), open(outs["0"] / n, "rb").read()
        assert len(b) > 1000 and a == b, n
