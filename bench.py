#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X PhyloCSF scoring core.

Metric (BASELINE.json): alignment codon-frames scored per second by the build-tracks hot path
(run_tracks, reference src/run.hpp:35-55) with the 58mammals model.  One "step" = one pass of
the HIP path over one batch of synthetic alignment columns already resident in HBM:
BASELINE.json configs[2] -- a 1 Mb, 58-row alignment seen in all 6 frames = 2,000,000
codon-frames per GPU.  Multi-GPU: columns shard embarrassingly (weak scaling: every rank owns
its own 2 M columns); the only exchange is the gather of the per-codon scores to rank 0 (the
wig writer), done with RCCL inside the timed step.

    python bench.py --gpus 1 --steps 20 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
           --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  Extra objects:
  "roofline"      dominant kernel vs the fp64-MFMA peak; duration measured with HIP events on the launch stream
  "strong"        the same path in STRONG scaling: BASELINE.json configs[3]'s total (100 Mb = 200 M codon-frames) cut into
                  contiguous 64-aligned ranges by phylocsfpp_amd.dist.shard_bounds, scores gathered to rank 0
                  (--scaling strong makes this the headline line instead)
  "gather_check"  (N > 1; also inside "strong") the gather of the last step PROVEN outside the timed region: ranks_seen, equal (each rank's
                  hash of its own scores == rank 0's hash of what RCCL delivered from it), recomputed_equal (rank 0 scored every rank's
                  columns again: bit-identical) -- phylocsfpp_amd.dist.ScoreGatherPipeline.check
  "end_to_end"    (N = 1) `phylocsf-amd build-tracks 58mammals <synthetic 1 Mb 100-species MAF>` -- text in, seven wig
                  files out -- run in child processes (reference: src/phylocsf++build_tracks.hpp:55-260, README.rst:116)
  "blocks"        (N = 1) the other half of BASELINE.json configs[2]: `phylocsf-amd score-msa --strategy mle|omega` on the same
                  synthetic MAF in child processes (reference: src/run.hpp:59-182 per block, src/phylocsf++score_msa.hpp:94-127):
                  blocks/s, codon columns/s, rounds, evaluations, which pruning kernel and engine; "roofline" = the products the pruning
                  kernels executed (counted on the device) x 131,072 FLOP over the scoring seconds, against the fp64-MFMA peak;
                  "cpu" = oracle_cli on a prefix of the file
  "other_models"  (N = 1) prune_fixed on 53birds (BASELINE.json configs[1]) and 100vertebrates (configs[4]): three steps each, roofline
  "cpu_baseline"  (N = 1) the CPU oracle, a reference-structured port, on this host's cores: kernel level on a sample of
                  the same batch ("value"), the same CLI driver on the same MAF ("end_to_end_value"), one thread
                  ("single_thread_value")
"""
import argparse
import contextlib
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

MODEL = "58mammals"
MODEL_PREFIX = MODEL   # a shipped model, by name (phylocsfpp_amd/models; reference: src/models.hpp:1443-1455)
SITES_PER_GPU = 2_000_000           # 1 Mb reference x 2 strands x 3 frames / 3 bases per codon
STRONG_TOTAL_SITES = 200_000_000    # BASELINE.json configs[3]: 100 Mb
POOL_SITES = 1 << 17                # unique synthetic columns; the batch samples them with replacement
ALIGNMENT_FRAMES = 600              # ~100 concatenated alignments x 6 frames in a 1 Mb MAF
PEAK_FP64_MFMA_TFLOPS = 78.6        # MI355X fp64 matrix peak (vendor); 78.0 measured by tools/mfma_f64_bench
METRIC = "alignment codons/sec (build-tracks, 58mammals) at 1/2/4/8 MI355X"
E2E_BASES = 1_000_000               # BASELINE.json configs[2]
E2E_SEED = 20261002
E2E_TILES = 10                      # steady-state leg: the segment tiled ten times (10 Mb, ~0.9 GB of text)
CLI_TIMEOUT_S = 900                # a build-tracks child of the bench that takes longer than this is ended (with its process group)
RANKS_TIMEOUT_S = 240              # ... and the --ranks legs (seconds of work): a communicator that never forms must not cost the bench its time limit
CLI = os.path.join(ROOT, "phylocsfpp_amd", "host", "phylocsf-amd")
ORACLE_CLI = os.path.join(ROOT, "oracle", "_build", "oracle_cli")


def flop_per_site(nl, cherries=0, triples=0, quads=0):
    """both models; internal edges as 64x64 mat-vec, element-wise products, root dot.  cherries: cherries whose message
    comes from a pair table (kernels.hip PAIRS) -- one product and one element-wise product less each: the work of the algorithm the
    kernel runs (every MFMA it executes is in this count: the PMC's SQ_VALU_MFMA_BUSY_CYCLES / 64 x 2,048 agrees)."""
    absorbed = cherries + triples + quads   # table nodes: every one removes the product of its own edge (a cherry inside a triple has a table too)
    return 2 * ((nl - 2 - absorbed) * 2 * 64 * 64 + (nl - 1 - absorbed) * 64 + 2 * 64)


def roofline_of(ctx_info, nl, n_sites, kernel_ms):
    """the roofline object of a prune_fixed launch: achieved on the work the kernel's algorithm does, and -- when cherry tables removed
    products -- what the same time means in SURVEY 8(d)'s count of the table-free algorithm"""
    cherries, triples, quads = ctx_info["cherries"], ctx_info.get("triples", 0), ctx_info.get("quads", 0)
    fl, fl_full = flop_per_site(nl, cherries, triples, quads), flop_per_site(nl)
    tf = n_sites * fl / (kernel_ms * 1e-3) / 1e12
    r = {"bound": "mfma", "achieved": tf, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": tf / PEAK_FP64_MFMA_TFLOPS,
         "kernel": "pcsf::prune_fixed", "kernel_ms": kernel_ms, "flop_per_codon_frame": fl}
    if cherries:
        tf_full = n_sites * fl_full / (kernel_ms * 1e-3) / 1e12
        assert ctx_info["products_run"] == nl - 2 - cherries - triples - quads, "table nodes and products do not add up"
        r["cherry_tables"] = {"cherries": cherries, "triples": triples, "quads": quads, "products_per_model": ctx_info["products_run"], "products_per_model_without_tables": ctx_info["products_full"],
                              "table_rows_per_model": ctx_info["rows_per_table_set"], "table_bytes": ctx_info["rows_per_table_set"] * 512 * 2,
                              "survey_8d_flop_per_codon_frame": fl_full, "survey_8d_equivalent_tflops": tf_full,
                              "survey_8d_equivalent_frac": tf_full / PEAK_FP64_MFMA_TFLOPS,
                              "note": "a cherry's message to its parent is looked up per codon pair (65 x 65 rows per cherry and model; one level up, a node over a "
                                      "cherry and a leaf: 65^3 rows), made once per "
                                      "context by the product the kernel would run: same bits: `achieved` / `frac` count the products the kernel runs; the "
                                      "survey_8d_* figures price the same time with SURVEY 8(d)'s table-free count and can exceed the peak"}
    return r


def make_pool(m, seed):
    from phylocsfpp_amd import synth
    return synth.synthetic_codon_columns(m.child1, m.child2, m.branch_len, m.S_c, m.pi_c, m.S_nc, m.pi_nc,
                                         POOL_SITES, seed=seed)


def sample_columns(pool_dev, lo, hi, seed):
    """Columns [lo, hi) of the endless synthetic site stream of `seed`: column i is pool column hash(i, seed), so any
    rank can materialise any range of the SAME stream on its own GPU (one shared generator seed, per-rank slice)."""
    i = torch.arange(lo, hi, dtype=torch.int64, device=pool_dev.device)
    mult, add = 0x1E3779B97F4A7C15, (seed * 0x632BE59BD9B4E019) & 0x3FFFFFFFFFFFFFFF
    idx = ((i * mult + add) >> 23) & (POOL_SITES - 1)   # int64 wrap-around is intended
    del i
    out = torch.empty((pool_dev.shape[0], hi - lo), dtype=torch.uint8, device=pool_dev.device)
    step = 1 << 24
    for a in range(0, hi - lo, step):   # bounded temporaries for 200 M columns
        out[:, a:a + step] = pool_dev[:, idx[a:a + step]]
    return out


def effective_cpus():
    """CPUs this process may actually use: affinity mask capped by the cgroup quota (the GPU box
    shows 256 hardware threads but its container is limited by cpu.max)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(codons, budget_sites):
    """Times the oracle's reference-structured run_tracks loop (model re-instantiated per
    alignment x frame, scalar pruning per codon, OpenMP schedule(dynamic,1) over alignments)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob  # cpu_baseline leg: the oracle is the thing timed here
    m = ob.OracleModel(os.path.join(ROOT, "phylocsfpp_amd", "models", MODEL))   # the oracle reads the files with its own reader
    n = min(budget_sites, codons.shape[1])
    n_units = max(1, int(round(ALIGNMENT_FRAMES * n / SITES_PER_GPU)))
    off = np.linspace(0, n, n_units + 1).astype(np.uint64)
    sample = np.ascontiguousarray(codons[:, :n])
    threads = effective_cpus()
    t0 = time.perf_counter()
    m.baseline_run_tracks(sample, off, threads=threads)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "codon-frames/s", "cores": threads, "kind": "port", "cpu_model": cpu_model_name(),
            "sample": "%d codon-frames in %d alignment-frames of the same batch, %.1f s; reference-structured port "
                      "(model re-instantiated per alignment-frame, scalar pruning), OpenMP schedule(dynamic,1) over "
                      "alignment-frames on %d threads (cgroup CPU quota of this host)" % (n, n_units, dt, threads)}


STATS_RE = re.compile(r"(\d+) alignments, (\d+) reference bases, (\d+) codon-frames, (\d+) backend calls")
SECONDS_RE = re.compile(r"seconds: total ([\d.]+); .*MAF read ([\d.]+), BLS ([\d.]+), batch packing ([\d.]+) \| scoring backend ([\d.]+) \| "
                        r"wig output ([\d.]+) \| smoothing ([\d.]+)")


RANK_RE = re.compile(r"\[rank (\d+) of (\d+), device (\d+)\] gather ([\d.]+) s(?: \(communicator set-up ([\d.]+) s\))?")
STARTUP_RE = re.compile(r"start-up seconds: options \+ model files ([\d.]+), first scoring context ([\d.]+)")


def run_cli(binary, maf, out_dir, extra_args=(), env=None, timeout_s=None):
    """One build-tracks run in a child process; returns the driver's own statistics and the process wall time."""
    os.makedirs(out_dir, exist_ok=True)
    cmd = [binary, "build-tracks", "--output", out_dir] + list(extra_args) + [MODEL_PREFIX, maf]
    t0 = time.perf_counter()
    # in a session of its own: a run that hangs (a --ranks run whose ranks wait for each other) is ended as a whole -- the launcher and its ranks
    # are one process group -- instead of holding up the bench
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, start_new_session=True)
    timeout_s = timeout_s or CLI_TIMEOUT_S
    try:
        out_text, err_text = proc.communicate(timeout=timeout_s)
    except subprocess.TimeoutExpired:
        import signal
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except OSError:
            pass
        out_text, err_text = proc.communicate()
        raise RuntimeError("%s did not finish within %d s: %s" % (" ".join(cmd), timeout_s, err_text[-1500:]))
    wall = time.perf_counter() - t0

    class _Done:
        returncode, stdout, stderr = proc.returncode, out_text, err_text
    p = _Done
    if p.returncode != 0:
        raise RuntimeError("%s failed (%d): %s" % (" ".join(cmd), p.returncode, p.stderr[-2000:]))
    a, s = STATS_RE.search(p.stderr), SECONDS_RE.search(p.stderr)
    if not a or not s:
        raise RuntimeError("cannot parse the statistics of %s: %s" % (binary, p.stderr[-2000:]))
    ranks = RANK_RE.findall(p.stderr)
    if ranks:   # --ranks N: one statistics line per process; the run lasts as long as its slowest rank, the ranks' work adds up
        al, se = STATS_RE.findall(p.stderr), SECONDS_RE.findall(p.stderr)
        if not (len(ranks) == len(al) == len(se)):
            raise RuntimeError("cannot parse the ranks' statistics of %s: %s" % (binary, p.stderr[-2000:]))
        slowest = max(range(len(se)), key=lambda i: float(se[i][0]))
        busy = ["maf_read", "bls", "batch_packing", "scoring_backend", "wig_output", "smoothing"]
        return {"alignments": sum(int(x[0]) for x in al), "ref_bases": sum(int(x[1]) for x in al), "codon_frames": sum(int(x[2]) for x in al),
                "backend_calls": sum(int(x[3]) for x in al), "seconds": float(se[slowest][0]), "process_wall_seconds": wall,
                "startup_seconds": None, "ranks": len(ranks), "devices": sorted(set(int(r[2]) for r in ranks)),
                "seconds_per_rank": [float(x[0]) for x in se], "gather_seconds_per_rank": [float(r[3]) for r in ranks],
                "communicator_setup_seconds_per_rank": [float(r[4]) if r[4] else None for r in ranks],
                "codon_frames_per_rank": [int(x[2]) for x in al],
                "stage_busy": {k: sum(float(x[1 + i]) for x in se) for i, k in enumerate(busy)}}
    return {"alignments": int(a.group(1)), "ref_bases": int(a.group(2)), "codon_frames": int(a.group(3)), "backend_calls": int(a.group(4)),
            "seconds": float(s.group(1)), "process_wall_seconds": wall,
            "startup_seconds": (lambda u: float(u.group(1)) + float(u.group(2)) if u else None)(STARTUP_RE.search(p.stderr)),
            "stage_busy": {"maf_read": float(s.group(2)), "bls": float(s.group(3)), "batch_packing": float(s.group(4)),
                           "scoring_backend": float(s.group(5)), "wig_output": float(s.group(6)), "smoothing": float(s.group(7))}}


def maf_prefix(src, dst, fraction):
    """The first `fraction` of a MAF, cut behind a complete alignment block."""
    size = os.path.getsize(src)
    want = max(1 << 16, int(size * fraction))
    with open(src, "rb") as f:
        data = f.read(min(size, want))
    cut = data.rfind(b"\n\n")
    with open(dst, "wb") as f:
        f.write(data[:cut + 2] if cut > 0 else data)
    return os.path.getsize(dst)


WIGS = ["PhyloCSFpower.wig"] + ["PhyloCSFRaw%s%d.wig" % (s, f) for s in "+-" for f in (1, 2, 3)]


def differing_lines(dir_a, dir_b):
    """Number of differing text lines over the seven wig files of two runs (None if a file is missing)."""
    total = 0
    for w in WIGS:
        pa_, pb_ = os.path.join(dir_a, w), os.path.join(dir_b, w)
        if not (os.path.exists(pa_) and os.path.exists(pb_)):
            return None
        a, b = open(pa_, "rb").read(), open(pb_, "rb").read()
        if a != b:
            la, lb = a.split(b"\n"), b.split(b"\n")
            total += abs(len(la) - len(lb)) + sum(1 for x, y in zip(la, lb) if x != y)
    return total


def end_to_end(device, runs, cpu_rate_all_cores, work_dir):
    """north_star's workload: build-tracks 58mammals on a synthetic 100-species MAF, text to wig files, GPU CLI and
    the oracle-backed CPU driver (same host code, CPU scoring) on the same file."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_synthetic_maf as gen
    maf = os.path.join(work_dir, "synthetic_%dbp_seed%d.maf" % (E2E_BASES, E2E_SEED))
    t0 = time.perf_counter()
    info = gen.generate(maf, bases=E2E_BASES, seed=E2E_SEED)
    gen_s = time.perf_counter() - t0
    cpus = effective_cpus()
    gpu_runs = [run_cli(CLI, maf, os.path.join(work_dir, "gpu"), ["--device", str(device)]) for _ in range(runs)]
    best = min(gpu_runs, key=lambda r: r["seconds"])
    e2e = {"command": "phylocsf-amd build-tracks --device %d %s <maf>" % (device, MODEL_PREFIX),
           "maf": "tools/gen_synthetic_maf.py --bases %d --seed %d: %d species rows (58 mapped by the model), %d blocks, %d bytes; generated in %.1f s"
                  % (E2E_BASES, E2E_SEED, info["rows"], info["blocks"], info["bytes"], gen_s),
           "maf_bytes": info["bytes"], "codon_frames": best["codon_frames"], "ref_bases": best["ref_bases"], "alignments": best["alignments"],
           "backend_calls": best["backend_calls"], "runs": runs, "host_threads": cpus,
           "seconds": best["seconds"], "seconds_all_runs": [r["seconds"] for r in gpu_runs],
           "process_wall_seconds": min(r["process_wall_seconds"] for r in gpu_runs),
           "process_wall_seconds_all_runs": [r["process_wall_seconds"] for r in gpu_runs],
           "process_wall_seconds_median": float(np.median([r["process_wall_seconds"] for r in gpu_runs])),
           "startup_seconds_all_runs": [r["startup_seconds"] for r in gpu_runs],
           "codon_frames_per_s": best["codon_frames"] / best["seconds"],
           "codon_frames_per_s_process_wall": best["codon_frames"] / min(r["process_wall_seconds"] for r in gpu_runs),
           "maf_gigabytes_per_s": info["bytes"] / best["seconds"] / 1e9,
           "stage_busy": best["stage_busy"],
           "note": "seconds = build_tracks() inside the child process (file mapped, parsed, scored, seven wig files written); "
                   "process_wall_seconds adds process start, HIP initialisation, model and backend set-up (startup_seconds: model "
                   "files + the first scoring context, i.e. HIP runtime, code object, branch matrices) and the tear-down of the HIP "
                   "runtime at exit; a HIP process that only creates a context and exits takes 0.28 s on this box "
                   "(profiles/r02/startup_profile.txt)"}
    # CPU driver on the same file, or on a prefix sized for ~20 s when the whole file would take longer than ~30 s
    env = dict(os.environ, PCSF_THREADS=str(cpus), OMP_NUM_THREADS=str(cpus))
    est = best["codon_frames"] / max(cpu_rate_all_cores, 1.0)
    frac = 1.0 if est <= 30.0 else 20.0 / est
    cpu_maf = maf
    if frac < 1.0:
        cpu_maf = os.path.join(work_dir, "cpu_prefix.maf")
        maf_prefix(maf, cpu_maf, frac)
    cpu_run = run_cli(ORACLE_CLI, cpu_maf, os.path.join(work_dir, "cpu"), env=env)
    cpu = {"end_to_end_value": cpu_run["codon_frames"] / cpu_run["seconds"], "end_to_end_seconds": cpu_run["seconds"],
           "end_to_end_sample": ("the whole file" if frac >= 1.0 else "the first %.1f %% of the file" % (100 * frac)) +
                                ": %d codon-frames, oracle/_build/oracle_cli build-tracks (the product's host driver on the CPU oracle), %d threads"
                                % (cpu_run["codon_frames"], cpus)}
    if frac >= 1.0:
        e2e["wig_lines_differing_from_cpu_driver"] = differing_lines(os.path.join(work_dir, "gpu"), os.path.join(work_dir, "cpu"))
    # one thread: a prefix sized for ~10 s
    est1 = best["codon_frames"] / max(cpu_rate_all_cores / cpus, 1.0)
    one_maf = os.path.join(work_dir, "cpu_one_thread.maf")
    maf_prefix(maf, one_maf, min(1.0, 10.0 / est1))
    env1 = dict(os.environ, PCSF_THREADS="1", OMP_NUM_THREADS="1")
    one = run_cli(ORACLE_CLI, one_maf, os.path.join(work_dir, "cpu1"), ["--threads", "1"], env=env1)
    cpu["single_thread_value"] = one["codon_frames"] / one["seconds"]
    cpu["single_thread_sample"] = "%d codon-frames of the same file in %.1f s, same driver, 1 thread" % (one["codon_frames"], one["seconds"])
    e2e["vs_cpu_driver"] = e2e["codon_frames_per_s"] / cpu["end_to_end_value"]
    # steady state of the same pipeline: the 1 Mb segment tiled ten times along the chromosome (one-time costs -- first
    # allocations, page-locking, pipeline fill -- are a third of a 1 Mb run)
    try:
        import replicate_maf
        big = os.path.join(work_dir, "synthetic_x%d.maf" % E2E_TILES)
        with contextlib.redirect_stdout(sys.stderr):   # the tool reports what it wrote; stdout carries the one JSON line
            replicate_maf.tile(maf, big, E2E_TILES, 1_100_000)
        big_runs = [run_cli(CLI, big, os.path.join(work_dir, "gpu_big"), ["--device", str(device)]) for _ in range(2)]
        b = min(big_runs, key=lambda r: r["seconds"])
        e2e["steady_state"] = {"maf": "the same segment tiled %d times along the chromosome (tools/replicate_maf.py), %d bytes" % (E2E_TILES, os.path.getsize(big)),
                               "codon_frames": b["codon_frames"], "seconds": b["seconds"], "seconds_all_runs": [r["seconds"] for r in big_runs],
                               "codon_frames_per_s": b["codon_frames"] / b["seconds"], "backend_calls": b["backend_calls"],
                               "maf_gigabytes_per_s": os.path.getsize(big) / b["seconds"] / 1e9, "stage_busy": b["stage_busy"]}
        # north_star's "single RCCL gather ... for the output wig stream" in the command a user runs, rehearsed on this one GPU: one process
        # per GPU (--ranks 1), every batch's text kept on the device, one grouped RCCL send / receive (the root's text to itself), the files
        # written from what came back -- against the plain run's files
        try:
            rk = min([run_cli(CLI, big, os.path.join(work_dir, "gpu_big_ranks"), ["--devices", str(device), "--ranks", "1", "--gather-root-text", "1"],
                                  timeout_s=RANKS_TIMEOUT_S) for _ in range(2)],
                     key=lambda r: r["process_wall_seconds"])
            e2e["steady_state"]["ranks_rehearsal"] = {
                "command": "phylocsf-amd build-tracks --ranks 1 --gather-root-text 1 58mammals <maf>", "ranks": rk["ranks"], "seconds": rk["seconds"],
                "gather_seconds": rk["gather_seconds_per_rank"][0], "communicator_setup_seconds": rk["communicator_setup_seconds_per_rank"][0],
                "process_wall_seconds": rk["process_wall_seconds"],
                "plain_run_process_wall_seconds": min(r["process_wall_seconds"] for r in big_runs),
                "wig_lines_differing_from_the_plain_run": differing_lines(os.path.join(work_dir, "gpu_big_ranks"), os.path.join(work_dir, "gpu_big"))}
        except Exception as ex:
            e2e["steady_state"]["ranks_rehearsal"] = {"error": str(ex)[-500:]}
        # the smoothing pass on the same file (--output-phylo 1 --output-regions 1, BASELINE.json configs[4]'s addition): the recurrences
        # on the device (default) and on the host (PCSF_HOST_HMM=1: host/hmm.hpp, six threads), the twelve smoothed files compared
        try:
            prm = json.load(open(os.path.join(ROOT, "tests", "golden", "build-tracks", "hmm_params_backsolved.json")))["hmm_params"]
            sm_args = ["--device", str(device), "--output-phylo", "1", "--output-regions", "1", "--hmm-params", ",".join("%.17g" % x for x in prm)]
            dev = min([run_cli(CLI, big, os.path.join(work_dir, "gpu_sm_dev"), sm_args) for _ in range(2)], key=lambda r: r["seconds"])
            host = run_cli(CLI, big, os.path.join(work_dir, "gpu_sm_host"), sm_args, env=dict(os.environ, PCSF_HOST_HMM="1"))
            names = sorted(n for n in os.listdir(os.path.join(work_dir, "gpu_sm_host")) if not n.startswith("PhyloCSFRaw") and n != "PhyloCSFpower.wig")
            same = all(open(os.path.join(work_dir, "gpu_sm_dev", n), "rb").read() == open(os.path.join(work_dir, "gpu_sm_host", n), "rb").read() for n in names)
            e2e["steady_state"]["with_smoothing"] = {
                "command": "phylocsf-amd build-tracks --output-phylo 1 --output-regions 1 --hmm-params <the reference's test parameters> 58mammals <maf>",
                "seconds": dev["seconds"], "smoothing_seconds": dev["stage_busy"]["smoothing"],
                "host_pass_seconds": host["seconds"], "host_pass_smoothing_seconds": host["stage_busy"]["smoothing"],
                "smoothed_files": len(names), "smoothed_files_identical_to_host_pass": bool(same)}
        except Exception as ex:
            e2e["steady_state"]["with_smoothing"] = {"error": str(ex)[-500:]}
        os.remove(big)
    except Exception as ex:
        e2e["steady_state"] = {"error": str(ex)[-500:]}
    return e2e, cpu


def host_ceiling(run, n_gpus):
    """What the host side alone allows a build-tracks run (`run`: run_cli's statistics): stage 1 (parse + branch-length scores + packing,
    one window ahead of the scoring, all host threads) and stage 3 (the writers) each take this long whatever the number of GPUs; the
    scoring backends share theirs."""
    sb = run["stage_busy"]
    stage1 = sb["maf_read"] + sb["bls"] + sb["batch_packing"]
    return {"stage1_parse_and_pack_seconds": stage1, "stage3_writers_seconds": sb["wig_output"],
            "scoring_backends_seconds_per_device": sb["scoring_backend"] / max(n_gpus, 1),
            "codon_frames_per_s": run["codon_frames"] / max(stage1, sb["wig_output"], 1e-9), "host_threads": effective_cpus(),
            "reading": "the run cannot be faster than its slowest stage: min(this ceiling, n_gpus x the one-device rate) bounds all_devices"}


def end_to_end_multi(n_gpus, work_dir):
    """BASELINE.json configs[3] in the form the C++ driver runs it: ONE process, `phylocsf-amd build-tracks --devices 0,..,N-1` on the
    synthetic segment tiled along the chromosome (10 Mb per GPU, at most 40 Mb), text in, seven wig files out; the same file on one
    device beside it.  Run by rank 0 while the other ranks wait at a barrier outside every timed region."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import gen_synthetic_maf as gen
    import replicate_maf
    maf = os.path.join(work_dir, "synthetic_%dbp_seed%d.maf" % (E2E_BASES, E2E_SEED))
    if not os.path.exists(maf):
        gen.generate(maf, bases=E2E_BASES, seed=E2E_SEED)
    tiles = min(E2E_TILES * n_gpus, 40)
    big = os.path.join(work_dir, "synthetic_x%d.maf" % tiles)
    with contextlib.redirect_stdout(sys.stderr):
        replicate_maf.tile(maf, big, tiles, 1_100_000)
    out = {"maf": "the 1 Mb synthetic segment tiled %d times (tools/replicate_maf.py), %d bytes" % (tiles, os.path.getsize(big)), "n_gpus": n_gpus}
    all_devs = ",".join(str(d) for d in range(n_gpus))
    for label, devs, extra in (("all_devices", all_devs, []), ("one_device", "0", []),
                               # one PROCESS per GPU, the ranks' wig text gathered onto rank 0's GPU over RCCL (csrc/comm.cpp): north_star's form
                               ("ranks_rccl", all_devs, ["--ranks", str(n_gpus)])):
        try:
            runs = [run_cli(CLI, big, os.path.join(work_dir, "gpu_multi_" + label), ["--devices", devs] + extra,
                            timeout_s=RANKS_TIMEOUT_S if label == "ranks_rccl" else None) for _ in range(2)]
        except Exception as ex:
            if label != "ranks_rccl":
                raise
            out[label] = {"error": str(ex)[-1500:]}
            continue
        b = min(runs, key=lambda r: r["seconds"] if label != "ranks_rccl" else r["process_wall_seconds"])
        if label == "ranks_rccl":
            out[label] = {"command": "phylocsf-amd build-tracks --devices %s --ranks %d %s <maf>" % (devs, n_gpus, MODEL_PREFIX), "codon_frames": b["codon_frames"],
                          "ranks": b["ranks"], "devices": b["devices"], "seconds_slowest_rank": b["seconds"], "seconds_per_rank": b["seconds_per_rank"],
                          "gather_seconds_per_rank": b["gather_seconds_per_rank"], "communicator_setup_seconds_per_rank": b["communicator_setup_seconds_per_rank"],
                          "codon_frames_per_rank": b["codon_frames_per_rank"],
                          "process_wall_seconds": b["process_wall_seconds"], "process_wall_seconds_all_runs": [r["process_wall_seconds"] for r in runs],
                          "codon_frames_per_s_process_wall": b["codon_frames"] / b["process_wall_seconds"],
                          "wig_lines_differing_from_one_device": differing_lines(os.path.join(work_dir, "gpu_multi_ranks_rccl"), os.path.join(work_dir, "gpu_multi_one_device")),
                          "note": "rates by process wall: the launcher's N processes each pay their own HIP start-up (~0.2 s) beside the others'"}
            out[label]["outputs_check"] = {"ranks_seen": b["ranks"], "devices_seen": len(b["devices"]), "equal": out[label]["wig_lines_differing_from_one_device"] == 0}
            continue
        out[label] = {"command": "phylocsf-amd build-tracks --devices %s %s <maf>" % (devs, MODEL_PREFIX), "codon_frames": b["codon_frames"], "seconds": b["seconds"],
                      "seconds_all_runs": [r["seconds"] for r in runs], "process_wall_seconds": min(r["process_wall_seconds"] for r in runs),
                      "codon_frames_per_s": b["codon_frames"] / b["seconds"], "backend_calls": b["backend_calls"], "stage_busy": b["stage_busy"]}
    out["speedup_vs_one_device"] = out["all_devices"]["codon_frames_per_s"] / out["one_device"]["codon_frames_per_s"]
    out["host_ceiling"] = host_ceiling(out["all_devices"], n_gpus)
    try:
        out["wig_lines_differing_between_the_two"] = differing_lines(os.path.join(work_dir, "gpu_multi_all_devices"), os.path.join(work_dir, "gpu_multi_one_device"))
    except Exception as ex:
        out["wig_lines_differing_between_the_two"] = str(ex)[-200:]
    # the multi-device run is only a measurement if it wrote what one device writes
    out["outputs_check"] = {"devices_seen": n_gpus, "equal": out["wig_lines_differing_between_the_two"] == 0}
    os.remove(big)
    return out


MSA_RE = re.compile(r"(\d+) blocks, (\d+) codon columns; seconds: total ([\d.]+) \(MAF read \+ translation \+ branch-length scores ([\d.]+), "
                    r"scoring backend ([\d.]+), output ([\d.]+)\)")
ROUNDS_RE = re.compile(r"run_(?:mle|omega): (\d+) blocks, (\w+) engine, (\w+): (\d+) rounds, (\d+) evaluations \((\d+) with a new eigen-system\): set-up ([\d.]+) s, rounds ([\d.]+) s")
PRODUCTS_RE = re.compile(r"run_(?:mle|omega): pruning products (\d+) =")
KERNEL_RE = re.compile(r"pruning kernel of this list: (\w+)")


def run_score_msa(binary, strategy, maf, out_dir, extra_args=(), env=None):
    """One score-msa run in a child process: the driver's own statistics, the lock-step engine's (PCSF_TIME_BLOCKS) and the wall time."""
    os.makedirs(out_dir, exist_ok=True)
    cmd = [binary, "score-msa", "--strategy", strategy, "--comp-anc", "1" if strategy == "mle" else "0", "--output", out_dir] + list(extra_args) + [MODEL_PREFIX, maf]
    t0 = time.perf_counter()
    p = subprocess.run(cmd, capture_output=True, text=True, env=dict(env or os.environ, PCSF_TIME_BLOCKS="1"))
    wall = time.perf_counter() - t0
    if p.returncode != 0:
        raise RuntimeError("%s failed (%d): %s" % (" ".join(cmd), p.returncode, p.stderr[-2000:]))
    m = MSA_RE.search(p.stderr)
    if not m:
        raise RuntimeError("cannot parse the statistics of %s: %s" % (binary, p.stderr[-2000:]))
    rounds = ROUNDS_RE.findall(p.stderr)
    k = KERNEL_RE.search(p.stderr)
    return {"blocks": int(m.group(1)), "codon_columns": int(m.group(2)), "seconds": float(m.group(3)), "read_seconds": float(m.group(4)),
            "scoring_seconds": float(m.group(5)), "process_wall_seconds": wall,
            "rounds_max": max([int(r[3]) for r in rounds], default=None), "evaluations": sum(int(r[4]) for r in rounds) or None,
            "eigen_solves": sum(int(r[5]) for r in rounds), "engine": rounds[0][1] if rounds else None,
            "products": sum(int(x) for x in PRODUCTS_RE.findall(p.stderr)),
            "groups": len(rounds) or None, "pruning_kernel": k.group(1) if k else None,
            "scores": os.path.join(out_dir, os.path.basename(maf) + ".scores")}


def read_scores(path):
    rows = [l.rstrip("\n").split("\t") for l in open(path) if not l.startswith("#")][1:]
    return np.array([[float(x) for x in r[4:]] for r in rows], dtype=np.float64)


def block_parity(strategy, gpu, cpu):
    """Rows of the GPU .scores against the oracle's, BOTH score columns (PhyloCSF score; ancestral score where it is computed), with the
    CONTROL beside them (tests/golden/oracle/*.npz: the oracle against itself under few-ulp perturbations of its sums, on the first rows of
    this very file; tests/golden/make_omega_control.py): rows beyond the reference's own tolerance (test/tests.sh:41 MLE d^2 <= 0.001,
    :46 OMEGA d^2 <= 0.1) and how many of those the control moves as well (spread >= 0.01 MLE / 0.1 OMEGA) -- a row beyond the tolerance
    that the control does NOT move would be an arithmetic difference.  cpu None: the committed oracle results of the whole file (MLE)."""
    gold = os.path.join(ROOT, "tests", "golden", "oracle")
    tol, moved_at = (0.001 ** 0.5, 0.01) if strategy == "mle" else (0.1 ** 0.5, 0.1)
    spread = base = sized = None
    try:
        if strategy == "mle":
            ctl = np.load(os.path.join(gold, "mle_control_full.npz"))
            base = ctl["base"]
            spread = np.max([np.abs(ctl["p%d" % b] - base) for b in (1, 2, 4, 7)], axis=0)
            sized = [np.abs(ctl["p%d" % b] - base) for b in (8, 16)]   # the oracle against itself, every site's ln z moved by <= 1.1e-13 relative
        else:
            ctl = np.load(os.path.join(gold, "omega_control.npz"))
            base = ctl["config3_omega_base"][:, None]
            spread = np.max([np.abs(ctl["config3_omega_p%d" % b][:, None] - base) for b in (1, 2, 4, 7)], axis=0)
    except Exception:
        pass
    if cpu is None:
        cpu = base
    out = {"rows": int(len(cpu)), "reference_tolerance_decibans": tol}
    names = ["phylocsf", "anc"] if strategy == "mle" else ["phylocsf"]
    for col, name in enumerate(names):
        g, o = gpu[:len(cpu), col], cpu[:, col]
        both = np.isfinite(g) & np.isfinite(o)
        d = np.where(both, np.abs(g - o), 0.0)
        rel = d / np.maximum(1e-6, np.abs(np.where(both, o, 1.0)))
        beyond = np.flatnonzero(d > tol)
        col_out = {"nan_pattern_equal": bool(np.array_equal(np.isnan(g), np.isnan(o))), "finite_rows": int(both.sum()),
                   "rows_within_1e-6_relative": int((rel[both] <= 1e-6).sum()), "max_abs_diff_decibans": float(d.max()) if len(d) else 0.0,
                   "median_abs_diff_decibans": float(np.median(d[both])) if both.any() else 0.0, "rows_beyond_reference_tolerance": [int(i) for i in beyond]}
        if spread is not None and base is not None and col < spread.shape[1]:
            n_ctl = min(len(cpu), spread.shape[0])
            same = bool(np.array_equal(o[:n_ctl], base[:n_ctl, col], equal_nan=True))     # the control was made on these rows by this oracle
            col_out["control_rows"] = n_ctl
            col_out["control_is_of_these_rows"] = same
            col_out["of_which_moved_in_control"] = int(sum(1 for i in beyond if i < n_ctl and spread[i, col] >= moved_at))
            col_out["beyond_tolerance_where_the_few_ulp_control_is_still"] = [int(i) for i in beyond if i < n_ctl and spread[i, col] < moved_at]
            col_out["rows_the_control_itself_moves_beyond_tolerance"] = int((np.nan_to_num(spread[:n_ctl, col]) > tol).sum())
            if sized is not None:   # a perturbation of the size measured between the HIP path and the oracle: which rows it moves is random, how many is not
                col_out["rows_beyond_tolerance_in_the_implementation_sized_control"] = [int((x[:n_ctl, col] > tol).sum()) for x in sized]
                col_out["rows_within_1e-6_in_the_implementation_sized_control"] = [int((x[:n_ctl, col] / np.maximum(1e-6, np.abs(base[:n_ctl, col])) <= 1e-6).sum()) for x in sized]
        out[name] = col_out
    out["note"] = ("of_which_moved_in_control: rows the oracle itself moves under a few-ulp reordering of its sums; the rows where that control is still are "
                   "searches that part from the oracle's in their last Brent steps with objective values equal to 5e-14 -- the GPU suite replays the oracle's "
                   "search on the device for every such row (tests/test_gpu_config3_blocks.py; profiles/r05/mle_divergence.txt); the implementation-sized "
                   "control (ln z of every site moved by <= 1.1e-13, independently per evaluation) moves far more rows than the device does")
    # (kept at the top level for continuity with rounds 3-4: the PhyloCSF score column)
    out["rows_within_1e-6_relative"] = out["phylocsf"]["rows_within_1e-6_relative"]
    out["finite_rows"] = out["phylocsf"]["finite_rows"]
    out["nan_pattern_equal"] = out["phylocsf"]["nan_pattern_equal"]
    out["max_abs_diff_decibans"] = out["phylocsf"]["max_abs_diff_decibans"]
    out["median_abs_diff_decibans"] = out["phylocsf"]["median_abs_diff_decibans"]
    return out


def block_strategies(device, maf, work_dir, runs):
    """configs[2], "fixed + omega MLE": score-msa --strategy mle and omega on the synthetic MAF (GPU CLI), and the oracle driver on a
    prefix of the same file sized for ~20 s per strategy; rows of the prefix compared between the two."""
    cpus = effective_cpus()
    out = {"maf": os.path.basename(maf), "model": MODEL, "host_threads": cpus}
    env = dict(os.environ, PCSF_THREADS=str(cpus), OMP_NUM_THREADS=str(cpus))
    # calibrate the CPU sample on a small prefix (MLE), then size both prefixes
    cal = os.path.join(work_dir, "blocks_cal.maf")
    maf_prefix(maf, cal, 0.004)
    t0 = time.perf_counter()
    c = run_score_msa(ORACLE_CLI, "mle", cal, os.path.join(work_dir, "cpu_cal"), env=env)
    per_block_mle = max(1e-3, (time.perf_counter() - t0) / max(1, c["blocks"]))
    total_blocks = None
    for strategy in ("mle", "omega"):
        gpu_runs = [run_score_msa(CLI, strategy, maf, os.path.join(work_dir, "gpu_" + strategy), ["--device", str(device)]) for _ in range(runs)]
        best = min(gpu_runs, key=lambda r: r["scoring_seconds"])
        total_blocks = best["blocks"]
        leg = {"command": "phylocsf-amd score-msa --strategy %s --device %d %s <maf>" % (strategy, device, MODEL_PREFIX),
               "blocks": best["blocks"], "codon_columns": best["codon_columns"], "runs": runs,
               "seconds": best["seconds"], "scoring_seconds": best["scoring_seconds"], "read_seconds": best["read_seconds"],
               "scoring_seconds_all_runs": [r["scoring_seconds"] for r in gpu_runs],
               "process_wall_seconds": min(r["process_wall_seconds"] for r in gpu_runs),
               "process_wall_seconds_median": float(np.median([r["process_wall_seconds"] for r in gpu_runs])),
               "blocks_per_s": best["blocks"] / best["scoring_seconds"], "codon_columns_per_s": best["codon_columns"] / best["scoring_seconds"],
               "blocks_per_s_process_wall": best["blocks"] / min(r["process_wall_seconds"] for r in gpu_runs),
               "rounds": best["rounds_max"], "evaluations": best["evaluations"], "eigen_solves": best["eigen_solves"], "block_groups": best["groups"],
               "pruning_kernel": best["pruning_kernel"], "engine": best["engine"]}
        # Roofline of the leg: the matrix work the pruning kernels EXECUTED -- every tile job counts its 64x64 x 64x16 products on the device
        # (FactoredParams::product_count), 64 x v_mfma_f64_16x16x4_f64 = 131,072 FLOP each -- over the scoring seconds of the whole call
        # (engine set-up, every round's eigen-solver / edge-scale / pruning launches, the host's share).  Not counted: the eigen-solver's
        # Jacobi sweeps (vector fp64, ~0.3 MFLOP per solve) and the element-wise work.
        flop = best["products"] * 131072.0
        leg["roofline"] = {"bound": "mfma", "achieved": flop / best["scoring_seconds"] / 1e12, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                           "frac": flop / best["scoring_seconds"] / 1e12 / PEAK_FP64_MFMA_TFLOPS, "executed_products": best["products"],
                           "executed_tflop": flop / 1e12, "seconds": best["scoring_seconds"],
                           "kernels": "pcsf::%s (products), pcsf::eigen_omega_kernel, pcsf::edge_scale_kernel, pcsf::advance_blocks" % best["pruning_kernel"],
                           "note": "whole-call fraction; the kernel split is in profiles/r04/config3_%s_kernel_stats.csv" % strategy}
        # CPU: a prefix sized for ~20 s (an OMEGA block costs the oracle ~5.5 MLE blocks)
        per_block = per_block_mle * (1.0 if strategy == "mle" else 5.5)
        frac = min(1.0, max(0.002, 20.0 / (per_block * best["blocks"])))
        cpu_maf = os.path.join(work_dir, "blocks_cpu_%s.maf" % strategy)
        maf_prefix(maf, cpu_maf, frac)
        cr = run_score_msa(ORACLE_CLI, strategy, cpu_maf, os.path.join(work_dir, "cpu_" + strategy), env=env)
        leg["cpu"] = {"kind": "port", "cores": cpus, "blocks": cr["blocks"], "codon_columns": cr["codon_columns"], "scoring_seconds": cr["scoring_seconds"],
                      "blocks_per_s": cr["blocks"] / cr["scoring_seconds"], "codon_columns_per_s": cr["codon_columns"] / cr["scoring_seconds"],
                      "sample": "the first %d blocks of the same file, oracle/_build/oracle_cli score-msa --strategy %s, %d threads" % (cr["blocks"], strategy, cpus)}
        leg["vs_cpu"] = leg["codon_columns_per_s"] / leg["cpu"]["codon_columns_per_s"]
        # the prefix's rows, GPU against oracle
        try:
            leg["parity_on_cpu_sample"] = block_parity(strategy, read_scores(best["scores"])[:cr["blocks"]], read_scores(cr["scores"]))
        except Exception as ex:
            leg["parity_on_cpu_sample"] = {"error": str(ex)[-300:]}
        if strategy == "mle":   # ... and ALL rows against the committed oracle results of this very file (no CPU time in the bench)
            try:
                leg["parity_on_the_whole_file"] = block_parity(strategy, read_scores(best["scores"]), None)
            except Exception as ex:
                leg["parity_on_the_whole_file"] = {"error": str(ex)[-300:]}
        out[strategy] = leg
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--sites", type=int, default=SITES_PER_GPU, help="codon-frames per GPU per step (weak scaling)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="strong: --total-sites codon-frames are cut into per-rank ranges (BASELINE.json configs[3])")
    ap.add_argument("--total-sites", type=int, default=STRONG_TOTAL_SITES, help="codon-frames of the whole job in strong scaling")
    ap.add_argument("--strong-steps", type=int, default=2, help="steps of the extra strong-scaling leg of a weak run (0 disables)")
    ap.add_argument("--cpu-baseline-sites", type=int, default=-1,
                    help="codon-frames for the CPU baseline sample (default: sized for ~10 s; 0 disables the CPU legs)")
    ap.add_argument("--end-to-end-runs", type=int, default=3, help="build-tracks CLI runs on the synthetic 1 Mb MAF (0 disables; N = 1 only)")
    ap.add_argument("--block-runs", type=int, default=2, help="score-msa --strategy mle / omega runs on the same MAF (0 disables; N = 1 only)")
    ap.add_argument("--other-models", type=int, default=1, help="1: also time prune_fixed on 53birds and 100vertebrates (three steps each; N = 1 only)")
    ap.add_argument("--table-forms", type=int, default=1, help="1: also time the launch under the other forms of tracks mode (full program, cherries only, "
                    "quads when the device has 150 GB free) and compare their bits with the default form's (N = 1 only)")
    ap.add_argument("--force-gather", type=int, default=0,
                    help="1: at N = 1, run the N > 1 code path anyway (a world-size-1 RCCL process group, the pipelined gather and its "
                         "gather_check) -- the one-GPU rehearsal of a SCALE run; changes what `value` times, so never the driver's line")
    ap.add_argument("--keep-files", default=None, help="directory for the synthetic MAF and the wig files (default: a temporary one)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the scoring core has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    gathered = world > 1 or bool(args.force_gather)     # the scores travel through an RCCL gather to rank 0
    if gathered:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29655")
        if rank == 0:
            try:
                os.remove("/tmp/pcsf_bench_done_%s" % os.environ.get("MASTER_PORT", "0"))
            except OSError:
                pass
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import phylocsfpp_amd as pa
    from phylocsfpp_amd import dist as pdist
    m = pa.load_model(MODEL_PREFIX)                 # the package's own reader (C-ABI pcsf_model_open)
    ctx = m.context(device=local_rank)
    pool_host = make_pool(m, seed=E2E_SEED)
    pool = torch.from_numpy(pool_host).to(dev)

    def timed(step, fence, steps, warmup):
        for _ in range(warmup):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        fence()
        elapsed = time.perf_counter() - t0
        if gathered:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return elapsed

    def run_leg(n_local, lo, steps, warmup, shard_lows=None):
        """Scores columns [lo, lo + n_local) of the synthetic stream `steps` times; N > 1: two score buffers per rank, the
        gather of step k (RCCL, its own stream) runs while step k+1 is scored.  Ragged strong-scaling shards are padded to
        the widest shard for the gather (like dist.gather_to_root)."""
        codons = sample_columns(pool, lo, lo + n_local, E2E_SEED)
        width = n_local
        if gathered:
            w = torch.tensor([n_local], dtype=torch.int64, device=dev)
            dist.all_reduce(w, op=dist.ReduceOp.MAX)
            width = int(w.item())
        pipe = pdist.ScoreGatherPipeline(width, torch.float64, dev, depth=2, dst=0) if gathered else None
        scores = torch.empty(width, dtype=torch.float64, device=dev) if pipe is None else None

        last_slot = [None]

        def step():
            if pipe is None:
                ctx.score_sites_fixed_torch(codons, out=scores[:n_local])
            else:
                ctx.score_sites_fixed_torch(codons, out=pipe.acquire()[:n_local])
                last_slot[0] = pipe.submit()

        def fence():
            if pipe is not None:
                pipe.drain()       # every rank's scores of every step have arrived on rank 0
                dist.barrier()
            torch.cuda.synchronize()

        elapsed = timed(step, fence, steps, warmup)
        # Outside the timed region, N > 1: PROVE the gather of the last step.  Every rank hashes its own score buffer, the hashes are
        # all-gathered, rank 0 hashes what RCCL delivered from every rank and compares; and rank 0 scores every rank's columns again
        # itself (the site stream is a counter-based generator: any rank can make any range) and compares bit for bit.
        check = None
        if pipe is not None:
            def recompute(r, n):
                r_lo = r * n_local if shard_lows is None else shard_lows[r]
                again = torch.empty(n, dtype=torch.float64, device=dev)
                ctx.score_sites_fixed_torch(sample_columns(pool, r_lo, r_lo + n, E2E_SEED), out=again)
                torch.cuda.synchronize()
                return again
            check = pipe.check(last_slot[0], n_local, recompute=recompute if rank == 0 else None)
            if check is not None:
                check["backend"] = dist.get_backend()
                check["what"] = ("last step: hash of each rank's own scores == hash of what rank 0 received from it (ranks_seen of %d); "
                                 "recomputed_equal: rank 0 scored every rank's columns again, bit-identical to what arrived" % world)
        return elapsed, codons, check

    bounds = pdist.shard_bounds(args.total_sites, world, 64)
    if args.scaling == "weak":
        n_sites = args.sites
        elapsed, codons, gather_check = run_leg(n_sites, rank * n_sites, args.steps, args.warmup)
        total_per_step = n_sites * world
    else:
        lo, hi = bounds[rank]
        n_sites = hi - lo
        elapsed, codons, gather_check = run_leg(n_sites, lo, args.steps, args.warmup, shard_lows=[b[0] for b in bounds])
        total_per_step = args.total_sites

    # dominant-kernel duration: HIP events recorded on the launch stream around each launch
    scores = torch.empty(n_sites, dtype=torch.float64, device=dev)
    ctx.set_timing(True)
    durs = []
    for _ in range(max(3, min(args.steps, 10))):
        ctx.score_sites_fixed_torch(codons, out=scores)
        durs.append(ctx.last_kernel_ms())
    ctx.set_timing(False)
    torch.cuda.synchronize()
    kernel_ms = float(np.mean(durs))
    tracks_info = ctx.tracks_program_info()
    codons_host = codons[:, :min(n_sites, 1 << 20)].cpu().numpy() if rank == 0 else None

    # The same launch under the other forms of tracks mode: the full program (rounds 1-5's kernel), cherries only, and -- opt-in
    # in the library (PCSF_TABLE_LEAVES=4; 65^4 rows per node: 94 GB for 58mammals, seconds of context creation) -- the nodes over four leaves.
    # Every form must give the default form's bits.  `value` above is the DEFAULT form's.
    table_forms = None
    if rank == 0 and world == 1 and args.table_forms:
        table_forms = {}
        saved = {k: os.environ.get(k) for k in ("PCSF_CHERRY_TABLES", "PCSF_TABLE_LEAVES", "PCSF_TABLE_BYTES")}
        forms = [("full_program", {"PCSF_CHERRY_TABLES": "0"}), ("cherries_only", {"PCSF_TABLE_LEAVES": "2"})]
        if torch.cuda.mem_get_info()[0] > (150 << 30):
            forms.append(("quads_opt_in", {"PCSF_TABLE_LEAVES": "4", "PCSF_TABLE_BYTES": str(140 << 30)}))
        for form, env in forms:
            try:
                for k in saved:
                    os.environ.pop(k, None)
                os.environ.update(env)
                t_c = time.perf_counter()
                cf = m.context(device=local_rank)
                t_c = time.perf_counter() - t_c
                out_f = torch.empty(n_sites, dtype=torch.float64, device=dev)
                cf.score_sites_fixed_torch(codons, out=out_f)
                cf.set_timing(True)
                df = []
                for _ in range(5):
                    cf.score_sites_fixed_torch(codons, out=out_f)
                    df.append(cf.last_kernel_ms())
                cf.set_timing(False)
                torch.cuda.synchronize()
                info_f = cf.tracks_program_info()
                rf = roofline_of(info_f, m.nl, n_sites, float(np.mean(df)))
                table_forms[form] = {"environment": env, "codon_frames_per_s": n_sites / (float(np.mean(df)) * 1e-3), "kernel_ms": float(np.mean(df)),
                                     "products_per_model": info_f["products_run"], "table_bytes": info_f["rows_per_table_set"] * 512 * 2 if info_f["cherries"] else 0,
                                     "context_creation_seconds": t_c, "frac": rf["frac"],
                                     "bit_identical_to_the_default_form": bool(torch.equal(out_f.view(torch.int64), scores.view(torch.int64)))}
                cf.close()
                del out_f
            except Exception as ex:
                table_forms[form] = {"error": str(ex)[-500:]}
        for k, v in saved.items():
            os.environ.pop(k, None)
            if v is not None:
                os.environ[k] = v
        torch.cuda.empty_cache()
    del codons, scores

    # the same kernel on the other two models the reference ships and BASELINE.json's configs name (53birds: configs[1]; 100vertebrates:
    # configs[4]): three steps each on synthetic columns of that model's tree, HIP events on the launch stream
    other_models = None
    if rank == 0 and world == 1 and args.other_models:
        other_models = {}
        for name in ("53birds", "100vertebrates"):
            try:
                m2 = pa.load_model(name)
                c2 = m2.context(device=local_rank)
                pool2 = torch.from_numpy(make_pool(m2, seed=E2E_SEED)).to(dev)
                cod2 = sample_columns(pool2, 0, args.sites, E2E_SEED)
                sc2 = torch.empty(args.sites, dtype=torch.float64, device=dev)
                c2.score_sites_fixed_torch(cod2, out=sc2)      # warm-up
                c2.set_timing(True)
                d2 = []
                for _ in range(3):
                    c2.score_sites_fixed_torch(cod2, out=sc2)
                    d2.append(c2.last_kernel_ms())
                c2.set_timing(False)
                torch.cuda.synchronize()
                ms = float(np.mean(d2))
                r2 = roofline_of(c2.tracks_program_info(), m2.nl, args.sites, ms)
                other_models[name] = {"n_leaves": int(m2.nl), "codon_frames": args.sites, "steps": 3, "kernel_ms": ms, "codon_frames_per_s": args.sites / (ms * 1e-3),
                                      "flop_per_codon_frame": r2["flop_per_codon_frame"], "roofline": r2}
                c2.close()
                del pool2, cod2, sc2
            except Exception as ex:
                other_models[name] = {"error": str(ex)[-500:]}
        torch.cuda.empty_cache()

    strong = None
    if args.scaling == "weak" and args.strong_steps > 0:
        lo, hi = bounds[rank]
        s_elapsed, c2, s_check = run_leg(hi - lo, lo, args.strong_steps, 1, shard_lows=[b[0] for b in bounds])
        del c2
        strong = {"scaling": "strong", "workload": "BASELINE.json configs[3] total: %d codon-frames (100 Mb x 6 frames) cut into contiguous "
                  "64-aligned ranges by dist.shard_bounds, one shared site stream, scores gathered to rank 0" % args.total_sites,
                  "total_codon_frames": args.total_sites, "steps": args.strong_steps, "warmup": 1,
                  "ms_per_step": s_elapsed / args.strong_steps * 1e3, "value": args.total_sites * args.strong_steps / s_elapsed,
                  "unit": "codon-frames/s", "n_gpus": world}
        if s_check is not None:
            strong["gather_check"] = s_check
    torch.cuda.empty_cache()

    if rank == 0:
        roof = roofline_of(tracks_info, m.nl, n_sites, kernel_ms)
        traffic, traffic_source = None, None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")  # per-launch HBM bytes from the committed PMC passes
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("sites_per_launch") == n_sites:
                    traffic = tj.get("hbm_bytes_per_launch")
                    traffic_source = "committed rocprofv3 --pmc passes of this command (%s), not measured in this run" % tj.get("source", "profiles/")
            except Exception:
                traffic = None
        out = {
            "metric": METRIC, "value": total_per_step * args.steps / elapsed, "unit": "codon-frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "build-tracks fixed (run_tracks) on a synthetic 58-row alignment, 6 frames: %d codon-frames on this GPU "
                                   "per step (%s), inputs resident in HBM; the kernel runs the context's default form (subtree tables: roofline.cherry_tables)"
                                   % (n_sites, "1 Mb per GPU, BASELINE.json configs[2]" if args.scaling == "weak" else
                                      "its contiguous share of %d = BASELINE.json configs[3]" % args.total_sites),
                       "model": MODEL, "n_leaves": int(m.nl), "sites_per_gpu": n_sites,
                       "unique_columns": POOL_SITES, "sharding": "columns, gather of scores to rank 0" if gathered else "none",
                       "ref_bases_per_s": total_per_step * args.steps / elapsed / 2.0},
            "roofline": dict(roof, traffic=traffic, traffic_source=traffic_source),
        }
        if gather_check is not None:
            out["gather_check"] = gather_check
        if strong is not None:
            out["strong"] = strong
        if other_models:
            out["other_models"] = other_models
        if table_forms:
            out["table_forms"] = table_forms
        if world == 1 and args.cpu_baseline_sites != 0:
            budget = args.cpu_baseline_sites
            if budget < 0:
                # ~2.5 k codon-frames/s/core incl. the per-alignment model set-up -> aim for ~10 s
                budget = int(min(codons_host.shape[1], max(20_000, 2500 * effective_cpus() * 10)))
            out["cpu_baseline"] = cpu_baseline(codons_host, budget)
            if args.end_to_end_runs > 0 or args.block_runs > 0:
                work = args.keep_files or tempfile.mkdtemp(prefix="pcsf_bench_")
                os.makedirs(work, exist_ok=True)
                try:
                    if args.end_to_end_runs > 0:
                        try:
                            e2e, cpu_e2e = end_to_end(local_rank, args.end_to_end_runs, out["cpu_baseline"]["value"], work)
                            out["end_to_end"] = e2e
                            out["cpu_baseline"].update(cpu_e2e)
                        except Exception as ex:  # the kernel-level line must survive a failure of the CLI leg
                            out["end_to_end"] = {"error": str(ex)[-1500:]}
                    if args.block_runs > 0:
                        try:
                            maf = os.path.join(work, "synthetic_%dbp_seed%d.maf" % (E2E_BASES, E2E_SEED))
                            if not os.path.exists(maf):
                                sys.path.insert(0, os.path.join(ROOT, "tools"))
                                import gen_synthetic_maf as gen
                                gen.generate(maf, bases=E2E_BASES, seed=E2E_SEED)
                            out["blocks"] = block_strategies(local_rank, maf, work, args.block_runs)
                        except Exception as ex:
                            out["blocks"] = {"error": str(ex)[-1500:]}
                finally:
                    if not args.keep_files:
                        shutil.rmtree(work, ignore_errors=True)
        if world > 1 and args.end_to_end_runs > 0:
            # the C++ driver over all GPUs of the node (one process, PCIe returns), while ranks 1.. wait below
            work = args.keep_files or tempfile.mkdtemp(prefix="pcsf_bench_")
            os.makedirs(work, exist_ok=True)
            try:
                out["end_to_end_multi"] = end_to_end_multi(world, work)
            except Exception as ex:
                out["end_to_end_multi"] = {"error": str(ex)[-1500:]}
            finally:
                if not args.keep_files:
                    shutil.rmtree(work, ignore_errors=True)
        print(json.dumps(out), flush=True)
    if world == 1 and gathered:
        dist.destroy_process_group()
    if world > 1:
        # ranks 1.. wait for rank 0's CLI leg on the HOST (a file, not a collective: an RCCL barrier would keep a spinning kernel on
        # the very GPUs the driver is being timed on)
        done = "/tmp/pcsf_bench_done_%s" % os.environ.get("MASTER_PORT", "0")
        if rank == 0:
            open(done, "w").close()
        else:
            t_wait = time.time()
            while not os.path.exists(done) and time.time() - t_wait < 1800:
                time.sleep(0.2)
        dist.barrier()
        if rank == 0:
            try:
                os.remove(done)
            except OSError:
                pass
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
