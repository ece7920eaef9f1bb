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

Prints ONE JSON line on rank 0.  Extra objects: "roofline" (dominant kernel vs the fp64-MFMA
peak; duration measured with HIP events on the launch stream) and "cpu_baseline" (the CPU
oracle, a reference-structured port, timed on this host's cores; rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

MODEL = "58mammals"
SITES_PER_GPU = 2_000_000           # 1 Mb reference x 2 strands x 3 frames / 3 bases per codon
POOL_SITES = 1 << 17                # unique synthetic columns; the batch samples them with replacement
ALIGNMENT_FRAMES = 600              # ~100 concatenated alignments x 6 frames in a 1 Mb MAF
PEAK_FP64_MFMA_TFLOPS = 78.6        # MI355X fp64 matrix peak (vendor); 78.0 measured by tools/mfma_f64_bench
METRIC = "alignment codons/sec (build-tracks, 58mammals) at 1/2/4/8 MI355X"


def flop_per_site(nl):
    """both models; internal edges as 64x64 mat-vec, element-wise products, root dot."""
    return 2 * ((nl - 2) * 2 * 64 * 64 + (nl - 1) * 64 + 2 * 64)


def load_model_arrays(name):
    """Model files -> arrays (host-side mirror of load_model, reference src/models.hpp:1757).
    Only the file readers of the test binding are used here; nothing is computed by the oracle."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob
    return ob.OracleModel(os.path.join(ROOT, "tests", "golden", "models", name))


def make_batch(m, n_sites, seed):
    from phylocsfpp_amd import synth
    pool = synth.synthetic_codon_columns(m.child1, m.child2, m.branch_len, m.S_c, m.pi_c, m.S_nc, m.pi_nc,
                                         POOL_SITES, seed=seed)
    rng = np.random.default_rng(seed + 1)
    idx = rng.integers(0, POOL_SITES, size=n_sites)
    return np.ascontiguousarray(pool[:, idx])


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


def cpu_baseline(m, codons, budget_sites):
    """Times the oracle's reference-structured run_tracks loop (model re-instantiated per
    alignment x frame, scalar pruning per codon, OpenMP schedule(dynamic,1) over alignments)."""
    import oracle_binding as ob  # noqa: F401  (cpu_baseline leg: the oracle is the thing timed here)
    n = min(budget_sites, codons.shape[1])
    n_units = max(1, int(round(ALIGNMENT_FRAMES * n / SITES_PER_GPU)))
    off = np.linspace(0, n, n_units + 1).astype(np.uint64)
    sample = np.ascontiguousarray(codons[:, :n])
    threads = effective_cpus()
    t0 = time.perf_counter()
    m.baseline_run_tracks(sample, off, threads=threads)
    dt = time.perf_counter() - t0
    return {"value": n / dt, "unit": "codon-frames/s", "cores": threads, "kind": "port",
            "sample": "%d codon-frames in %d alignment-frames of the same batch, %.1f s; reference-structured port "
                      "(model re-instantiated per alignment-frame, scalar pruning), OpenMP schedule(dynamic,1) over "
                      "alignment-frames on %d threads (cgroup CPU quota of this host)" % (n, n_units, dt, threads)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--sites", type=int, default=SITES_PER_GPU, help="codon-frames per GPU per step")
    ap.add_argument("--cpu-baseline-sites", type=int, default=-1,
                    help="codon-frames for the CPU baseline sample (default: sized for ~10-30 s; 0 disables)")
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
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    import phylocsfpp_amd as pa
    m = load_model_arrays(MODEL)
    ctx = pa.ScoringContext(m.child1, m.child2, m.branch_len, m.S_c, m.pi_c, m.S_nc, m.pi_nc, device=local_rank)
    n_sites = args.sites
    codons_host = make_batch(m, n_sites, seed=20261002 + rank)
    codons = torch.from_numpy(codons_host).to(dev)
    scores = torch.empty(n_sites, dtype=torch.float64, device=dev)
    # N > 1: two score buffers per rank; the gather of step k (RCCL, its own stream) runs while step k+1 is scored
    pipe = None
    if world > 1:
        from phylocsfpp_amd import dist as pdist
        pipe = pdist.ScoreGatherPipeline(n_sites, torch.float64, dev, depth=2, dst=0)

    def step():
        if pipe is None:
            ctx.score_sites_fixed_torch(codons, out=scores)
        else:
            ctx.score_sites_fixed_torch(codons, out=pipe.acquire())
            pipe.submit()

    def fence():
        if pipe is not None:
            pipe.drain()       # every rank's scores of every step have arrived on rank 0
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # dominant-kernel duration: HIP events recorded on the launch stream around each launch
    ctx.set_timing(True)
    durs = []
    for _ in range(max(3, min(args.steps, 10))):
        ctx.score_sites_fixed_torch(codons, out=scores)
        durs.append(ctx.last_kernel_ms())
    ctx.set_timing(False)
    torch.cuda.synchronize()
    kernel_ms = float(np.mean(durs))

    if rank == 0:
        total_sites = n_sites * world * args.steps
        fl = flop_per_site(m.nl)
        achieved = n_sites * fl / (kernel_ms * 1e-3) / 1e12
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")  # per-launch HBM bytes from the committed PMC passes
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("sites_per_launch") == n_sites:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": METRIC, "value": total_sites / elapsed, "unit": "codon-frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "build-tracks fixed (run_tracks) on a synthetic 58-row 1 Mb alignment, 6 frames: "
                                   "%d codon-frames per GPU per step, inputs resident in HBM" % n_sites,
                       "model": MODEL, "n_leaves": int(m.nl), "sites_per_gpu": n_sites,
                       "unique_columns": POOL_SITES, "sharding": "columns, gather of scores to rank 0" if world > 1 else "none",
                       "ref_bases_per_s": total_sites / elapsed / 2.0},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": PEAK_FP64_MFMA_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_FP64_MFMA_TFLOPS, "traffic": traffic,
                         "kernel": "pcsf::prune_fixed", "kernel_ms": kernel_ms, "flop_per_codon_frame": fl},
        }
        if world == 1:
            budget = args.cpu_baseline_sites
            if budget < 0:
                # ~2.5 k codon-frames/s/core incl. the per-alignment model set-up -> aim for ~15 s
                budget = int(min(n_sites, max(20_000, 2500 * effective_cpus() * 15)))
            if budget > 0:
                out["cpu_baseline"] = cpu_baseline(m, codons_host, budget)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
