"""N > 1 path on CPU: world_size-2 (and 3) gloo process groups exercise the sharding + gather of
phylocsfpp_amd/dist.py with the CPU oracle standing in for the per-rank GPU scorer."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from phylocsfpp_amd import dist as pdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_sites, out_path):
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    sys.path.insert(0, ROOT)
    import oracle_binding as ob
    from phylocsfpp_amd import synth
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = ob.OracleModel(os.path.join(ROOT, "tests", "golden", "models", "53birds"))
    cod = synth.synthetic_codon_columns(m.child1, m.child2, m.branch_len, m.S_c, m.pi_c, m.S_nc, m.pi_nc, max(n_sites, 1), seed=9)[:, :n_sites]
    full = pdist.score_sites_sharded(lambda c: torch.from_numpy(m.run_tracks(c)), cod, align=64)
    # whole-block FIXED scores through the block sharding
    off = np.unique(np.concatenate([[0, n_sites], np.random.default_rng(1).integers(0, n_sites + 1, 7)])).astype(np.uint64)

    def blocks(c, o):
        ph, an = [], []
        for b in range(len(o) - 1):
            p, a = m.run(ob.FIXED, np.ascontiguousarray(c[:, int(o[b]):int(o[b + 1])]))
            ph.append(p)
            an.append(a)
        return np.array(ph, np.float32), np.array(an, np.float32)
    ph, an = pdist.score_blocks_sharded(blocks, cod, off)
    if rank == 0:
        want = m.run_tracks(cod)
        wph, wan = blocks(cod, off)
        np.save(out_path, np.array([float(np.array_equal(full.numpy(), want)), float(np.array_equal(ph, wph) and np.array_equal(an, wan)),
                                    float(len(full))]))
    else:
        assert full is None and ph is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_sites", [(2, 333), (2, 1), (3, 200), (2, 64)])
def test_sharded_scores_equal_unsharded(tmp_path, world, n_sites):
    out = str(tmp_path / "res.npy")
    mp.spawn(_worker, args=(world, _free_port(), n_sites, out), nprocs=world, join=True)
    res = np.load(out)
    assert res[0] == 1.0 and res[1] == 1.0 and res[2] == n_sites


def test_shard_bounds_properties():
    for n in (0, 1, 63, 64, 65, 1000, 2_000_000):
        for w in (1, 2, 4, 8):
            b = pdist.shard_bounds(n, w, 64)
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert all(lo % 64 == 0 or lo == n for lo, _ in b)
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 64 or n < 64 * w
    off = np.array([0, 5, 5, 900, 910, 1500, 1501, 4000], dtype=np.uint64)
    for w in (1, 2, 3, 8):
        r = pdist.shard_blocks(off, w)
        assert r[0][0] == 0 and r[-1][1] == 7 and all(r[i][1] == r[i + 1][0] for i in range(w - 1))


def _pipeline_worker(rank, world, port, out_path):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pipe = pdist.ScoreGatherPipeline(1000, torch.float64, torch.device("cpu"), depth=2)
    ok = True
    seen = {}
    for step in range(7):
        buf = pipe.acquire()
        if rank == 0 and step >= 2:     # the gather that used this slot two steps ago has completed by now
            got = pipe.result(step % 2)
            ok = ok and all(bool((got[r] == 100.0 * (step - 2) + r).all()) for r in range(world))
        buf.fill_(100.0 * step + rank)
        seen[step] = pipe.submit()
    pipe.drain()
    if rank == 0:
        for step in (5, 6):
            got = pipe.result(seen[step])
            ok = ok and all(bool((got[r] == 100.0 * step + r).all()) for r in range(world))
        np.save(out_path, np.array([float(ok)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_overlapped_gather_pipeline(tmp_path, world):
    """bench.py's N > 1 steady state: the gather of step k overlaps the scoring of step k+1 (two buffers per rank)."""
    out = str(tmp_path / "pipe.npy")
    mp.spawn(_pipeline_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert np.load(out)[0] == 1.0


def _gather_check_worker(rank, world, port, out_path):
    import sys
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    width = 1000
    n_local = width - 37 * rank                      # ragged shards, padded for the gather like bench.py's strong leg
    pipe = pdist.ScoreGatherPipeline(width, torch.float64, torch.device("cpu"), depth=2)

    def produce(r, n, step):                          # what rank r computes in a step (stands for the scoring kernel)
        g = torch.Generator().manual_seed(1000 * step + r)
        v = torch.randn(n, dtype=torch.float64, generator=g)
        v[::97] = float("nan")                        # NaN scores must compare equal to themselves
        return v

    last = None
    for step in range(3):
        buf = pipe.acquire()
        buf[:n_local] = produce(rank, n_local, step)
        last = pipe.submit()
    pipe.drain()
    good = pipe.check(last, n_local, recompute=(lambda r, n: produce(r, n, 2)) if rank == 0 else None)
    wrong_recompute = pipe.check(last, n_local, recompute=(lambda r, n: produce(r, n, 1)) if rank == 0 else None)
    if rank == 0:
        pipe.result(last)[world - 1][5] += 1.0        # one word of the last rank's received buffer changes
    bad = pipe.check(last, n_local)
    if rank == 0:
        ok = (good == {"ranks_seen": world, "equal": True, "ranks_differing": [], "recomputed_equal": True,
                       "items_per_rank": [width - 37 * r for r in range(world)]}
              and wrong_recompute["equal"] and wrong_recompute["recomputed_equal"] is False
              and bad["equal"] is False and bad["ranks_differing"] == [world - 1] and bad["recomputed_equal"] is None)
        np.save(out_path, np.array([float(ok)]))
    else:
        assert good is None and bad is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gather_check_proves_what_rank_0_received(tmp_path, world):
    """bench.py --gpus N, after the timed region: every rank's hash of its own last buffer against rank 0's hash of what it received
    from that rank, and rank 0's recomputation of every rank's shard (ScoreGatherPipeline.check)."""
    out = str(tmp_path / "check.npy")
    mp.spawn(_gather_check_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    assert np.load(out)[0] == 1.0


def test_content_hash_sees_every_word_and_its_position():
    x = torch.arange(1, 1001, dtype=torch.float64)
    h = int(pdist.content_hash(x))
    y = x.clone(); y[123] += 1e-9
    z = x.clone(); z[[10, 11]] = z[[11, 10]]
    assert int(pdist.content_hash(x.clone())) == h and int(pdist.content_hash(y)) != h and int(pdist.content_hash(z)) != h
    assert int(pdist.content_hash(x[:999])) != h and int(pdist.content_hash(torch.zeros(0))) == 0
    assert int(pdist.content_hash(torch.arange(13, dtype=torch.uint8))) != int(pdist.content_hash(torch.arange(14, dtype=torch.uint8)))


def test_bench_site_stream_is_the_same_for_every_sharding():
    """bench.py materialises its synthetic site stream per rank on the rank's own GPU from a counter-based generator
    (column i = pool column hash(i, seed)): the strong-scaling shards of any world size must tile the one-GPU stream."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    rng = np.random.default_rng(0)
    pool = torch.from_numpy(rng.integers(0, 65, size=(5, bench.POOL_SITES), dtype=np.uint8))
    total = 100_003
    whole = bench.sample_columns(pool, 0, total, seed=7)
    assert whole.shape == (5, total) and whole.dtype == torch.uint8
    assert len(torch.unique(whole[0])) > 30 and not torch.equal(whole[:, :1000], bench.sample_columns(pool, 0, 1000, seed=8))
    for world in (1, 2, 3, 8):
        bounds = pdist.shard_bounds(total, world, 64)
        assert bounds[0][0] == 0 and bounds[-1][1] == total and all(b[0] % 64 == 0 for b in bounds)
        parts = [bench.sample_columns(pool, lo, hi, seed=7) for lo, hi in bounds]
        assert torch.equal(torch.cat(parts, dim=1), whole)


def test_bench_host_ceiling_of_a_multi_device_run():
    """bench.py --gpus N puts the host stages' ceiling next to end_to_end_multi (stage 1 and the writers run once whatever the number of
    GPUs, the scoring backends share theirs): the arithmetic on a made-up run's statistics."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    run = {"codon_frames": 200_000_000, "stage_busy": {"maf_read": 0.34, "bls": 0.0, "batch_packing": 0.6, "scoring_backend": 5.5, "wig_output": 0.15, "smoothing": 0.0}}
    c = bench.host_ceiling(run, 8)
    assert abs(c["stage1_parse_and_pack_seconds"] - 0.94) < 1e-12 and c["stage3_writers_seconds"] == 0.15
    assert abs(c["codon_frames_per_s"] - 200_000_000 / 0.94) < 1e-3 and abs(c["scoring_backends_seconds_per_device"] - 5.5 / 8) < 1e-12
    assert c["host_threads"] >= 1
    run["stage_busy"]["wig_output"] = 2.0          # the writers as the slowest stage
    assert abs(bench.host_ceiling(run, 2)["codon_frames_per_s"] - 1e8) < 1e-3


def test_bench_reads_the_statistics_of_a_ranks_run(tmp_path):
    """bench.py's run_cli on a --ranks run (one statistics line per process): the ranks' work adds up to the plain run's, the run lasts as
    long as its slowest rank.  On the test command line (CPU, file-based gather): the same driver output the GPU command line prints."""
    import subprocess
    import sys
    import bench
    syn = tmp_path / "syn.maf"
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_synthetic_maf.py"), str(syn), "--bases", "30000", "--start", "985000", "--seed", "5"],
                          stdout=subprocess.DEVNULL)
    env = dict(os.environ, PCSF_ORACLE_BACKEND_FAKE="1")
    plain = bench.run_cli(bench.ORACLE_CLI, str(syn), str(tmp_path / "plain"), env=env)
    ranks = bench.run_cli(bench.ORACLE_CLI, str(syn), str(tmp_path / "ranks"), ["--ranks", "3", "--batch-sites", "6000", "--window-bytes", "150000"], env=env)
    assert "ranks" not in plain and ranks["ranks"] == 3 and ranks["devices"] == [0, 1, 2]
    for k in ("alignments", "ref_bases", "codon_frames"):
        assert ranks[k] == plain[k], k
    assert len(ranks["seconds_per_rank"]) == 3 and ranks["seconds"] == max(ranks["seconds_per_rank"])
    assert sum(ranks["codon_frames_per_rank"]) == plain["codon_frames"]
    assert bench.differing_lines(str(tmp_path / "plain"), str(tmp_path / "ranks")) == 0
