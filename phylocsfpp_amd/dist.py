"""Multi-GPU sharding of the scoring path: one process per GPU (torch.distributed, backend "nccl" =
RCCL over xGMI on MI355X; "gloo" on CPU for tests).

Every alignment column is independent given the (replicated, ~7 MB) model, so the reference's
OpenMP loop over byte-range jobs (src/phylocsf++build_tracks.hpp:88) becomes: rank r scores a
CONTIGUOUS range of the site stream in file order, and the only exchange is one gather of the
per-codon scores to rank 0, which owns the wig writers.  Concatenating the rank outputs restores
file order without a sort.  Whole-block strategies (MLE / OMEGA) shard by contiguous block ranges
balanced by site count; two floats per block come back.
"""
import numpy as np


def shard_bounds(n_items, world, align=1):
    """Contiguous, balanced [lo, hi) per rank; boundaries are multiples of `align` (64 = one
    work-group of the pruning kernel) except the last."""
    units = (n_items + align - 1) // align
    base, extra = divmod(units, world)
    bounds, lo = [], 0
    for r in range(world):
        hi = lo + (base + (1 if r < extra else 0)) * align
        bounds.append((min(lo, n_items), min(hi, n_items)))
        lo = hi
    return bounds


def shard_blocks(site_offsets, world):
    """Contiguous block ranges [b_lo, b_hi) per rank with roughly equal site counts (each block also
    costs a constant: its branch matrices are rebuilt at every optimiser step)."""
    off = np.asarray(site_offsets, dtype=np.int64)
    nb = len(off) - 1
    cost = np.diff(off) + 16          # ~16 site-equivalents of expm work per block and step
    cum = np.concatenate([[0], np.cumsum(cost)])
    cuts = [0]
    for r in range(1, world):
        target = cum[-1] * r / world
        cuts.append(int(np.searchsorted(cum, target, side="left")))
    cuts.append(nb)
    cuts = np.maximum.accumulate(np.clip(cuts, 0, nb))
    return [(int(cuts[r]), int(cuts[r + 1])) for r in range(world)]


def gather_to_root(local, counts, dst=0, group=None):
    """Gathers ragged 1-D tensors to rank `dst` in rank order.  local: this rank's tensor (length
    counts[rank]); counts: per-rank lengths known to every rank.  Returns the concatenation on
    `dst`, None elsewhere.  One padded dist.gather: on xGMI every peer has its own link to the root."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    width = max(int(max(counts)), 1)
    buf = torch.zeros(width, dtype=local.dtype, device=local.device)
    buf[:local.numel()] = local
    recv = [torch.empty_like(buf) for _ in range(world)] if rank == dst else None
    dist.gather(buf, recv, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([recv[r][:int(counts[r])] for r in range(world)])


class ScoreGatherPipeline:
    """Steady-state form of the gather: `depth` score buffers per rank, the gather of step k runs (async, on the
    collective's own stream) while step k+1 is being scored into the next buffer.

        buf = pipe.acquire()      # the buffer to score into; waits for the gather that last read it
        ... launch the scoring kernel writing `buf` on the current stream ...
        pipe.submit()             # starts the gather of `buf` to rank 0
        ...
        pipe.drain()              # all gathers done; on rank 0 pipe.result(i) = list of the ranks' buffers of slot i
    """

    def __init__(self, n_items, dtype, device, depth=2, dst=0, group=None):
        import torch
        import torch.distributed as dist
        self._dist, self.group, self.dst, self.depth = dist, group, dst, depth
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.bufs = [torch.empty(n_items, dtype=dtype, device=device) for _ in range(depth)]
        self.recv = [[torch.empty(n_items, dtype=dtype, device=device) for _ in range(self.world)] if self.rank == dst else None
                     for _ in range(depth)]
        self.pending = [None] * depth
        self.k = 0
        self._cur = None

    def acquire(self):
        slot = self.k % self.depth
        if self.pending[slot] is not None:
            self.pending[slot].wait()   # orders the current stream behind that gather; no host block on GPU streams
            self.pending[slot] = None
        self._cur = slot
        return self.bufs[slot]

    def submit(self):
        slot = self._cur
        self.pending[slot] = self._dist.gather(self.bufs[slot], self.recv[slot], dst=self.dst, group=self.group, async_op=True)
        self.k += 1
        return slot

    def drain(self):
        for slot in range(self.depth):
            if self.pending[slot] is not None:
                self.pending[slot].wait()
                self.pending[slot] = None

    def result(self, slot):
        return self.recv[slot]

    def check(self, slot, n_valid, recompute=None):
        """Proves the gather of `slot` AFTER drain(): every rank hashes the first n_valid items of ITS OWN buffer of that slot (64-bit,
        position-weighted, wrapping), the (hash, n_valid) pairs are all-gathered, and rank `dst` hashes what it RECEIVED from every
        rank and compares.  recompute(rank, n) -> tensor (optional, rank dst only): the items rank `rank` should have produced,
        computed again on `dst` -- compared bit for bit with what arrived.  Returns on dst a dict
        {"ranks_seen", "equal", "ranks_differing", "recomputed_equal"}; None elsewhere.  A collective: every rank must call it."""
        import torch
        own = torch.stack([content_hash(self.bufs[slot][:n_valid]), torch.tensor(int(n_valid), dtype=torch.int64, device=self.bufs[slot].device)])
        pairs = [torch.empty_like(own) for _ in range(self.world)]
        self._dist.all_gather(pairs, own, group=self.group)
        if self.rank != self.dst:
            return None
        differing, recomputed = [], (True if recompute is not None else None)
        for r in range(self.world):
            want, n = int(pairs[r][0].item()), int(pairs[r][1].item())
            if int(content_hash(self.recv[slot][r][:n]).item()) != want:
                differing.append(r)
            if recompute is not None:
                again = recompute(r, n)
                if not torch.equal(again.contiguous().view(torch.uint8), self.recv[slot][r][:n].contiguous().view(torch.uint8)):   # bytes: NaN == NaN
                    recomputed = False
        return {"ranks_seen": len(pairs), "equal": not differing, "ranks_differing": differing, "recomputed_equal": recomputed,
                "items_per_rank": [int(p_[1].item()) for p_ in pairs]}


def content_hash(t):
    """A 64-bit hash of a tensor's BYTES as a 0-d int64 tensor on its device: the 8-byte words times odd position weights, summed with
    wrap-around (any changed, missing or misplaced word changes it with probability 1 - 2^-64 for random data)."""
    import torch
    if t.numel() == 0:
        return torch.zeros((), dtype=torch.int64, device=t.device)
    raw = t.contiguous().view(torch.uint8)
    pad = (-raw.numel()) % 8
    if pad:
        raw = torch.cat([raw, torch.zeros(pad, dtype=torch.uint8, device=raw.device)])
    w = raw.view(torch.int64)
    weights = torch.arange(w.numel(), dtype=torch.int64, device=w.device) * 2 + 1
    weights = weights * -7046029254386353131      # 0x9E3779B97F4A7C15 as int64: odd, so every weight stays odd
    return (w * weights).sum()


def score_sites_sharded(score_fn, codons, group=None, align=64, to_device=None):
    """run_tracks over `codons` (uint8 [n_leaves, n_sites], the same array on every rank or at least
    this rank's columns valid) sharded across the process group.
    score_fn(codons_slice ndarray) -> 1-D torch tensor of per-codon scores for that slice.
    Returns the full score vector (torch tensor) on rank 0, None on the other ranks."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    n = codons.shape[1]
    bounds = shard_bounds(n, world, align)
    lo, hi = bounds[rank]
    local = score_fn(np.ascontiguousarray(codons[:, lo:hi]))
    if to_device is not None:
        local = local.to(to_device)
    return gather_to_root(local, [b[1] - b[0] for b in bounds], dst=0, group=group)


def score_blocks_sharded(score_fn, codons, site_offsets, group=None):
    """run() over a CSR batch of blocks sharded by contiguous block ranges.
    score_fn(codons_slice, offsets_slice) -> (phylo float32 ndarray, anc float32 ndarray).
    Returns (phylo, anc) ndarrays on rank 0, (None, None) elsewhere."""
    import torch
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    off = np.asarray(site_offsets, dtype=np.uint64)
    ranges = shard_blocks(off, world)
    b_lo, b_hi = ranges[rank]
    s_lo, s_hi = int(off[b_lo]), int(off[b_hi])
    sub = np.ascontiguousarray(codons[:, s_lo:max(s_hi, s_lo + 1)]) if s_hi > s_lo else np.full((codons.shape[0], 1), 64, np.uint8)
    ph, an = score_fn(sub, (off[b_lo:b_hi + 1] - off[b_lo]).astype(np.uint64))
    dev = "cuda" if dist.get_backend(group) == "nccl" else "cpu"
    both = torch.from_numpy(np.concatenate([np.asarray(ph, np.float32), np.asarray(an, np.float32)])).to(dev)
    counts = [2 * (r[1] - r[0]) for r in ranges]
    allv = gather_to_root(both, counts, dst=0, group=group)
    if rank != 0:
        return None, None
    allv = allv.cpu().numpy()
    phs, ans, pos = [], [], 0
    for r in ranges:
        k = r[1] - r[0]
        phs.append(allv[pos:pos + k])
        ans.append(allv[pos + k:pos + 2 * k])
        pos += 2 * k
    return np.concatenate(phs), np.concatenate(ans)
