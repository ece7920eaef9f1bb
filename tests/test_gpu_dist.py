"""The RCCL code path of phylocsfpp_amd/dist.py on the one GPU a test box has: a world-size-1 "nccl" process group (backend nccl IS
RCCL on ROCm) carries the padded gather, the pipelined gather of bench.py and the sharded scoring with the real GPU scorer; results must
equal direct scoring.  (world sizes 2 and 3 run on gloo in tests/test_dist_gloo.py; a node with several GPUs is the driver's.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import numpy as np, torch, torch.distributed as dist
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = "29653"
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import phylocsfpp_amd as pa
from phylocsfpp_amd import dist as pdist, synth
m = pa.load_model("58mammals")
ctx = m.context(device=0)
cod = synth.synthetic_codon_columns(m.child1, m.child2, m.branch_len, m.S_c, m.pi_c, m.S_nc, m.pi_nc, 5000, seed=4)
direct = ctx.score_sites_fixed(cod)
# sharded scoring + one padded gather to rank 0 (device tensors through RCCL)
full = pdist.score_sites_sharded(lambda c: torch.from_numpy(ctx.score_sites_fixed(c)).cuda(), cod, align=64)
assert full.is_cuda and np.array_equal(full.cpu().numpy(), direct)
# ragged gather: counts smaller than the padded width
t = torch.arange(37, dtype=torch.float64, device="cuda")
g = pdist.gather_to_root(t, [37])
assert torch.equal(g, t)
# the pipelined gather of bench.py: two buffers, gathers overlapping the next step's scoring
d = torch.from_numpy(cod).cuda()
pipe = pdist.ScoreGatherPipeline(cod.shape[1], torch.float64, torch.device("cuda", 0), depth=2, dst=0)
for step in range(5):
    ctx.score_sites_fixed_torch(d, out=pipe.acquire())
    slot = pipe.submit()
pipe.drain()
torch.cuda.synchronize()
for slot in range(2):
    assert np.array_equal(pipe.result(slot)[0].cpu().numpy(), direct)
# bench.py's proof of the gather (device hashes through an RCCL all-gather, rank 0's recomputation)
def again(r, n):
    o = torch.empty(n, dtype=torch.float64, device="cuda")
    ctx.score_sites_fixed_torch(d, out=o)
    torch.cuda.synchronize()
    return o
chk = pipe.check(slot, cod.shape[1], recompute=again)
assert chk["ranks_seen"] == 1 and chk["equal"] and chk["recomputed_equal"] is True and chk["items_per_rank"] == [cod.shape[1]], chk
pipe.result(slot)[0][17] += 1.0
chk = pipe.check(slot, cod.shape[1])
assert chk["equal"] is False and chk["ranks_differing"] == [0], chk
# whole-block strategies through the block sharding (two floats per block gathered)
off = np.array([0, 40, 41, 300, 5000], dtype=np.uint64)
ph, an = pdist.score_blocks_sharded(lambda c, o: ctx.score_blocks(pa.FIXED, c, o, compute_anc=True), cod, off)
ph0, an0 = ctx.score_blocks(pa.FIXED, cod, off, compute_anc=True)
assert np.array_equal(ph, ph0) and np.array_equal(an, an0)
dist.barrier()
dist.destroy_process_group()
ctx.close()
print("nccl world-1 ok")
'''


def test_rccl_gather_paths_on_a_world_of_one():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", SCRIPT % {"root": ROOT}], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "nccl world-1 ok" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
