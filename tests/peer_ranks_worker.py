"""Several ranks of the sharded sweep on ONE GPU (run by tests/test_gpu_multi.py in a subprocess):
every rank is a context with its own stream and its own shard of the market; the "symmetric" buffers
are ordinary device allocations of this process.  Checks, over 30 evaluations with alternating launch
order: every rank's {Psi, acc} is finite, bit-identical across ranks, and equal to the rank-ordered sum
of the ranks' local results."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import cfmmrouter_amd as cr
from cfmmrouter_amd import synth
from cfmmrouter_amd.dist import shard_batches
from helpers import oracle_sweep, rel_to_max

world = int(sys.argv[1])
n = 512
market = [synth.product_pools(240_000, n, seed=71), synth.geomean_pools(80_000, n, seed=72)]
count = n + 1
bufs = [torch.zeros(4 * count, dtype=torch.float64, device="cuda") for _ in range(world)]   # cfmm_peer_buffer_bytes
ptrs = [b.data_ptr() for b in bufs]
ranks = [cr.DeviceBackend(n, shard_batches(market, r, world)) for r in range(world)]
for be in ranks:
    be.ctx.set_option("alternate", 0)   # local and sharded sweeps are compared bit for bit: same tile direction every time
outs = [torch.zeros(count, dtype=torch.float64, device="cuda") for _ in range(world)]
loc = [torch.zeros(count, dtype=torch.float64, device="cuda") for _ in range(world)]
torch.cuda.synchronize()
seq = 0
for it in range(30):
    v = synth.sweep_prices(n, seed=700 + it)
    vt = torch.from_numpy(v).cuda()
    torch.cuda.synchronize()
    for r, be in enumerate(ranks):                     # local results first (sharding off)
        be.ctx.set_peers([], 0, 0, 0)
        be.ctx.sweep_dev(vt.data_ptr(), loc[r].data_ptr(), it % 2 == 0)
    torch.cuda.synchronize()
    want = loc[0].clone()
    for r in range(1, world):
        want += loc[r]                                 # rank order, as reduce_gather adds
    for r, be in enumerate(ranks):
        be.ctx.set_peers(ptrs, world, r, seq)
    order = list(range(world)) if it % 3 else list(reversed(range(world)))   # launch order must not matter
    for r in order:
        ranks[r].ctx.sweep_dev(vt.data_ptr(), outs[r].data_ptr(), it % 2 == 0)
    torch.cuda.synchronize()
    seq += 1
    for r in range(world):
        assert bool(torch.isfinite(outs[r]).all()), ("timeout", it, r)
        assert torch.equal(outs[r], want), ("mismatch", it, r)
psi_o = oracle_sweep(market, n, v, nthreads=8)[2]
assert rel_to_max(outs[0].cpu().numpy()[:n], psi_o) <= 1e-12
for be in ranks:
    be.close()
print(f"PEER_RANKS_OK world={world}")
