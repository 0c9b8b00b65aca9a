"""Single-rank exercise of the peer all-reduce + sharded router under torchrun (RCCL backend)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, torch.distributed as dist
import cfmmrouter_amd as cr
from cfmmrouter_amd import synth, dist as crd
lr = int(os.environ.get("LOCAL_RANK", 0)); torch.cuda.set_device(lr)
dist.init_process_group("nccl", device_id=torch.device("cuda", lr))
dev = torch.device("cuda", lr)
par = crd.PeerAllReduce.create(257, None, dev)
print("peer path available:", par is not None)
if par is not None:
    out = torch.empty(257, dtype=torch.float64, device=dev)
    for k in range(6):
        x = torch.arange(257, dtype=torch.float64, device=dev) * (k + 1)
        par.slot().copy_(x); par.reduce(out); torch.cuda.synchronize()
        assert torch.equal(out, x), k
    print("6 consecutive reduces ok (parity/flags)")
n = 32
market = [synth.product_pools(5000, n, seed=1), synth.geomean_pools(2000, n, seed=2)]
obj = cr.LinearNonnegative(synth.linear_prices(n, seed=1))
r = crd.ShardedRouter(obj, market, n, device=lr)
cr.route_(r, v=np.ones(n))
single = cr.Router(obj, market, n, device=lr)
cr.route_(single, v=np.ones(n))
print("sharded(1) vs single netflow diff:", float(np.max(np.abs(cr.netflows(r) - cr.netflows(single)))),
      "in-library peer path:", isinstance(r._backend, cr.DeviceBackend))
r2 = crd.ShardedRouter(obj, market, n, device=lr)
cr.route_(r2, v=np.ones(n), solver="native")
cr.route_(single, v=np.ones(n), solver="native")
print("native one-call sharded route vs single:", float(np.max(np.abs(cr.netflows(r2) - cr.netflows(single)))), r2.info)
dist.destroy_process_group()
