"""Run under `python -m torch.distributed.run --nproc-per-node 1` on the MI355X box by
tests/test_gpu_dist.py: the multi-process sharded paths of cfmmrouter.jl_amd/dist.py at world = 1 --
real RCCL ("nccl") process group, the library's IPC peer buffers, cfmm_set_peers (fold + gather in one
launch, pre-armed route!) and the RCCL fall-back -- each compared with the unsharded router.  Prints one JSON line."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist

import cfmmrouter_amd as cr
from cfmmrouter_amd import dist as crd
from cfmmrouter_amd import synth

lr = int(os.environ.get("LOCAL_RANK", 0))
torch.cuda.set_device(lr)
dist.init_process_group("nccl", device_id=torch.device("cuda", lr))
dev = torch.device("cuda", lr)
out = {"world": dist.get_world_size(), "backend": dist.get_backend()}

# sharded routers: in-library peer path, then RCCL all-reduce driven from Python
n = 512
market = [synth.product_pools(120_000, n, seed=1), synth.geomean_pools(30_000, n, seed=2)]
obj = cr.LinearNonnegative(synth.linear_prices(n, seed=1))
single = cr.Router(obj, market, n, device=lr)
v = synth.sweep_prices(n, seed=3)
cr.find_arb_(single, v)
psi_fixed, D_fixed, L_fixed = cr.netflows(single).copy(), single.Δs.copy(), single.Λs.copy()
# the three collectives of ShardedRouter, in its order of preference: peer exchange inside the fold launch, RCCL inside the
# library (cfmm_rccl_init_rank: ncclAllReduce behind every fold), torch.distributed
for name, no_peer, no_lib in (("peer", "0", "0"), ("lib_rccl", "1", "0"), ("rccl", "1", "1")):
    os.environ["CFMM_AMD_NO_PEER"] = no_peer
    os.environ["CFMM_AMD_NO_LIB_RCCL"] = no_lib
    r = crd.ShardedRouter(obj, market, n, device=lr)
    cr.find_arb_(r, v)
    rec = {"in_library_collective": isinstance(r._backend, cr.DeviceBackend), "collective": r.collective,
           "fixed_v_netflow_equal": bool(np.array_equal(cr.netflows(r), psi_fixed)),
           "trades_equal": bool(np.array_equal(r.Δs, D_fixed) and np.array_equal(r.Λs, L_fixed))}
    for solver in ("native", "scipy"):
        cr.route_(single, v=np.ones(n), solver=solver)
        t0 = time.perf_counter()
        cr.route_(r, v=np.ones(n), solver=solver)
        rec[f"route_{solver}_ms"] = 1e3 * (time.perf_counter() - t0)
        rec[f"route_{solver}_netflow_rel_diff"] = float(np.max(np.abs(cr.netflows(r) - cr.netflows(single))) /
                                                        np.max(np.abs(cr.netflows(single))))
        rec[f"route_{solver}_evaluations"] = r.info.get("funcalls")
    out[name] = rec
    r.close()
single.close()
print("DIST_WORLD1 " + json.dumps(out), flush=True)
dist.destroy_process_group()
