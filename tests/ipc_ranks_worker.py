"""Two (or more) PROCESSES sharing one MI355X, launched by tests/test_gpu_dist.py through
torch.distributed.run with the gloo backend (RCCL refuses two ranks on one device): the sharded route!
with the library's own IPC peer buffers (cfmm_peer_buffer_alloc/_open) and the fold + gather launch --
a real multi-process run of the N > 1 path, minus the xGMI links.  Prints one JSON line on rank 0."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist

import cfmmrouter_amd as cr
from cfmmrouter_amd import dist as crd
from cfmmrouter_amd import synth

torch.cuda.set_device(0)
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
if "--selftest-fail" in sys.argv:     # ONE rank reports a disagreement in the start-up check: all ranks must fall back
    os.environ["CFMM_AMD_PEER_SELFTEST_FAIL"] = "1"
n = 512
CONFIG4 = "--config4" in sys.argv      # BASELINE config 4 at its stated size: 4M ProductTwoCoin pools, 512 tokens, world = 8 shards
if CONFIG4:
    from benchlib.workloads import build_market, objective_for
    market = build_market("config4", 0, 1, "weak")
    obj = objective_for("config4", n)
else:
    market = [synth.product_pools(300_000, n, seed=81), synth.geomean_pools(100_000, n, seed=82),
              synth.bounded_product_pools(60_000, n, seed=83, consistent=True)]
    obj = cr.LinearNonnegative(synth.linear_prices(n, seed=81))
r = crd.ShardedRouter(obj, market, n, device=0)
if isinstance(r._backend, cr.DeviceBackend):
    # Several ranks on ONE GPU: a pre-armed launch polls for its prices while it occupies its CUs; with the ranks' shards of
    # the SAME evaluation queued on the same device, one rank's waiting launch would starve the evaluation it waits for.
    # (One rank per GPU -- the deployment -- has no such coupling; the library switches arming off by itself only for
    # multi-device contexts that list a device twice.)
    r._backend.ctx.set_option("armed", 0)
out = {"world": world, "in_library_collective": isinstance(r._backend, cr.DeviceBackend),
       "buffers": type(getattr(r._backend, "peer", None)).__name__, "collective": r.collective,
       "guard": hasattr(r, "_guard")}
if "--fail-route-once" in sys.argv and isinstance(r._backend, cr.DeviceBackend) and rank == world - 1:
    # the last rank's first native route! fails BEFORE it publishes anything: the other ranks run into the peer time-out,
    # every rank votes, the exchange is re-aligned and the route repeated (PeerGuard, router.py::_route_native)
    real_route, state = r._backend.ctx.route, {"failed": False}

    def flaky_route(*a, **kw):
        if not state["failed"]:
            state["failed"] = True
            raise cr.CFMMDeviceError("injected failure (test)")
        return real_route(*a, **kw)

    r._backend.ctx.route = flaky_route
v = synth.sweep_prices(n, seed=84)
cr.find_arb_(r, v)
psi_fixed = cr.netflows(r).copy()
rows_ok = True
if CONFIG4:
    # every rank: ALL trade rows of its shard bit-equal to the CPU restatement of the reference at the same prices
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import oracle_sweep
    Do, Lo, _, _ = oracle_sweep(crd.shard_batches(market, rank, world), n, v, nthreads=8)
    rows_ok = bool(np.array_equal(r.Δs, Do) and np.array_equal(r.Λs, Lo))
cr.route_(r, v=np.ones(n), solver="native")
psi_native, v_native, ev_native = cr.netflows(r).copy(), r.v.copy(), r.info["funcalls"]
out["collective_retries"] = getattr(r, "collective_retries", 0)
cr.route_(r, v=np.ones(n), solver="scipy")
psi_scipy = cr.netflows(r).copy()
gathered = [None] * world
dist.all_gather_object(gathered, (psi_fixed, psi_native, v_native, len(r.Δs), rows_ok))
if rank == 0 and CONFIG4:
    # against the CPU restatement (fixed v: serial pool-order sums over all 4M pools) and the Fortran L-BFGS-B fixture
    _, _, psi_o, _ = oracle_sweep(market, n, v, nthreads=16)
    g = np.load(os.path.join(ROOT, "tests", "golden", "route_fortran.npz"))
    psi_f = g["full_config4_psi"]
    scale = np.max(np.abs(psi_f))
    out.update(
        ranks_bit_identical=all(np.array_equal(x[0], gathered[0][0]) and np.array_equal(x[1], gathered[0][1]) and
                                np.array_equal(x[2], gathered[0][2]) for x in gathered),
        pools_total=sum(x[3] for x in gathered), all_trade_rows_bit_equal=all(x[4] for x in gathered),
        fixed_v_rel_err_vs_oracle=float(np.max(np.abs(psi_fixed - psi_o)) / np.max(np.abs(psi_o))),
        route_native_vs_fortran=float(np.max(np.abs(psi_native - psi_f)) / scale),
        route_scipy_vs_fortran=float(np.max(np.abs(psi_scipy - psi_f)) / scale),
        fortran_reorder_slack=float(g["full_config4_slack"]), evaluations=ev_native, fortran_evaluations=int(g["full_config4_evaluations"]))
    print("IPC_RANKS " + json.dumps(out), flush=True)
elif rank == 0:
    single = cr.Router(obj, market, n, device=0)
    cr.find_arb_(single, v)
    ref_fixed = cr.netflows(single).copy()
    cr.route_(single, v=np.ones(n), solver="native")
    ref_native = cr.netflows(single).copy()
    scale = np.max(np.abs(ref_native))
    out.update(
        ranks_bit_identical=all(np.array_equal(g[0], gathered[0][0]) and np.array_equal(g[1], gathered[0][1]) and
                                np.array_equal(g[2], gathered[0][2]) for g in gathered),
        pools_total=sum(g[3] for g in gathered),
        fixed_v_rel_diff=float(np.max(np.abs(psi_fixed - ref_fixed)) / np.max(np.abs(ref_fixed))),
        route_native_rel_diff=float(np.max(np.abs(psi_native - ref_native)) / scale),
        route_scipy_rel_diff=float(np.max(np.abs(psi_scipy - ref_native)) / scale),
        evaluations=ev_native)
    single.close()
    print("IPC_RANKS " + json.dumps(out), flush=True)
r.close()
dist.barrier()
dist.destroy_process_group()
