"""A/B of one library option on cfmm_route, same context, interleaved rounds: python scripts/route_ab.py WORKLOAD OPTION [ROUNDS]
Prints per setting the minimum and the median of the library's own route timer over ROUNDS x 40 warm routes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cfmmrouter_amd as cr
import bench
name, opt = sys.argv[1], sys.argv[2]
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 5
desc, n, _ = bench.WORKLOADS[name]
batches = bench.build_market(name, 0, 1, "weak")
obj = bench.objective_for(name, n)
v0 = np.ones(n) if isinstance(obj, cr.LinearNonnegative) else None
r = cr.Router(obj, batches, n)
ctx = r._backend.ctx
t = {0: [], 1: []}
ev = {}
for rnd in range(rounds):
    for val in (1, 0):
        ctx.set_option(opt, val)
        for k in range(45):
            cr.route_(r, v=v0, solver="native")
            if k >= 5:
                t[val].append(1e3 * r.info["total_seconds"])
        ev[val] = r.info["funcalls"]
for val in (1, 0):
    a = np.array(t[val])
    print(f"{name} {opt}={val}: evaluations {ev[val]}  route ms min {a.min():.4f} median {np.median(a):.4f} p90 {np.percentile(a, 90):.4f}")
r.close()
