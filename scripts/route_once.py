"""One warm cfmm_route on a bench workload (for rocprofv3 --kernel-trace, see route_kernel_gaps.py).
usage: python scripts/route_once.py WORKLOAD ARMED"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cfmmrouter_amd as cr
import bench
name, armed = sys.argv[1], int(sys.argv[2])
desc, n, _ = bench.WORKLOADS[name]
batches = bench.build_market(name, 0, 1, "weak")
obj = bench.objective_for(name, n)
v0 = np.ones(n) if isinstance(obj, cr.LinearNonnegative) else None
r = cr.Router(obj, batches, n)
r._backend.ctx.set_option("armed", armed)
for _ in range(3):
    cr.route_(r, v=v0, solver="native")
print(name, "armed", armed, "evaluations", r.info["funcalls"], "route ms", 1e3 * r.info["total_seconds"])
r.close()
