"""Prints the dual value and projected-gradient norm of every L-BFGS-B evaluation of route! on a workload
(GPU sweeps, library solver driven from Python) -- to see where the factr = 1e1 stopping test ends the run."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cfmmrouter_amd as cr
from cfmmrouter_amd import objectives as ob, synth
from cfmmrouter_amd._lib import lbfgsb_minimize
import bench
name = sys.argv[1] if len(sys.argv) > 1 else "config3"
desc, n, build = bench.WORKLOADS[name]
batches = build(0)
obj = bench.objective_for(name, n)
for opts in ({}, {"xcd_map": 0}):
    be = cr.DeviceBackend(n, batches)
    for k, v in opts.items(): be.ctx.set_option(k, v)
    lo = ob.lower_limit(obj); hist = []
    def fg(x):
        psi, acc = be.eval(x)
        G = np.zeros(n); ob.grad_(G, obj, x); G += psi
        f = ob.f(obj, x) + acc
        pg = np.max(np.abs(np.where(G > 0, np.minimum(x - lo, G), G)))
        hist.append((f, pg)); return f, G
    v0 = np.ones(n) if isinstance(obj, cr.LinearNonnegative) else np.ones(n) / n
    x, info = lbfgsb_minimize(fg, v0, [(lo[j], None) for j in range(n)], reference_boxed=True)
    print(name, opts, "evaluations", info["evaluations"], "iterations", info["iterations"], "status", info["status"])
    f_end = hist[-1][0]
    for k, (f, pg) in enumerate(hist):
        print(f"  eval {k:3d}  f - f_end = {f - f_end:+.3e}  rel {abs(f - f_end) / abs(f_end):.1e}  |proj g| = {pg:.3e}")
    be.close()
