"""The reference's own published benchmark, reproduced: benchmark/scaling.jl:8-38 times route! on
m = round(10^x), x in 10 points of [2, 4], ProductTwoCoin pools, R = 1000·rand(2), γ ∈ {0.997, 1}, n_tokens = {1, 2, 4}·√m,
LinearNonnegative(rand(n)), v0 = ones(n).  These are the ONLY numbers the reference publishes (a plot, README.md:48-53,
read off by eye in BASELINE.md §1, MacBook Pro 2.3 GHz 8-core i9).  Julia's RNG stream cannot be reproduced here, so the
pools come from this repo's generator with the same distributions (synth.product_pools, seed 1234 like :13)."""
import time

import numpy as np

import cfmmrouter_amd as cr
from cfmmrouter_amd import synth

# values read off benchmark/router_scaling.png (±15 %), seconds
REFERENCE_PLOT_S = {(100, 1): 2.1e-3, (100, 2): 3.7e-3, (100, 4): 6.8e-3, (10000, 1): 0.19, (10000, 2): 0.48, (10000, 4): 0.20}


def grid_points():
    ms = [int(round(10 ** x)) for x in np.linspace(2, 4, 10)]       # scaling.jl:8
    return [(m, f, int(round(f * np.sqrt(m)))) for m in ms for f in (1, 2, 4)]   # :9, :15


def median_ms(fn, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts))


def run_grid(device=0, cpu_route=None, reps=7):
    """route! on every grid point through the GPU path: the library's one-call route! (cfmm_route) and the SciPy-driven
    loop; `cpu_route(obj, market, n, v0, psi_device)` (bench.py: the CPU restatement -- test infrastructure) adds its
    wall-clock and the netflow distance per point."""
    rows = []
    for m, f, n in grid_points():
        market = [synth.product_pools(m, n, seed=1234)]
        obj, v0 = cr.LinearNonnegative(synth.linear_prices(n, seed=1234)), np.ones(n)
        r = cr.Router(obj, market, n, device=device)
        try:
            cr.route_(r, v=v0, solver="native")          # warm
            row = {"m": m, "factor": f, "n_tokens": n,
                   "native_ms": median_ms(lambda: cr.route_(r, v=v0, solver="native"), reps),
                   "native_evaluations": int(r.info["funcalls"])}
            psi = cr.netflows(r).copy()
            row["scipy_driven_ms"] = median_ms(lambda: cr.route_(r, v=v0, solver="scipy"), max(3, reps // 2))
            row["scipy_driven_evaluations"] = int(r.info["funcalls"])
            if (m, f) in REFERENCE_PLOT_S:
                row["reference_plot_ms"] = 1e3 * REFERENCE_PLOT_S[(m, f)]
            if cpu_route is not None:
                row.update(cpu_route(obj, market, n, v0, psi))
            rows.append(row)
        finally:
            r.close()
    return rows
