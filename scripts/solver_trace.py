"""Where do csrc/lbfgsb.cpp and SciPy's L-BFGS-B (C translation of the Fortran 3.0 the reference calls through
LBFGSB.jl, src/router.jl:60,105) part ways?  Same callback (the CPU restatement -- test infrastructure), same
starting point, same call shape (nbd = 2 everywhere, upper bound unreachable); every evaluation point of both
solvers is logged and the first evaluation whose x differs is reported.

    python scripts/solver_trace.py [--pools 300000] [--tokens 256] [--workload config5|arb] [--threads 8]
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cfmmrouter_amd as cr  # noqa: E402
from cfmmrouter_amd import synth  # noqa: E402
from cfmmrouter_amd._lib import lbfgsb_minimize  # noqa: E402
from helpers import oracle_objective, oracle_poolset  # noqa: E402
from oracle import cfmm_oracle as orc  # noqa: E402


def make_problem(workload, m, n, threads):
    if workload == "config5":
        b = synth.bounded_product_pools(m, n, seed=1234, consistent=True)
        obj, v0 = cr.BasketLiquidation(1, synth.basket(n, seed=1234)), np.ones(n) / n
    elif workload == "univ3_ticks":
        b = synth.univ3_ragged_pools(m, n, seed=1234)
        obj, v0 = cr.BasketLiquidation(1, synth.basket(n, seed=1234)), np.ones(n) / n
    else:
        b = synth.product_pools(m, n, seed=1234)
        obj, v0 = cr.LinearNonnegative(synth.linear_prices(n, seed=1234)), np.ones(n)
    ps, oo = oracle_poolset([b], n), oracle_objective(obj)

    def fg(v):
        D, L = ps.sweep(v, threads)
        acc = orc.dual_acc(D, L, ps.Ai, v)
        G = oo.grad(v)
        orc.grad_scatter(G, D, L, ps.Ai)
        return oo.f(v) + acc, G

    return fg, oo.lower_limit(), v0


def trace(workload="config5", m=300_000, n=256, threads=8, factr=1e1, pgtol=1e-5, verbose=True):
    from scipy.optimize import fmin_l_bfgs_b

    fg, lo, v0 = make_problem(workload, m, n, threads)
    logs = {"scipy": [], "native": []}

    def logged(key):
        def fun(x):
            f, g = fg(np.array(x, dtype=np.float64))
            logs[key].append((np.array(x, dtype=np.float64), float(f)))
            return f, g
        return fun

    bounds = [(lo[j], orc.BOXED_INF) for j in range(n)]
    xs, fs, info = fmin_l_bfgs_b(logged("scipy"), v0.copy(), bounds=bounds, m=5, factr=factr, pgtol=pgtol, iprint=-1)
    xn, infon = lbfgsb_minimize(logged("native"), v0.copy(), [(lo[j], None) for j in range(n)], m=5, factr=factr,
                                pgtol=pgtol, reference_boxed=True)
    a, b = logs["scipy"], logs["native"]
    first = None
    for k in range(min(len(a), len(b))):
        dx = np.max(np.abs(a[k][0] - b[k][0]) / np.maximum(np.abs(a[k][0]), 1e-300))
        if verbose:
            print(f"eval {k:3d}  f_scipy {a[k][1]:.15e}  f_native {b[k][1]:.15e}  max rel dx {dx:.2e}")
        if first is None and dx > 1e-10:
            first = k
    out = {"scipy_evals": len(a), "native_evals": len(b), "scipy_iters": info["nit"], "native_iters": infon["iterations"],
           "first_divergent_eval": first, "f_scipy": fs, "f_native": infon["f"],
           "x_rel_diff": float(np.max(np.abs(xs - xn) / np.abs(xs)))}
    if verbose:
        print(out)
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--pools", type=int, default=300_000)
    ap.add_argument("--tokens", type=int, default=256)
    ap.add_argument("--workload", default="config5")
    ap.add_argument("--threads", type=int, default=8)
    ap.add_argument("--factr", type=float, default=1e1)
    ap.add_argument("--pgtol", type=float, default=1e-5)
    a = ap.parse_args()
    trace(a.workload, a.pools, a.tokens, a.threads, a.factr, a.pgtol)
