"""Pins the OUTER LOOP of route! to the solver lineage the reference actually calls.

    /opt/conda/bin/python3.9 tests/golden/make_route_golden.py        # -> tests/golden/route_fortran.npz
    /opt/conda/bin/python3.9 tests/golden/make_route_golden.py --only full_config4 --merge   # add / refresh one market

The reference's route! (src/router.jl:58-108) hands its dual problem to LBFGSB.jl (src/router.jl:60,105), a `ccall`
wrapper of the FORTRAN L-BFGS-B 3.0 `setulb` in a reverse-communication loop, with the call shape

    nbd = 2 for EVERY variable, l = lower_limit(objective), u = Inf      (src/router.jl:67-70)
    m = 5, factr = 1e1, pgtol = 1e-5, maxfun = maxiter = 15 000           (src/router.jl:58)
    x0 = ones(n)/n unless v is given                                      (src/router.jl:61-65)

The system SciPy (1.15) ships a C TRANSLATION of that code; SciPy 1.7.1 under /opt/conda (Python 3.9) is the last
line that F2PY-wraps the Fortran 3.0 source itself (scipy/optimize/lbfgsb_src/lbfgsb.f, "3.0 (released April 25,
2011)"; SciPy's only interface change is the `maxls` argument in place of the hard-coded 20 backtracking steps, passed as
20 here).  This script drives `scipy.optimize._lbfgsb.setulb` DIRECTLY -- not through fmin_l_bfgs_b, which would map the
infinite upper bound to nbd = 1 and lose the Fortran's "boxed" unit first step -- exactly like LBFGSB.jl's loop does
(task = START; call setulb; on FG evaluate f then g; on NEW_X check the iteration / evaluation caps; anything else ends),
with fn / g! restated on the CPU oracle (oracle/cfmm_oracle.c: serial pool-order sums, src/router.jl:79-83, :98-100) and
the reference's v-cache rule (:74-77, :92-95).  Recorded per market: every evaluation point x_k (the first KEEP of them,
as float64 bits), the dual value at each, the task string at the end, v*, Ψ* (netflows at v*), evaluation and
iteration counts.

These ARE outputs of the reference's solver code on a faithful restatement of its callbacks -- the closest thing to
reference output this image can produce (no Julia).  tests/test_route_fortran_pin.py holds csrc/lbfgsb.cpp and the
device route! against them.

Markets (pure functions of (seed, pool index): cfmmrouter.jl_amd/synth.py):
    readme          README.md:27-38                                          2 pools,  2 tokens
    config2_mini    config 2 in miniature: ProductTwoCoin, LinearNonnegative 20 000 pools, 64 tokens
    config3_mini    config 3 in miniature: Product + GeometricMean           20 000 + 20 000 pools, 128 tokens
    config4_mini    config 4's shard in miniature: ProductTwoCoin            50 000 pools, 512 tokens
    config5_300k    config 5: BoundedProduct, BasketLiquidation              300 000 pools, 256 tokens (interior optimum)
    univ3_mini      ragged multi-tick UniV3 ladders, BasketLiquidation       30 000 pools, 128 tokens
"""
import os
import sys
import warnings

warnings.filterwarnings("ignore")
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import scipy
from scipy.optimize import _lbfgsb

import cfmmrouter_amd as cr
from cfmmrouter_amd import synth
from oracle import cfmm_oracle as orc
from helpers import oracle_objective, oracle_poolset

KEEP = 48          # evaluation points kept per miniature market
KEEP_FULL = 24     # ... per BASELINE-size market (the tests compare the first 20)
PERMS = 3          # re-runs of the SAME market with its pools in another order (the reference sums in pool order)


def markets(only=()):
    for item in all_markets(only):
        if not only or item[0] in only:
            yield item


def all_markets(only=()):
    b = cr.ProductTwoCoin.batch([[1e6, 1e6], [1e3, 2e3]], [1.0, 1.0], [[1, 2], [1, 2]])
    yield "readme", [b], 2, cr.LinearNonnegative(np.ones(2)), None
    n = 64
    yield "config2_mini", [synth.product_pools(20_000, n, seed=1234)], n, cr.LinearNonnegative(synth.linear_prices(n, seed=1234)), np.ones(n)
    n = 128
    yield ("config3_mini", [synth.product_pools(20_000, n, seed=1234), synth.geomean_pools(20_000, n, seed=1234)], n,
           cr.LinearNonnegative(synth.linear_prices(n, seed=1234)), np.ones(n))
    n = 512
    yield "config4_mini", [synth.product_pools(50_000, n, seed=1234)], n, cr.LinearNonnegative(synth.linear_prices(n, seed=1234)), np.ones(n)
    n = 256
    yield ("config5_300k", [synth.bounded_product_pools(300_000, n, seed=1234, consistent=True)], n,
           cr.BasketLiquidation(1, synth.basket(n, seed=1234)), None)
    n = 128
    yield ("univ3_mini", [synth.univ3_ragged_pools(30_000, n, seed=1234)], n,
           cr.BasketLiquidation(1, synth.basket(n, seed=1234)), None)
    # the bench's workloads at BASELINE size (benchlib/workloads.py: the markets bench.py and the GPU tests build)
    from benchlib.workloads import WORKLOADS, build_market, objective_for
    # "config4" = BASELINE config 4 at its stated size: ALL 4M ProductTwoCoin pools, 512 tokens (round 6; the pool order is
    # the global one: shard after shard of the 8-GPU split)
    for name in ("config2", "config3", "config4shard", "config5", "product1m", "univ3_ticks", "config4"):
        if only and "full_" + name not in only:
            continue
        n = WORKLOADS[name][1]
        obj = objective_for(name, n)
        yield "full_" + name, build_market(name, 0, 1, "weak"), n, obj, (np.ones(n) if isinstance(obj, cr.LinearNonnegative) else None)


def permuted(batch, idx):
    """The same pools in another order (a router built from a differently ordered cfmms vector)."""
    from cfmmrouter_amd._lib import KIND_GEOMEAN, KIND_PRODUCT
    from cfmmrouter_amd.cfmms import PoolBatch
    if batch.kind == KIND_PRODUCT:
        return PoolBatch(batch.kind, R=batch.R[idx], γ=batch.γ[idx], Ai=batch.Ai[idx])
    if batch.kind == KIND_GEOMEAN:
        return PoolBatch(batch.kind, R=batch.R[idx], w=batch.w[idx], γ=batch.γ[idx], Ai=batch.Ai[idx])
    lens = np.diff(batch.tick_off)[idx]
    off = np.zeros(len(idx) + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    src = np.repeat(batch.tick_off[:-1][idx] - off[:-1], lens) + np.arange(off[-1])
    return PoolBatch(batch.kind, current_price=batch.current_price[idx], tick_off=off, lower_ticks=batch.lower_ticks[src],
                     liquidity=batch.liquidity[src], γ=batch.γ[idx], Ai=batch.Ai[idx])


def route_fortran(obj, batches, n, v0, m=5, factr=1e1, pgtol=1e-5, maxfun=15_000, maxiter=15_000, threads=8):
    """route! with the Fortran setulb in LBFGSB.jl's reverse-communication loop."""
    ps, oo = oracle_poolset(batches, n), oracle_objective(obj)
    state = {"v": (np.ones(n) / n if v0 is None else np.array(v0, dtype=np.float64)), "sweeps": 0}

    def sweep(v):
        state["D"], state["L"] = ps.sweep(v, threads)
        state["sweeps"] += 1

    def fn(v):                                                   # src/router.jl:73-86
        if not np.all(v == state["v"]):
            sweep(v)
            state["v"] = v.copy()
        return oo.f(v) + orc.dual_acc(state["D"], state["L"], ps.Ai, v)

    def g(v):                                                    # src/router.jl:89-102
        if not np.all(v == state["v"]):
            sweep(v)
            state["v"] = v.copy()
        G = oo.grad(v)
        orc.grad_scatter(G, state["D"], state["L"], ps.Ai)
        return G

    sweep(state["v"])                                            # :104
    x = state["v"].copy()
    low = np.array(oo.lower_limit(), dtype=np.float64)           # :69
    upp = np.full(n, np.inf)                                     # :70  (upper_limit = Inf)
    nbd = np.full(n, 2, dtype=np.int32)                          # :68  bounds[1, :] .= 2
    f = np.array(0.0, np.float64)
    grad = np.zeros(n, np.float64)
    wa = np.zeros(2 * m * n + 5 * n + 11 * m * m + 8 * m, np.float64)
    iwa = np.zeros(3 * n, np.int32)
    task = np.zeros(1, "S60")
    csave = np.zeros(1, "S60")
    lsave = np.zeros(4, np.int32)
    isave = np.zeros(44, np.int32)
    dsave = np.zeros(29, np.float64)
    task[:] = "START"
    xs, fs = [], []
    evals = 0
    while True:
        _lbfgsb.setulb(m, x, low, upp, nbd, f, grad, factr, pgtol, wa, iwa, task, -1, csave, lsave, isave, dsave, 20)
        t = task.tobytes()
        if t.startswith(b"FG"):
            xs.append(x.copy())
            f[...] = fn(x)
            grad[:] = g(x)
            fs.append(float(f))
            evals += 1
        elif t.startswith(b"NEW_X"):
            if isave[29] >= maxiter:
                task[:] = "STOP: TOTAL NO. of ITERATIONS REACHED LIMIT"
            elif isave[33] >= maxfun:
                task[:] = "STOP: TOTAL NO. of f AND g EVALUATIONS EXCEEDS LIMIT"
        else:
            break
    v = x.copy()
    state["v"] = v                                               # :106
    sweep(v)                                                     # :107
    psi = orc.netflows(state["D"], state["L"], ps.Ai, n)
    return {"v": v, "psi": psi, "f": float(f), "evaluations": evals, "iterations": int(isave[29]),
            "task": task.tobytes().rstrip(b"\x00 ").decode(), "xs": np.array(xs), "fs": np.array(fs),
            "x_last": np.array(xs[-1]), "pools": sum(len(b) for b in batches)}


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="", help="comma-separated market names (default: all)")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "route_fortran.npz"))
    ap.add_argument("--merge", action="store_true", help="keep the markets already in --out that are not regenerated (--only)")
    args = ap.parse_args()
    only = set(filter(None, args.only.split(",")))
    out = {"scipy_version": np.array(scipy.__version__), "keep": np.array(KEEP)}
    if args.merge and os.path.exists(args.out):
        old = np.load(args.out)
        assert str(old["scipy_version"]) == scipy.__version__, "merge into a fixture of another SciPy build"
        out.update({k: old[k] for k in old.files})
    for name, batches, n, obj, v0 in markets(only):
        r = route_fortran(obj, batches, n, v0)
        r["xs"] = r["xs"][:KEEP_FULL if name.startswith("full_") else KEEP]
        # The reference's own reproducibility: the SAME market with its pools in another order (fn / g! sum in pool order,
        # src/router.jl:81-83, :98-100, so only the rounding of those sums changes) through the SAME Fortran solver.
        # `slack` = the largest distance max|dPsi| / max|Psi| between such runs and the run above: what "the reference's
        # netflows" are defined to, at the reference's tolerances, on this market.
        scale = np.max(np.abs(r["psi"]))
        rng = np.random.default_rng(20260926)
        perm_psi, perm_evals = [], []
        for k in range(PERMS if r["pools"] > 2 else 0):
            pb = [permuted(b, (np.arange(len(b))[::-1] if k == 0 else rng.permutation(len(b)))) for b in batches]
            rp = route_fortran(obj, pb, n, v0)
            perm_psi.append(rp["psi"])
            perm_evals.append(rp["evaluations"])
        slack = max([float(np.max(np.abs(p - r["psi"])) / scale) for p in perm_psi], default=0.0)
        print(f"{name:18s} pools {r['pools']:7d} tokens {n:4d} evaluations {r['evaluations']:4d} {perm_evals} iterations {r['iterations']:4d} "
              f"f* {r['f']:.15e} max|psi| {scale:.6e} slack {slack:.2e}  {r['task']}", flush=True)
        for k in ("v", "psi", "xs", "fs", "x_last"):
            out[f"{name}_{k}"] = r[k]
        out[f"{name}_perm_psi"] = np.array(perm_psi)
        out[f"{name}_perm_evaluations"] = np.array(perm_evals)
        out[f"{name}_slack"] = np.array(slack)
        out[f"{name}_f"] = np.array(r["f"])
        out[f"{name}_evaluations"] = np.array(r["evaluations"])
        out[f"{name}_iterations"] = np.array(r["iterations"])
        out[f"{name}_task"] = np.array(r["task"])
    np.savez_compressed(args.out, **out)


if __name__ == "__main__":
    main()
