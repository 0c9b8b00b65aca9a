"""The library's own L-BFGS-B (csrc/lbfgsb.cpp) against SciPy's translation of the Fortran L-BFGS-B 3.0
(the solver family the reference uses through LBFGSB.jl).  CPU only: the solver is host code."""
import numpy as np
import pytest
from scipy.optimize import fmin_l_bfgs_b

import cfmmrouter_amd as cr
from cfmmrouter_amd import synth
from cfmmrouter_amd._lib import lbfgsb_minimize
from oracle import cfmm_oracle as orc
from helpers import oracle_objective, oracle_poolset, rel_to_max


def rosen(x):
    f = np.sum(100.0 * (x[1:] - x[:-1] ** 2) ** 2 + (1 - x[:-1]) ** 2)
    g = np.zeros_like(x)
    g[:-1] = -400 * x[:-1] * (x[1:] - x[:-1] ** 2) - 2 * (1 - x[:-1])
    g[1:] += 200 * (x[1:] - x[:-1] ** 2)
    return f, g


@pytest.mark.parametrize("n", [2, 10, 50])
def test_rosenbrock_unbounded(n):
    x0 = np.full(n, -1.2)
    x, info = lbfgsb_minimize(rosen, x0, [(None, None)] * n, m=10, factr=1e1, pgtol=1e-8)
    assert info["status"] in (0, 1)
    np.testing.assert_allclose(x, np.ones(n), atol=1e-5)


@pytest.mark.parametrize("n", [2, 10, 50])
def test_rosenbrock_box_active(n):
    bounds = [(-1.5, 0.5)] * n   # the optimum (1,...,1) is outside: constraints active
    x0 = np.full(n, -1.0)
    x, info = lbfgsb_minimize(rosen, x0, bounds, m=5, factr=1e1, pgtol=1e-8)
    xs, fs, d = fmin_l_bfgs_b(rosen, x0, bounds=bounds, m=5, factr=1e1, pgtol=1e-8)
    assert abs(info["f"] - fs) <= 1e-8 * max(1.0, abs(fs))
    np.testing.assert_allclose(x, xs, atol=1e-5)
    assert np.all(x >= -1.5) and np.all(x <= 0.5)


def test_quadratic_mixed_bounds():
    rng = np.random.default_rng(0)
    n = 40
    A = rng.standard_normal((n, n))
    Q = A @ A.T + 0.1 * np.eye(n)
    b = rng.standard_normal(n) * 5
    fun = lambda x: (0.5 * x @ Q @ x - b @ x, Q @ x - b)
    bounds = [(0.0, None) if i % 3 == 0 else ((None, 0.3) if i % 3 == 1 else (-0.2, 0.2)) for i in range(n)]
    x0 = np.zeros(n)
    x, info = lbfgsb_minimize(fun, x0, bounds, m=5, factr=1e1, pgtol=1e-9)
    xs, fs, d = fmin_l_bfgs_b(fun, x0, bounds=bounds, m=5, factr=1e1, pgtol=1e-9)
    assert abs(info["f"] - fs) <= 1e-9 * max(1.0, abs(fs))
    np.testing.assert_allclose(x, xs, atol=1e-6)


def test_starting_point_is_projected_and_pgtol_at_start():
    fun = lambda x: (float(np.sum((x - 3.0) ** 2)), 2 * (x - 3.0))
    x, info = lbfgsb_minimize(fun, np.array([10.0, -10.0]), [(0.0, 1.0), (0.0, 1.0)])
    np.testing.assert_array_equal(x, [1.0, 1.0])
    x, info = lbfgsb_minimize(fun, np.array([1.0, 1.0]), [(0.0, 1.0), (0.0, 1.0)])
    assert info["status"] == 0 and info["evaluations"] == 1 and info["iterations"] == 0


@pytest.mark.parametrize("m,n,kind", [(2, 2, "readme"), (100, 10, "arb"), (3000, 32, "arb"), (100, 10, "basket"),
                                      (20000, 64, "arb")])
def test_dual_problem_matches_scipy(m, n, kind):
    """The actual problem route! solves (src/router.jl:58-108), objective evaluated by the oracle:
    own solver vs SciPy -> same netflows within north_star's 1e-6 of max|Ψ|."""
    if kind == "readme":
        b = cr.ProductTwoCoin.batch([[1e6, 1e6], [1e3, 2e3]], [1.0, 1.0], [[1, 2], [1, 2]])
        obj, v0 = cr.LinearNonnegative(np.ones(2)), np.ones(2) / 2
    else:
        b = synth.product_pools(m, n, seed=m)
        if kind == "arb":
            obj, v0 = cr.LinearNonnegative(synth.linear_prices(n, seed=m)), np.ones(n)
        else:
            obj, v0 = cr.BasketLiquidation(1, synth.basket(n, seed=m)), np.ones(n) / n
    ps, oo = oracle_poolset([b], n), oracle_objective(obj)
    ref = orc.route_oracle(oo, ps, v0=v0)

    def fg(v):
        D, L = ps.sweep(v)
        G = oo.grad(v)
        orc.grad_scatter(G, D, L, ps.Ai)
        return oo.f(v) + orc.dual_acc(D, L, ps.Ai, v), G

    lo = oo.lower_limit()
    # the reference's own call shape (nbd = 2 everywhere, u = Inf -> the Fortran's "boxed" first step)
    vb, infob = lbfgsb_minimize(fg, v0, [(lo[j], None) for j in range(n)], reference_boxed=True)
    Db, Lb = ps.sweep(vb)
    assert rel_to_max(orc.netflows(Db, Lb, ps.Ai, n), ref["psi"]) <= 1e-6
    assert abs(infob["f"] - ref["f"]) <= 1e-9 * max(1.0, abs(ref["f"]))
    v, info = lbfgsb_minimize(fg, v0, [(lo[j], None) for j in range(n)])
    assert info["status"] in (0, 1), info
    D, L = ps.sweep(v)
    psi = orc.netflows(D, L, ps.Ai, n)
    assert rel_to_max(psi, ref["psi"]) <= 1e-6
    assert abs(info["f"] - ref["f"]) <= 1e-9 * max(1.0, abs(ref["f"]))
    # not wildly more expensive than the Fortran lineage
    assert info["evaluations"] <= 3 * ref["info"]["funcalls"] + 10


def _config5_problem(m, n, seed, threads=4):
    """The dual of BASELINE config 5 in miniature (BasketLiquidation over BoundedProduct pools quoted around one price
    vector: interior optimum), evaluated by the CPU restatement."""
    b = synth.bounded_product_pools(m, n, seed=seed, consistent=True)
    obj = cr.BasketLiquidation(1, synth.basket(n, seed=seed))
    ps, oo = oracle_poolset([b], n), oracle_objective(obj)

    def fg(v):
        v = np.array(v, dtype=np.float64)
        D, L = ps.sweep(v, threads)
        G = oo.grad(v)
        orc.grad_scatter(G, D, L, ps.Ai)
        return oo.f(v) + orc.dual_acc(D, L, ps.Ai, v), G

    return fg, oo.lower_limit(), np.ones(n) / n


def _both_solvers(fg, lo, v0):
    n, logs = v0.size, {"scipy": [], "native": []}

    def logged(key):
        def fun(x):
            logs[key].append(np.array(x, dtype=np.float64))
            return fg(x)
        return fun

    fmin_l_bfgs_b(logged("scipy"), v0.copy(), bounds=[(lo[j], orc.BOXED_INF) for j in range(n)], m=5, factr=1e1, pgtol=1e-5)
    lbfgsb_minimize(logged("native"), v0.copy(), [(lo[j], None) for j in range(n)], m=5, factr=1e1, pgtol=1e-5,
                    reference_boxed=True)
    return logs["scipy"], logs["native"]


def test_interior_optimum_trajectory_is_scipys_until_rounding_takes_over():
    """where do csrc/lbfgsb.cpp and SciPy's L-BFGS-B (the translation of the Fortran 3.0 code the
    reference calls, src/router.jl:60,105) part ways on an interior optimum?  Same callback, same start, same call
    shape: NOWHERE as far as decisions go -- every evaluation point of the two solvers agrees to rounding
    (≤ 1e-12 relative over the first 20 evaluations: Cauchy point, subspace step, initial step 1, every Moré-Thuente
    trial), the difference then grows smoothly with the conditioning of the problem (≤ 1e-6 at every common
    evaluation, no jump that a different branch would cause), and the runs end within a few evaluations ... or tens of
    evaluations of each other, depending on which side of factr·eps the LAST decrease of the dual value falls (see the
    next test).  scripts/solver_trace.py prints the full trace at any size."""
    fg, lo, v0 = _config5_problem(60_000, 128, seed=1234)
    a, b = _both_solvers(fg, lo, v0)
    k = min(len(a), len(b))
    assert k >= 40
    dx = np.array([np.max(np.abs(a[i] - b[i]) / np.abs(a[i])) for i in range(k)])
    assert np.all(dx[:20] <= 1e-12), dx[:20]
    assert np.all(dx <= 1e-5), dx.max()
    assert np.all(dx[-10:] <= 1e-6)          # still the same trajectory at the end of the shorter run


def test_interior_optimum_evaluation_counts_match_scipys_on_average():
    """On one market the two solvers' evaluation counts differ by up to ±30 % EITHER way (163 vs 123 on the device's
    config 5 in round 3; 126 vs 148 on seed 1 here): the run ends when a decrease of the dual value falls below
    factr·eps·|f| ≈ 1.6e-11, which is the rounding noise of a sum over the pools, and a side that misses that test by
    a last-place digit spends up to 20 more evaluations per line search among values that differ in the last bits
    (the reference's solver does the same).  There is no systematic difference: over a set of markets the mean counts
    agree within a few per cent."""
    counts = []
    for seed in range(1, 9):
        fg, lo, v0 = _config5_problem(30_000, 96, seed=seed)
        a, b = _both_solvers(fg, lo, v0)
        counts.append((len(a), len(b)))
    counts = np.array(counts, dtype=np.float64)
    ratio = counts[:, 1].mean() / counts[:, 0].mean()
    print("evaluations scipy / native per market:", counts.astype(int).tolist(), "ratio of means", ratio)
    assert 0.88 <= ratio <= 1.12
