"""polish_ / dual_jacobian (cfmmrouter.jl_amd/router.py) on the CPU: the host logic driven through the test-only
OracleBackend.  What is shown here is the argument the GPU tests then make with the HIP path on one side:

  two evaluations of the SAME dual problem that differ only in rounding (here: the same pools in a different order,
  hence different summation order of Ψ and of the dual value) end route! (src/router.jl:58-108) up to ~1e-6·max|Ψ|
  apart -- both stop on factr = 1e1 with a stationarity residual left, each inside its own rounding noise of the dual
  VALUE -- and the projected chord-Newton polish, which works on the gradient Ψ alone, brings both to the same
  point: Ψ* equal to ≤ 1e-12·max|Ψ|.  So the distance at default tolerances is termination slack of L-BFGS-B,
  not a property of either evaluation.
"""
import numpy as np
import pytest

import cfmmrouter_amd as cr
from cfmmrouter_amd import synth
from cfmmrouter_amd.cfmms import PoolBatch
from helpers import OracleBackend, rel_to_max


def permuted(b, perm):
    if hasattr(b, "tick_off") and b.tick_off is not None and b.kind == 2:
        nt = np.diff(b.tick_off)
        off = np.concatenate([[0], np.cumsum(nt[perm])]).astype(np.int64)
        idx = np.concatenate([np.arange(b.tick_off[p], b.tick_off[p + 1]) for p in perm])
        return PoolBatch(b.kind, current_price=b.current_price[perm], tick_off=off, lower_ticks=b.lower_ticks[idx],
                         liquidity=b.liquidity[idx], γ=b.γ[perm], Ai=b.Ai[perm])
    kw = dict(R=b.R[perm], γ=b.γ[perm], Ai=b.Ai[perm])
    if getattr(b, "w", None) is not None:
        kw["w"] = b.w[perm]
    return PoolBatch(b.kind, **kw)


def market(kind, m, n, seed):
    if kind == "basket_bounded":      # config 5 in miniature: interior optimum
        return (synth.bounded_product_pools(m, n, seed=seed, consistent=True),
                cr.BasketLiquidation(1, synth.basket(n, seed=seed)), None)
    if kind == "basket_ticks":        # the multi-tick workload in miniature
        return (synth.univ3_ragged_pools(m, n, seed=seed), cr.BasketLiquidation(1, synth.basket(n, seed=seed)), None)
    if kind == "arb_geomean":
        return (synth.geomean_pools(m, n, seed=seed), cr.LinearNonnegative(synth.linear_prices(n, seed=seed)), np.ones(n))
    return (synth.product_pools(m, n, seed=seed), cr.LinearNonnegative(synth.linear_prices(n, seed=seed)), np.ones(n))


@pytest.mark.parametrize("kind,m,n", [("basket_bounded", 30_000, 64), ("basket_ticks", 8_000, 48), ("arb_product", 40_000, 96),
                                      ("arb_geomean", 20_000, 32)])
def test_polish_brings_two_roundings_of_one_problem_to_the_same_point(kind, m, n):
    b, obj, v0 = market(kind, m, n, seed=17)
    perm = np.random.default_rng(5).permutation(m)
    runs = []
    for bb in (b, permuted(b, perm)):
        r = cr.Router(obj, [bb], n, _backend=OracleBackend(n, [bb], nthreads=4))
        cr.route_(r, v=v0)
        default = cr.netflows(r).copy()
        cr.polish_(r)
        runs.append((default, cr.netflows(r).copy(), r.v.copy(), dict(r.info["polish"])))
    (d0, p0, v0s, i0), (d1, p1, v1s, i1) = runs
    scale = np.max(np.abs(p0))
    assert rel_to_max(d1, d0) <= 1e-5                       # default tolerances: sanity bound only
    assert rel_to_max(p1, p0) <= 1e-12                      # converged: the same point
    assert np.max(np.abs(v1s - v0s) / v0s) <= 1e-12
    for info in (i0, i1):
        assert info["residual"] <= 1e-12 * scale and info["residual"] <= info["residual0"]
        assert info["sweeps"] <= (n + 1) + 8 * 7 + 2
    # the polished point satisfies the dual's optimality conditions: G = ∇f + Ψ vanishes off the bounds, pushes outward on them
    lo = cr.lower_limit(obj)
    G = np.zeros(n)
    cr.grad_(G, obj, v0s)
    G += p0
    on = v0s <= lo
    assert np.max(np.abs(G[~on])) <= 1e-12 * scale and np.all(G[on] >= -1e-12 * scale)


def test_polish_with_a_foreign_chord_matrix_reaches_the_same_point():
    """The limit point belongs to the backend whose gradient is driven to zero, not to the matrix: polishing run B with
    the Jacobian taken at run A's end point gives B's own fixed point (what bench.py / the GPU tests do with the
    device's Jacobian on the CPU restatement's side)."""
    n, m = 48, 20_000
    b, obj, _ = market("basket_bounded", m, n, seed=3)
    ra = cr.Router(obj, [b], n, _backend=OracleBackend(n, [b]))
    cr.route_(ra)
    J = cr.dual_jacobian(ra)
    assert J.shape == (n, n) and np.max(np.abs(J - J.T)) <= 1e-3 * np.max(np.abs(J))   # a Hessian: symmetric up to FD error
    # homogeneity of degree 0 in the prices: J·ν = 0 (only price RATIOS matter to a pool)
    assert np.max(np.abs(J @ ra.v)) <= 1e-4 * np.max(np.abs(J)) * np.max(ra.v)
    cr.polish_(ra, jacobian=J)
    bb = permuted(b, np.random.default_rng(1).permutation(m))
    rb = cr.Router(obj, [bb], n, _backend=OracleBackend(n, [bb]))
    cr.route_(rb, v=np.ones(n) / n * (1 + 1e-3 * np.arange(n)))     # a different trajectory altogether
    sw = rb.n_sweeps
    cr.polish_(rb, jacobian=J)
    assert rb.n_sweeps - sw <= 8 * 7 + 2                             # no Jacobian sweeps of its own
    assert rel_to_max(cr.netflows(rb), cr.netflows(ra)) <= 1e-12


def test_polish_on_a_converged_corner_solution_is_a_no_op():
    """README.md:27-38: both prices end where L-BFGS-B put them; polish_ must not move a converged point by more than
    rounding, and r.Δs / r.Λs describe the point it leaves."""
    pools = [cr.ProductTwoCoin([1e6, 1e6], 1, [1, 2]), cr.ProductTwoCoin([1e3, 2e3], 1, [1, 2])]
    b = cr.ProductTwoCoin.batch([[1e6, 1e6], [1e3, 2e3]], [1.0, 1.0], [[1, 2], [1, 2]])
    r = cr.Router(cr.LinearNonnegative(np.ones(2)), pools, 2, _backend=OracleBackend(2, [b]))
    cr.route_(r)
    before = cr.netflows(r).copy()
    cr.polish_(r)
    after = cr.netflows(r)
    assert abs(after[1] - 171.4) < 0.1 and rel_to_max(after, before) <= 1e-6
    assert r.info["polish"]["residual"] <= max(r.info["polish"]["residual0"], 1e-9)
    D, L = r.Δs, r.Λs
    np.testing.assert_allclose((L - D).sum(axis=0)[[0, 1]], after, rtol=0, atol=1e-9)
