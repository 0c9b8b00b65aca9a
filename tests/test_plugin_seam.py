"""The reference's plugin seam: a Router takes ANY `CFMM{T}` subtype that defines `find_arb!(Δ, Λ, cfmm, v)`
(src/cfmms.jl:35,56 -- dispatch on the pool's type; src/router.jl:40 calls it per pool).  Here: a pool whose type has no
device kernel is evaluated by its own `find_arb_(Δ, Λ, v)` on the host at every evaluation and its (Λ − Δ) / dual term is
added to the device's Ψ / acc before L-BFGS-B sees them (router.py: HostSegment, MixedBackend).

CPU: the seam over the test-only OracleBackend.  GPU: over the HIP sweep."""
import math

import numpy as np
import pytest

import cfmmrouter_amd as cr
from cfmmrouter_amd import synth
from cfmmrouter_amd._lib import KIND_PRODUCT
from helpers import OracleBackend, oracle_objective, oracle_poolset, rel_to_max
from oracle import cfmm_oracle as orc


class MyProduct(cr.CFMM):
    """A user's own two-coin pool type: the constant-product closed form (src/cfmms.jl:125-140) written by the user, in
    Python, with the reference's operation order -- so the CPU oracle's ProductTwoCoin is its exact check."""

    def __init__(self, R, γ, Ai):
        self.R, self.γ, self.Ai = np.array(R, dtype=np.float64), float(γ), np.array(Ai, dtype=np.int64)

    def find_arb_(self, Δ, Λ, v):
        R, γ = self.R, self.γ
        k = R[0] * R[1]
        arb_δ = lambda m, r: max(math.sqrt(γ * m * k) - r, 0.0) / γ
        arb_λ = lambda m, r: max(r - math.sqrt(k / (m * γ)), 0.0)
        Δ[0], Δ[1] = arb_δ(v[1] / v[0], R[0]), arb_δ(v[0] / v[1], R[1])
        Λ[0], Λ[1] = arb_λ(v[0] / v[1], R[0]), arb_λ(v[1] / v[0], R[1])

    def update_reserves_(self, Δ, Λ, v):
        self.R[:] = self.R + self.γ * Δ - Λ


class ConstantSum3(cr.CFMM):
    """A three-coin pool no kernel knows: φ(R) = p·R (constant sum with fixed internal prices p), fee γ.  Its arbitrage at
    prices v: tender the coin with the lowest v_i/p_i and take out ALL of the coin with the highest, if that pays after
    the fee -- a bang-bang rule, nothing like the closed forms on the device."""

    def __init__(self, R, p, γ, Ai):
        self.R, self.p, self.γ, self.Ai = np.array(R, float), np.array(p, float), float(γ), np.array(Ai, dtype=np.int64)

    def find_arb_(self, Δ, Λ, v):
        Δ[:] = 0.0
        Λ[:] = 0.0
        q = v / self.p                      # market value per unit of pool value, per coin
        lo, hi = int(np.argmin(q)), int(np.argmax(q))
        if self.γ * q[hi] > q[lo]:          # receive hi, tender lo: p_lo·γ·Δ_lo = p_hi·Λ_hi
            Λ[hi] = self.R[hi]
            Δ[lo] = self.p[hi] * self.R[hi] / (self.γ * self.p[lo])


def market(n, m_dev, m_host, seed):
    b = synth.product_pools(m_dev + m_host, n, seed=seed)
    pools = [b[i] for i in range(m_dev + m_host)]
    rng = np.random.default_rng(seed)
    host_at = np.sort(rng.choice(m_dev + m_host, size=m_host, replace=False))
    mixed = list(pools)
    for i in host_at:
        mixed[i] = MyProduct(pools[i].R, pools[i].γ, pools[i].Ai)
    return b, mixed, host_at


def check_fixed_v_and_route(make_backend_router, n, b, mixed, host_at, tol_route):
    v = synth.sweep_prices(n, seed=5, spread=0.3)
    ps = oracle_poolset([b], n)
    D, L = ps.sweep(v, 4)
    psi_o, acc_o = orc.netflows(D, L, ps.Ai, n), orc.dual_acc(D, L, ps.Ai, v)
    r = make_backend_router(mixed)
    try:
        cr.find_arb_(r, v)
        assert rel_to_max(cr.netflows(r), psi_o) <= 1e-12                      # fixed v: the whole router's Ψ
        assert abs(r._acc - acc_o) <= 1e-12 * max(1.0, abs(acc_o))
        Ds, Ls = r.Δs, r.Λs                                                    # router order, host pools' vectors included
        assert len(Ds) == len(mixed)
        for i in list(host_at[:50]) + [0, 1, len(mixed) - 1]:
            np.testing.assert_allclose(Ds[i], D[i], rtol=1e-13, atol=0)
            np.testing.assert_allclose(Ls[i], L[i], rtol=1e-13, atol=0)
        obj = cr.LinearNonnegative(synth.linear_prices(n, seed=6))
        r.objective = obj
        ref = orc.route_oracle(oracle_objective(obj), ps, v0=np.ones(n), nthreads=4)
        for solver in ("scipy", "native"):
            cr.route_(r, v=np.ones(n), solver=solver)
            gap = rel_to_max(cr.netflows(r), ref["psi"])
            assert gap <= tol_route, (solver, gap)
            assert np.min(cr.netflows(r)) >= -1e-6 * np.max(np.abs(ref["psi"]))   # LinearNonnegative: Ψ ≥ 0 at the optimum
    finally:
        r.close()


def test_user_defined_pool_type_over_the_oracle_backend():
    n = 12
    b, mixed, host_at = market(n, 3000, 40, seed=31)
    dev_only = [c for c in mixed if c.kind == KIND_PRODUCT]

    def make(m):
        return cr.Router(cr.LinearNonnegative(np.ones(n)), m, n,
                         _backend=OracleBackend(n, [cr.PoolBatch.from_pools(KIND_PRODUCT, dev_only)], nthreads=4))

    check_fixed_v_and_route(make, n, b, mixed, host_at, 1e-6)


def test_a_pool_without_find_arb_is_rejected_and_update_reserves_needs_the_method():
    class Nothing(cr.CFMM):
        Ai = np.array([1, 2])

    with pytest.raises(cr.ArgumentError, match="find_arb_"):
        cr.Router(cr.LinearNonnegative(np.ones(3)), [Nothing()], 3,
                  _backend=OracleBackend(3, [cr.ProductTwoCoin.batch([[100.0, 100.0]], [1.0], [[1, 2]])], 1))
    pool = ConstantSum3([10.0, 10.0, 10.0], [1.0, 1.0, 1.0], 0.99, [1, 2, 3])
    r = cr.Router(cr.LinearNonnegative(np.ones(3)), [cr.ProductTwoCoin([100.0, 100.0], 1.0, [1, 2]), pool], 3,
                  _backend=OracleBackend(3, [cr.ProductTwoCoin.batch([[100.0, 100.0]], [1.0], [[1, 2]])], 1))
    cr.find_arb_(r, np.array([1.0, 2.0, 1.5]))
    assert len(r.Δs[1]) == 3 and r.Λs[1][1] == 10.0 and r.Δs[1][0] > 0          # three coins: all of coin 2 out, coin 1 in
    with pytest.raises(cr.ArgumentError, match="update_reserves_"):
        cr.update_reserves_(r)


def test_bare_evaluations_leave_host_pool_trades_at_the_find_arb_prices():
    """ADVICE r5 (router.py MixedBackend): an evaluation that materialises nothing on the device (dual_jacobian, polish_, the
    solver's fn/g! calls) must not move the host pools' Δ/Λ either -- r.Δs / r.Λs then all belong to ONE price vector, the
    latest find_arb!'s, and update_reserves! applies consistent trades.  A refused device update leaves host pools alone."""
    n = 8
    b, mixed, host_at = market(n, 200, 6, seed=9)
    dev_only = [c for c in mixed if c.kind == KIND_PRODUCT]
    r = cr.Router(cr.LinearNonnegative(0.1 * np.ones(n)), mixed, n,
                  _backend=OracleBackend(n, [cr.PoolBatch.from_pools(KIND_PRODUCT, dev_only)], nthreads=1))
    v = synth.sweep_prices(n, seed=5, spread=0.3)
    cr.find_arb_(r, v)
    before = [(np.array(r._host.Δs[k]), np.array(r._host.Λs[k])) for k in range(len(host_at))]
    assert any(np.any(d != 0) for d, _ in before)
    r._backend.eval(v * np.linspace(0.5, 1.5, n))          # a bare evaluation at other prices
    r.v[:] = v
    cr.dual_jacobian(r)
    for k, (d, l) in enumerate(before):
        assert np.array_equal(r._host.Δs[k], d) and np.array_equal(r._host.Λs[k], l)
    # the device half refuses (this backend cannot reload): the host pools' reserves and trades are untouched
    R0 = [np.array(c.R) for c in r._host.pools]
    inner = r._backend.inner

    class NoReload:
        n_tokens = n
        eval, find_arb, trades = inner.eval, inner.find_arb, inner.trades

    r._backend.inner = NoReload()
    with pytest.raises(NotImplementedError):
        cr.update_reserves_(r)
    for c, R in zip(r._host.pools, R0):
        assert np.array_equal(c.R, R)
    for k, (d, l) in enumerate(before):
        assert np.array_equal(r._host.Δs[k], d)


@pytest.mark.gpu
def test_user_defined_pool_type_mixed_into_a_device_router():
    """a user-defined CFMM subclass mixed into a 10k-pool router -- fixed v <= 1e-12, route! <= 1e-6
    (both the SciPy-driven loop and csrc/lbfgsb.cpp through the Python callback; cfmm_route's one-call path has no host
    pools and is not used)."""
    n = 24
    b, mixed, host_at = market(n, 10_000, 60, seed=77)
    check_fixed_v_and_route(lambda m: cr.Router(cr.LinearNonnegative(np.ones(n)), m, n), n, b, mixed, host_at, 1e-6)


@pytest.mark.gpu
def test_three_coin_host_pool_next_to_device_pools():
    """A pool type with THREE coins and a bang-bang arbitrage rule next to 5 000 device pools: the router's Ψ is the device's
    plus the pool's own (Λ − Δ), route! stays feasible (Ψ ≥ 0 under LinearNonnegative) and update_reserves! reaches the
    pool's own method."""
    n = 10
    b = synth.product_pools(5_000, n, seed=41)
    big = ConstantSum3([500.0, 400.0, 300.0], [1.0, 1.1, 0.9], 0.997, [2, 5, 7])
    big.update_reserves_ = lambda Δ, Λ, v: big.R.__iadd__(big.γ * Δ - Λ)
    pools = [b[i] for i in range(len(b))] + [big]
    r = cr.Router(cr.LinearNonnegative(synth.linear_prices(n, seed=42)), pools, n)
    try:
        v = synth.sweep_prices(n, seed=43, spread=0.4)
        cr.find_arb_(r, v)
        ps = oracle_poolset([b], n)
        D, L = ps.sweep(v, 4)
        Dh, Lh = np.zeros(3), np.zeros(3)
        big.find_arb_(Dh, Lh, v[[1, 4, 6]])
        psi = orc.netflows(D, L, ps.Ai, n)
        psi[[1, 4, 6]] += Lh - Dh
        assert rel_to_max(cr.netflows(r), psi) <= 1e-12
        cr.route_(r, v=np.ones(n))
        assert np.min(cr.netflows(r)) >= -1e-6 * np.max(np.abs(cr.netflows(r)))
        R_before = big.R.copy()
        Dv, Lv = r.Δs[-1].copy(), r.Λs[-1].copy()
        cr.update_reserves_(r)
        np.testing.assert_allclose(big.R, R_before + big.γ * Dv - Lv)
    finally:
        r.close()
