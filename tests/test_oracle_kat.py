"""Pins the CPU oracle on the reference's own known-answer tests and predicates.

Restates test/cfmms.jl (product KATs :70-98, geo-mean optimality :100-107, UniV3 scenarios
:112-203 with the predicates :3-56) and test/objectives.jl (:1-46) against oracle/cfmm_oracle.c.
Julia's RNG stream cannot be reproduced outside Julia, so the random cases keep the reference's
distributions and predicates with a numpy seed.
"""
import math

import numpy as np
import pytest

from oracle import cfmm_oracle as orc

SQRT_EPS = math.sqrt(np.finfo(float).eps)


def isapprox(a, b, atol=0.0):
    # Julia's isapprox default rtol = sqrt(eps) when atol == 0
    rtol = SQRT_EPS if atol == 0.0 else 0.0
    return abs(a - b) <= max(atol, rtol * max(abs(a), abs(b)))


def optimality_conditions_met(c, D, L, R, gamma, phi, grad_phi):
    """test/cfmms.jl:3-22"""
    Rp = R + gamma * D - L
    pfeas = np.all(D >= 0) and np.all(L >= 0)
    phiR, phiRp = phi(R), phi(Rp)
    g = grad_phi(Rp)
    cfmm_sat = isapprox(phiR, phiRp) and phiRp >= phiR - SQRT_EPS
    opt = max(gamma * g[i] / c[i] for i in range(2)) <= min(g[i] / c[i] for i in range(2)) + SQRT_EPS
    return pfeas and cfmm_sat and opt


class TestProduct:
    def test_no_arb_feeless(self):  # test/cfmms.jl:71-80
        for v in ([1.0, 1.0], [2.0, 2.0]):
            D, L = orc.product_find_arb([1, 1], 1, v)
            assert np.all(D == 0) and np.all(L == 0)

    def test_easy_arb_feeless(self):  # test/cfmms.jl:83-86
        D, L = orc.product_find_arb([1, 1], 1, [2.0, 1.0])
        assert isapprox(D[0], 0) or D[0] == 0
        assert isapprox(D[1], math.sqrt(2) - 1)
        assert isapprox(L[0], 1 - math.sqrt(1 / 2))
        assert L[1] == 0

    def test_exact_literals(self):
        # the same KAT, to the last bit: sqrt/div are correctly rounded on both sides
        D, L = orc.product_find_arb([1, 1], 1, [2.0, 1.0])
        assert D[1] == math.sqrt((1.0 * (2.0 / 1.0)) * 1.0) - 1.0
        assert L[0] == 1.0 - math.sqrt(1.0 / ((2.0 / 1.0) * 1.0))

    def test_random_optimality(self):  # test/cfmms.jl:63-67,92-96
        rng = np.random.default_rng(1234)
        gammas, Rs, nus = rng.random(3), rng.random((3, 2)) * 10, rng.random((3, 2))
        for R in Rs:
            for g in gammas:
                for nu in nus:
                    D, L = orc.product_find_arb(R, g, nu)
                    assert optimality_conditions_met(nu, D, L, R, g, orc.product_phi, orc.product_grad_phi)

    def test_one_direction_only(self):  # at most one direction is non-zero
        rng = np.random.default_rng(7)
        for _ in range(200):
            R, g, nu = rng.random(2) * 1000, rng.choice([0.997, 1.0, 0.9]), rng.random(2) + 0.01
            D, L = orc.product_find_arb(R, g, nu)
            assert not (D[0] > 0 and D[1] > 0)
            assert not (L[0] > 0 and L[1] > 0)


class TestGeoMean:
    def test_random_optimality(self):  # test/cfmms.jl:100-107
        rng = np.random.default_rng(1234)
        gammas, Rs, nus = rng.random(3), rng.random((3, 2)) * 10, rng.random((3, 2))
        ws = [np.array([w1, 1 - w1]) for w1 in rng.random(3)]
        for R in Rs:
            for g in gammas:
                for nu in nus:
                    for w in ws:
                        D, L = orc.geomean_find_arb(R, w, g, nu)
                        assert optimality_conditions_met(
                            nu, D, L, R, g, lambda r: orc.geomean_phi(r, w),
                            lambda r: orc.geomean_grad_phi(r, w))

    def test_equal_weights_is_product(self):
        # w = (1/2, 1/2) is the constant-product pool: same trades up to pow rounding
        rng = np.random.default_rng(5)
        for _ in range(50):
            R, g, nu = rng.random(2) * 1000, 0.997, rng.random(2) + 0.1
            Dg, Lg = orc.geomean_find_arb(R, [0.5, 0.5], g, nu)
            Dp, Lp = orc.product_find_arb(R, g, nu)
            np.testing.assert_allclose(Dg, Dp, rtol=1e-9, atol=1e-9)
            np.testing.assert_allclose(Lg, Lp, rtol=1e-9, atol=1e-9)


# ---- UniV3: test/cfmms.jl:112-203 -------------------------------------------------------------

CURRENT_PRICE = 15.0
LOWER_TICKS = [30.0, 20, 10, 5]
LIQUIDITY = [1.0, 2.0, 1.5, 0.0]
SCENARIOS = [15.0, 16.0, 14.0, 25.0, 7.5, 4.0, 35.0]


def univ3_conditions(c, D, L, pool):
    """test/cfmms.jl:25-56 (ForwardDiff replaced by a central difference)."""
    p_opt = c[0] / c[1]
    g, q = pool.gamma, pool.current_price
    if g * q <= p_opt <= q / g:
        assert D[0] == 0 and D[1] == 0
        return "noarb"
    if p_opt > pool.lower_ticks[0]:
        lam = pool.forward_trade(D)
        assert isapprox(lam, L[0]) and L[1] == 0
        return "drained-down"
    if p_opt < pool.lower_ticks[-1] and pool.liquidity[-1] == 0:
        lam = pool.forward_trade(D)
        assert isapprox(lam, L[1]) and L[0] == 0
        return "drained-up"
    h = 1e-6
    j = 0 if q > p_opt else 1
    e = np.zeros(2)
    e[j] = h
    impact = (pool.forward_trade(D + e) - pool.forward_trade(D - e)) / (2 * h)
    target = p_opt if j == 0 else 1 / p_opt
    assert isapprox(impact, target, atol=1e-6)
    return "interior"


class TestUniV3:
    def test_current_tick(self):  # src/cfmms.jl:235
        assert orc.UniV3(15.0, LOWER_TICKS, LIQUIDITY, 1.0).current_tick == 2
        assert orc.UniV3(20.0, LOWER_TICKS, LIQUIDITY, 1.0).current_tick == 2
        assert orc.UniV3(30.0, LOWER_TICKS, LIQUIDITY, 1.0).current_tick == 1
        assert orc.UniV3(31.0, LOWER_TICKS, LIQUIDITY, 1.0).current_tick == 0
        assert orc.UniV3(1.0, LOWER_TICKS, LIQUIDITY, 1.0).current_tick == 4

    @pytest.mark.parametrize("p", SCENARIOS)
    def test_no_fees(self, p):  # test/cfmms.jl:122-160
        pool = orc.UniV3(CURRENT_PRICE, LOWER_TICKS, LIQUIDITY, 1.0)
        v = np.array([p, 1.0])
        D, L = pool.find_arb(v)
        univ3_conditions(v, D, L, pool)

    @pytest.mark.parametrize("p", [15.0 * (1 + 0.997) / 2] + SCENARIOS[1:])
    def test_fees(self, p):  # test/cfmms.jl:162-201
        pool = orc.UniV3(CURRENT_PRICE, LOWER_TICKS, LIQUIDITY, 0.997)
        v = np.array([p, 1.0])
        D, L = pool.find_arb(v)
        univ3_conditions(v, D, L, pool)

    def test_scenario_kinds(self):
        pool = orc.UniV3(CURRENT_PRICE, LOWER_TICKS, LIQUIDITY, 0.997)
        kinds = [univ3_conditions(np.array([p, 1.0]), *pool.find_arb([p, 1.0]), pool) for p in SCENARIOS[1:]]
        assert kinds == ["interior", "interior", "interior", "interior", "drained-up", "drained-down"]

    def test_example_hand_trace(self):
        # examples/Univ3.jl:27 (v=[25,1], gamma=0.997): tick 2 drained, tick 1 partial
        # (SURVEY 3.4 hand trace: Delta_2 ~ 1.373, Lambda_1 ~ 0.0722)
        pool = orc.UniV3(CURRENT_PRICE, LOWER_TICKS, LIQUIDITY, 0.997)
        D, L = pool.find_arb([25.0, 1.0])
        assert D[0] == 0 and L[1] == 0
        assert abs(D[1] - 1.373) < 2e-3 and abs(L[0] - 0.0722) < 1e-4

    def test_tick_is_bounded_product(self):
        # src/cfmms.jl:294-313: invariant (R1+alpha)(R2+beta) == k at every tick
        pool = orc.UniV3(CURRENT_PRICE, LOWER_TICKS, LIQUIDITY, 1.0)
        for idx in (1, 2, 3):
            t = pool.compute_at_tick(idx)
            assert isapprox((t.R_1 + t.alpha) * (t.R_2 + t.beta), t.k)
        t = pool.compute_at_tick(3)  # above current tick: all in asset 2
        assert t.R_1 == 0.0
        t = pool.compute_at_tick(1)  # below current tick: all in asset 1
        assert t.R_2 == 0.0


# ---- objectives: test/objectives.jl -----------------------------------------------------------

class TestObjectives:
    def test_linear_nonnegative(self):  # :3-18
        with pytest.raises(ValueError):
            orc.LinearNonnegative(-np.ones(2))
        obj = orc.LinearNonnegative(np.ones(2))
        assert obj.f(2 * np.ones(2)) == 0
        assert math.isinf(obj.f(0.5 * np.ones(2)))
        assert np.all(obj.grad(2 * np.ones(2)) == 0)
        assert np.all(np.isinf(obj.grad(0.5 * np.ones(2))))
        np.testing.assert_array_equal(obj.lower_limit(), np.ones(2) + 1e-8)

    def test_basket_liquidation(self):  # :20-33 (i is 0-based in the oracle)
        with pytest.raises(ValueError):
            orc.BasketLiquidation(-1, [0.0, 1.0])
        obj = orc.BasketLiquidation(0, [0, 1])
        assert obj.f([2, 3]) == 3
        assert math.isinf(obj.f(0.5 * np.ones(2)))
        np.testing.assert_array_equal(obj.grad(2 * np.ones(2)), [0, 1])
        assert np.all(np.isinf(obj.grad(0.5 * np.ones(2))))
        lo = obj.lower_limit()
        assert lo[0] == 1 + SQRT_EPS and lo[1] == SQRT_EPS

    def test_swap(self):  # :35-44
        swap = orc.Swap(0, 1, 5.0, 3)
        np.testing.assert_array_equal(swap.Din, [0.0, 5.0, 0.0])
        assert swap.i == 0
