"""Committed golden vectors (tests/golden/golden.npz, made by tests/golden/make_golden.py from the
CPU oracle).  CPU: the oracle still reproduces them bit for bit.  GPU: the HIP path matches them."""
import os

import numpy as np
import pytest

import cfmmrouter_amd as cr
from cfmmrouter_amd import synth
from oracle import cfmm_oracle as orc
from helpers import oracle_objective, oracle_poolset, oracle_sweep, rel_to_max

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden.npz"))

CASES = {
    "prod": (lambda: synth.product_pools(257, 7, seed=11), 7),
    "geo": (lambda: synth.geomean_pools(129, 5, seed=12), 5),
    "uni": (lambda: synth.univ3_pools(64, 9, 7, seed=13), 9),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_golden(name):
    make, n = CASES[name]
    D, L, psi, acc = oracle_sweep([make()], n, G[f"{name}_v"])
    tol = dict(rtol=0, atol=0) if name != "geo" else dict(rtol=1e-13, atol=1e-13)  # geo: libm pow may differ by an ulp across hosts
    np.testing.assert_allclose(D, G[f"{name}_D"], **tol)
    np.testing.assert_allclose(L, G[f"{name}_L"], **tol)
    np.testing.assert_allclose(psi, G[f"{name}_psi"], rtol=1e-13, atol=1e-12)


def test_oracle_reproduces_univ3_fixture():
    for g, p, d1, d2, l1, l2 in G["uni_fixture"]:
        D, L = orc.UniV3(15.0, [30.0, 20, 10, 5], [1.0, 2.0, 1.5, 0.0], g).find_arb([p, 1.0])
        assert (D[0], D[1], L[0], L[1]) == (d1, d2, l1, l2)


def test_oracle_route_reproduces_golden():
    b = synth.product_pools(100, 10, seed=21)
    for name, obj, v0 in (("arb", cr.LinearNonnegative(synth.linear_prices(10, seed=21)), np.ones(10)),
                          ("basket", cr.BasketLiquidation(1, synth.basket(10, seed=22)), None)):
        ref = orc.route_oracle(oracle_objective(obj), oracle_poolset([b], 10), v0=v0)
        assert rel_to_max(ref["psi"], G[f"route_{name}_psi"]) <= 1e-9


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_gpu_matches_golden(name):
    make, n = CASES[name]
    be = cr.DeviceBackend(n, [make()])
    psi, acc = be.find_arb(G[f"{name}_v"])
    D, L = be.trades()
    be.close()
    if name == "geo":
        np.testing.assert_allclose(D, G["geo_D"], rtol=0, atol=1e-12 * 1000)
        np.testing.assert_allclose(L, G["geo_L"], rtol=0, atol=1e-12 * 1000)
    else:
        np.testing.assert_array_equal(D, G[f"{name}_D"])
        np.testing.assert_array_equal(L, G[f"{name}_L"])
    assert rel_to_max(psi, G[f"{name}_psi"]) <= 1e-12
    assert abs(acc - float(G[f"{name}_acc"])) <= 1e-11 * max(1.0, abs(float(G[f"{name}_acc"])))


@pytest.mark.gpu
def test_gpu_univ3_fixture_golden():
    for g, p, d1, d2, l1, l2 in G["uni_fixture"]:
        Δ, Λ = np.zeros(2), np.zeros(2)
        cr.find_arb_(Δ, Λ, cr.UniV3(15.0, [30.0, 20, 10, 5], [1.0, 2.0, 1.5, 0.0], g, [1, 2]), [p, 1.0])
        assert (Δ[0], Δ[1], Λ[0], Λ[1]) == (d1, d2, l1, l2)


@pytest.mark.gpu
@pytest.mark.parametrize("solver", ["scipy", "native"])
def test_gpu_route_matches_golden(solver):
    b = synth.product_pools(100, 10, seed=21)
    for name, obj, v0 in (("arb", cr.LinearNonnegative(synth.linear_prices(10, seed=21)), np.ones(10)),
                          ("basket", cr.BasketLiquidation(1, synth.basket(10, seed=22)), None)):
        r = cr.Router(obj, b, 10)
        cr.route_(r, v=v0, solver=solver)
        assert rel_to_max(cr.netflows(r), G[f"route_{name}_psi"]) <= 1e-6
        np.testing.assert_allclose(r.v, G[f"route_{name}_v"], rtol=1e-6)
        r.close()
