"""The reference's three examples (examples/arbitrage.jl, liquidate.jl, Univ3.jl) run on the device
through the host mirror and are checked against the CPU restatement."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examples"))

import cfmmrouter_amd as cr
from oracle import cfmm_oracle as orc
from helpers import oracle_objective, oracle_poolset, rel_to_max

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("solver", ["scipy", "native"])
def test_arbitrage_example(solver, capsys):
    import arbitrage
    Ψ, v, D, L = arbitrage.main(solver)
    assert "Profit" in capsys.readouterr().out
    pools = [cr.ProductTwoCoin.batch([[1e6, 1e6], [1e3, 2e3]], [1, 1], [[1, 2], [1, 2]]),
             cr.GeometricMeanTwoCoin.batch([[1e4, 2e4]], [[.4, .6]], [1], [[1, 2]])]
    ref = orc.route_oracle(oracle_objective(cr.LinearNonnegative(np.ones(2))), oracle_poolset(pools, 2))
    assert rel_to_max(Ψ, ref["psi"]) <= 1e-6
    assert np.all(Ψ >= -1e-3) and Ψ.sum() > 100           # an arbitrage profit exists in this market
    np.testing.assert_allclose(D, ref["Delta"], atol=1e-4 * np.max(ref["Delta"]))


@pytest.mark.parametrize("solver", ["scipy", "native"])
def test_liquidate_example(solver, capsys):
    import liquidate
    (Ψ1, v1), (Ψ2, v2) = liquidate.main(solver)
    assert "Amount received" in capsys.readouterr().out
    pools = [cr.ProductTwoCoin.batch([[1e3, 1e4], [1e3, 1e2], [1e3, 2e4]], [0.997] * 3, [[1, 2], [2, 3], [1, 3]])]
    for Ψ, (i, Din) in ((Ψ1, (1, [0, 1e1, 1e2])), (Ψ2, (2, [1e1, 0.0, 0.0]))):
        ref = orc.route_oracle(oracle_objective(cr.BasketLiquidation(i, Din)), oracle_poolset(pools, 3))
        assert rel_to_max(Ψ, ref["psi"]) <= 1e-6
        assert Ψ[i - 1] > 0                                             # something is received
        others = [j for j in range(3) if j != i - 1]
        np.testing.assert_allclose(Ψ[others], -np.asarray(Din)[others], atol=1e-3)  # the basket is fully tendered


def test_univ3_example(capsys):
    import univ3
    Δ, Λ = univ3.main()
    out = capsys.readouterr().out
    assert "Tendered basket" in out and "2: 1.37" in out and "1: 0.072" in out   # SURVEY §3.4 hand trace (≈1.373, ≈0.0722)
    D, L = orc.UniV3(15.0, [30.0, 20, 10, 5], [1.0, 2.0, 1.5, 0.0], 0.997).find_arb([25.0, 1.0])
    np.testing.assert_array_equal(Δ, D)
    np.testing.assert_array_equal(Λ, L)


@pytest.mark.parametrize("devices", ["0", "0,0,0"])
def test_sequential_sharded_example(devices, capsys):
    import sequential_sharded
    first, second = sequential_sharded.main(devices)
    assert "round 2" in capsys.readouterr().out
    assert first > 1e3 and abs(second) <= 1e-6 * first      # the market is cleared after one pass


@pytest.mark.parametrize("solver", ["scipy", "native"])
def test_chain_snapshot_example(solver, capsys):
    """on-chain units -> reference pool types -> route! on the device.  Checked: the trades and netflows at the
    device's v* against a CPU sweep at the same prices (the parity statement), and the profit against the CPU
    restatement's own route! (two L-BFGS-B runs on a stiff 17-pool market agree on the objective, not on every
    component of a flat optimum)."""
    import chain_snapshot
    from helpers import oracle_sweep
    tokens, Ψ, v, batches, D, L = chain_snapshot.main(solver)
    assert "Profit" in capsys.readouterr().out and len(tokens) == 5
    usd = {"USDC": 1.0, "DAI": 1.0, "USDT": 1.0, "FRAX": 0.998, "LUSD": 1.004}
    c = np.array([usd[t] for t in tokens])
    Do, Lo, psio, _ = oracle_sweep(batches, 5, v)
    m_prod, m_geo = len(batches[0]), len(batches[1])
    geo = slice(m_prod, m_prod + m_geo)
    exact = np.r_[0:m_prod, m_prod + m_geo:len(D)]                   # ProductTwoCoin and UniV3 rows: bit for bit
    np.testing.assert_array_equal(D[exact], Do[exact])
    np.testing.assert_array_equal(L[exact], Lo[exact])
    np.testing.assert_allclose(D[geo], Do[geo], rtol=0, atol=1e-12 * np.max(batches[1].R))
    scale = max(np.max(b.R) if hasattr(b, "R") and b.R is not None else 0.0 for b in batches[:2])   # gross flows are of the reserves' order; the net is tiny
    assert np.max(np.abs(Ψ - psio)) <= 1e-12 * scale
    ref = orc.route_oracle(oracle_objective(cr.LinearNonnegative(c)), oracle_poolset(batches, 5), v0=c.copy())
    assert abs(c @ Ψ - c @ ref["psi"]) <= 1e-5 * abs(c @ ref["psi"]) and c @ Ψ > 10
    assert np.all(Ψ >= -1e-3 * np.max(np.abs(Ψ)))
