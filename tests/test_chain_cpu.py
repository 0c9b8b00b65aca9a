"""Chain-data intake (cfmmrouter_amd/chain.py): on-chain pool state -> the reference's pool parametrisation.
CPU only; the converted pools are solved by the CPU restatement and checked against concentrated-liquidity swap
arithmetic written independently here (sqrt-price form, raw on-chain units)."""
import json
import math

import numpy as np
import pytest

import cfmmrouter_amd as cr
from cfmmrouter_amd import chain
from cfmmrouter_amd._lib import KIND_GEOMEAN, KIND_PRODUCT, KIND_UNIV3
from oracle import cfmm_oracle as orc

Q96 = 1 << 96


def v3_record(L1, L2, ta, tb, tc, price_tick, d0=18, d1=6, fee_pips=3000):
    """two ranges: [ta, tb) with liquidity L1, [tb, tc) with L2; the pool's price sits at 1.0001^price_tick"""
    sqrt_p = 1.0001 ** (price_tick / 2.0)
    active = L2 if price_tick >= tb else L1
    return {"type": "concentrated", "tokens": ["WETH", "USDC"], "decimals": [d0, d1], "fee_pips": fee_pips,
            "sqrt_price_x96": str(int(sqrt_p * Q96)), "liquidity": str(active),
            "ticks": [[ta, L1], [tb, L2 - L1], [tc, -L2]]}


def test_constant_product_and_weighted_units():
    lines = [json.dumps({"type": "constant_product", "tokens": ["0xA", "0xB"], "decimals": [18, 6],
                         "reserves": ["2500000000000000000000", "5000000000000"], "fee_bps": 30}),
             {"type": "weighted", "tokens": ["0xB", "0xC"], "decimals": [6, 8], "balances": [7_000_000_000, "1234500000000"],
              "weights": [4, 1], "fee": 0.001},
             {"type": "constant_product", "tokens": ["0xC", "0xA"], "reserves": [10**18, 3 * 10**18], "fee_pips": 500}]
    tokens, batches = chain.load_snapshot(lines)
    assert tokens == ["0xA", "0xB", "0xC"]
    bp, bg = batches
    assert bp.kind == KIND_PRODUCT and bg.kind == KIND_GEOMEAN and len(bp) == 2 and len(bg) == 1
    np.testing.assert_array_equal(bp.R, [[2500.0, 5_000_000.0], [1.0, 3.0]])
    np.testing.assert_allclose(bp.γ, [0.997, 0.9995], rtol=0, atol=1e-15)
    np.testing.assert_array_equal(bp.Ai, [[1, 2], [3, 1]])
    np.testing.assert_array_equal(bg.R, [[7000.0, 12345.0]])
    np.testing.assert_allclose(bg.w, [[0.8, 0.2]])
    np.testing.assert_array_equal(bg.Ai, [[2, 3]])


@pytest.mark.parametrize("direction", ["up", "down"])
def test_concentrated_pool_matches_sqrt_price_arithmetic(direction):
    L1, L2 = 3 * 10**15, 5 * 10**15
    ta, tb, tc, d0, d1 = -200_520, -200_280, -200_040, 18, 6           # ETH/USDC-like: 1.0001^-200280 * 1e12 ~ 2000
    rec = v3_record(L1, L2, ta, tb, tc, price_tick=-200_160, d0=d0, d1=d1)
    tokens, (b,) = chain.load_snapshot([rec])
    assert b.kind == KIND_UNIV3 and tokens == ["WETH", "USDC"]
    lo, hi = int(b.tick_off[0]), int(b.tick_off[1])
    ticks, liq, P, g = b.lower_ticks[lo:hi], b.liquidity[lo:hi], float(b.current_price[0]), float(b.γ[0])
    scale = 10.0 ** (d0 - d1)
    np.testing.assert_allclose(ticks, [1.0001 ** tc * scale, 1.0001 ** tb * scale, 1.0001 ** ta * scale], rtol=1e-12)
    np.testing.assert_allclose(liq, [L2**2 / 10.0 ** (d0 + d1), L1**2 / 10.0 ** (d0 + d1), 0.0], rtol=1e-12)
    assert 1900 < P < 2100 and g == 1 - 0.003
    # the converted pool through the CPU restatement of find_arb!(::UniV3): external price m = token1 per token0
    m = P * (1.004 if direction == "up" else 1 / 1.004)                     # beyond the fee, inside the active range
    v = [m, 1.0]                                                         # v1/v2 = m
    D, Lam = orc.UniV3(P, ticks, liq, g).find_arb(v)
    # independent arithmetic in the pool's own (human-unit) sqrt-price form, active range liquidity L2
    Lh = L2 / 10.0 ** ((d0 + d1) / 2)
    sp = math.sqrt(P)
    if direction == "up":      # token1 is tendered until the pool's price is γ·m
        spn = math.sqrt(g * m)
        d_in, l_out = Lh * (spn - sp) / g, Lh * (1 / sp - 1 / spn)
        got_in, got_out, zero = D[1], Lam[0], (D[0], Lam[1])
    else:                      # token0 is tendered until the pool's price is m/γ
        spn = math.sqrt(m / g)
        d_in, l_out = Lh * (1 / spn - 1 / sp) / g, Lh * (sp - spn)
        got_in, got_out, zero = D[0], Lam[1], (D[1], Lam[0])
    assert zero == (0.0, 0.0) and d_in > 0 and l_out > 0
    assert abs(got_in - d_in) <= 1e-9 * d_in and abs(got_out - l_out) <= 1e-9 * l_out


def test_concentrated_pool_crossing_a_tick_conserves_the_invariants():
    """a move that leaves the active range: amounts are the sum over both ranges (sqrt-price arithmetic per range)"""
    L1, L2 = 3 * 10**15, 5 * 10**15
    ta, tb, tc, d0, d1 = -200_520, -200_280, -200_040, 18, 6
    rec = v3_record(L1, L2, ta, tb, tc, price_tick=-200_260, d0=d0, d1=d1)      # just above tb
    _, (b,) = chain.load_snapshot([rec])
    ticks, liq, P, g = b.lower_ticks, b.liquidity, float(b.current_price[0]), float(b.γ[0])
    scale = 10.0 ** (d0 - d1)
    pb = 1.0001 ** tb * scale
    m = pb * 0.99                                                         # pushes the price below tb, into range 1
    D, Lam = orc.UniV3(P, ticks, liq, g).find_arb([m, 1.0])
    h = 10.0 ** ((d0 + d1) / 2)
    spn = math.sqrt(m / g)
    x_in = (L2 / h) * (1 / math.sqrt(pb) - 1 / math.sqrt(P)) + (L1 / h) * (1 / spn - 1 / math.sqrt(pb))
    y_out = (L2 / h) * (math.sqrt(P) - math.sqrt(pb)) + (L1 / h) * (math.sqrt(pb) - spn)
    assert abs(D[0] - x_in / g) <= 1e-9 * x_in and abs(Lam[1] - y_out) <= 1e-9 * y_out


def test_snapshot_validation():
    good = v3_record(10**15, 2 * 10**15, -600, 0, 600, price_tick=100, d0=18, d1=18)
    chain.load_snapshot([good])
    bad = dict(good, liquidity=str(7 * 10**15))
    with pytest.raises(cr.ArgumentError, match="does not match the ticks"):
        chain.load_snapshot([bad])
    bad = dict(good, ticks=[[-600, 10**15], [0, 10**15], [600, -10**15]])
    with pytest.raises(cr.ArgumentError, match="does not sum to zero"):
        chain.load_snapshot([bad])
    with pytest.raises(cr.ArgumentError, match="exactly one of fee"):
        chain.load_snapshot([{"type": "constant_product", "tokens": ["a", "b"], "reserves": [1, 1]}])
    with pytest.raises(cr.ArgumentError, match="two distinct"):
        chain.load_snapshot([{"type": "constant_product", "tokens": ["a", "a"], "reserves": [1, 1], "fee": 0.0}])
    with pytest.raises(cr.ArgumentError, match="unknown pool type"):
        chain.load_snapshot([{"type": "stableswap", "tokens": ["a", "b"], "fee": 0.0}])


def test_price_above_the_highest_tick_gets_an_empty_interval():
    rec = v3_record(10**15, 2 * 10**15, -600, 0, 600, price_tick=900, d0=18, d1=18)
    rec["liquidity"] = "0"
    _, (b,) = chain.load_snapshot([rec])
    assert b.liquidity[0] == 0.0 and b.lower_ticks[0] > b.current_price[0] > b.lower_ticks[1]
    D, Lam = orc.UniV3(float(b.current_price[0]), b.lower_ticks, b.liquidity, float(b.γ[0])).find_arb([1.0, 1.0])
    assert D[0] > 0 and Lam[1] > 0          # the market price 1.0 is below the pool's: token0 flows in through the ranges below
