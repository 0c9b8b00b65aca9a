"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle.

Bar (BASELINE.json north_star): netflows within 1e-6 relative.  What is actually asserted is far
tighter and stated per test: ProductTwoCoin / UniV3 trades BIT-EXACT (IEEE sqrt and / on both
sides, same operation order), GeometricMeanTwoCoin within a few ulp (different pow
implementations), reductions within 1e-12 of max|Ψ| (different summation order).
"""
import math
import os

import numpy as np
import pytest

import cfmmrouter_amd as cr
from cfmmrouter_amd import synth
from oracle import cfmm_oracle as orc
from helpers import oracle_objective, oracle_poolset, oracle_sweep, rel_to_max, route_converged

pytestmark = pytest.mark.gpu

REDUCE_TOL = 1e-12   # of max|Ψ|: fp64 sums in a different order
GEOM_RTOL = 1e-12    # per trade, relative to the reserve scale
ROUTE_TOL = 1e-6     # north_star tolerance on route!-level netflows, relative to max|Ψ|


def device_sweep(batches, n, v, materialize=True, **opts):
    be = cr.DeviceBackend(n, batches)
    for k, val in opts.items():
        be.ctx.set_option(k, val)
    try:
        if materialize:
            psi, acc = be.find_arb(v)
            D, L = be.trades()
        else:
            psi, acc = be.eval(v)
            D = L = None
        return D, L, psi, acc
    finally:
        be.close()


# ---- reference KATs straight through the device (test/cfmms.jl:70-90) ---------------------------

class TestProductKAT:
    def test_no_arb_and_easy_arb(self):
        Δ, Λ = np.empty(2), np.empty(2)
        equal_pool = cr.ProductTwoCoin([1, 1], 1, [1, 2])
        for v in ([1.0, 1.0], [2.0, 2.0]):
            cr.find_arb_(Δ, Λ, equal_pool, v)
            assert np.all(Δ == 0) and np.all(Λ == 0)
        cr.find_arb_(Δ, Λ, equal_pool, [2.0, 1.0])
        assert Δ[0] == 0 and Δ[1] == math.sqrt(2) - 1
        assert Λ[0] == 1 - math.sqrt(1 / 2) and Λ[1] == 0

    def test_ctor_errors(self):
        assert len(cr.ProductTwoCoin([1, 1], .9, [1, 2])) == 2
        with pytest.raises(cr.ArgumentError):
            cr.ProductTwoCoin([1, 1], .9, [1])


class TestUniV3KAT:
    """test/cfmms.jl:112-203 on the device; the oracle already satisfies the reference's predicates
    (tests/test_oracle_kat.py), so bit-equality with it carries them over."""

    @pytest.mark.parametrize("γ", [1.0, 0.997])
    @pytest.mark.parametrize("p", [15.0, 15.0 * (1 + 0.997) / 2, 16.0, 14.0, 25.0, 7.5, 4.0, 35.0])
    def test_fixture(self, γ, p):
        cfmm = cr.UniV3(15.0, [30., 20, 10, 5], [1.0, 2.0, 1.5, 0.0], γ, [1, 2])
        assert cfmm.current_tick == 2
        Δ, Λ = np.zeros(2), np.zeros(2)
        cr.find_arb_(Δ, Λ, cfmm, [p, 1.0])
        D, L = orc.UniV3(15.0, [30., 20, 10, 5], [1.0, 2.0, 1.5, 0.0], γ).find_arb([p, 1.0])
        np.testing.assert_array_equal(Δ, D)
        np.testing.assert_array_equal(Λ, L)


# ---- fixed-v sweeps ---------------------------------------------------------------------------------

@pytest.mark.parametrize("m,n", [(1, 2), (255, 7), (256, 64), (257, 64), (1000, 3), (100_000, 64), (300_001, 257)])
def test_product_sweep_bit_exact(m, n):
    b = synth.product_pools(m, n, seed=m + n)
    v = synth.sweep_prices(n, seed=m)
    D, L, psi, acc = device_sweep([b], n, v)
    Do, Lo, psio, acco = oracle_sweep([b], n, v)
    np.testing.assert_array_equal(D, Do)
    np.testing.assert_array_equal(L, Lo)
    assert rel_to_max(psi, psio) <= REDUCE_TOL
    assert abs(acc - acco) <= REDUCE_TOL * max(abs(acco), 1.0)


def test_product_bit_exact_over_wide_dynamic_range():
    """The direction-select evaluation must equal the reference's four closed forms bit for bit
    everywhere, including pools sitting exactly on / next to the no-arbitrage boundary."""
    rng = np.random.default_rng(2024)
    m, n = 400_000, 32
    R = 10.0 ** rng.uniform(-9, 12, (m, 2))
    γ = rng.choice([1.0, 0.997, 0.9999999999, 0.5, 0.01, 1.0 - 2.0 ** -52], m)
    Ai = synth.token_pairs(9, 1, m, n)
    v = 10.0 ** rng.uniform(-6, 6, n)
    # a quarter of the pools are placed ON the boundary v1·R1 == γ·v2·R2 (up to rounding), and
    # another quarter at the no-fee parity point v1·R1 == v2·R2
    i1, i2 = Ai[:, 0] - 1, Ai[:, 1] - 1
    q = m // 4
    R[:q, 0] = γ[:q] * v[i2[:q]] * R[:q, 1] / v[i1[:q]]
    R[q:2 * q, 0] = v[i2[q:2 * q]] * R[q:2 * q, 1] / v[i1[q:2 * q]]
    R[:2 * q, 0] *= 1.0 + rng.integers(-3, 4, 2 * q) * 2.0 ** -52      # ± a few ulp around it
    b = cr.ProductTwoCoin.batch(R, γ, Ai)
    D, L, psi, acc = device_sweep([b], n, v)
    Do, Lo, psio, acco = oracle_sweep([b], n, v)
    np.testing.assert_array_equal(D, Do)
    np.testing.assert_array_equal(L, Lo)
    assert np.count_nonzero(D[:2 * q]) > 0 and np.count_nonzero(D[:2 * q] == 0) > 0
    assert rel_to_max(psi, psio) <= 1e-11


def test_univ3_bit_exact_ragged_and_degenerate():
    """ragged tick counts 1..40, empty current ticks, prices on tick boundaries, γ = 1 and tiny."""
    rng = np.random.default_rng(7)
    m, n = 30_000, 16
    nt = rng.integers(1, 41, m)
    off = np.zeros(m + 1, dtype=np.int64)
    np.cumsum(nt, out=off[1:])
    T = int(off[-1])
    ticks, liq, cp = np.empty(T), np.empty(T), np.empty(m)
    for i in range(m):
        t = np.sort(10.0 ** rng.uniform(-3, 3, nt[i]))[::-1]
        t *= 1.0 + np.arange(nt[i])[::-1] * 1e-9            # strictly descending even on ties
        ticks[off[i]:off[i + 1]] = t
        lq = 10.0 ** rng.uniform(-2, 6, nt[i])
        lq[rng.random(nt[i]) < 0.3] = 0.0                   # empty ticks, possibly the current one
        liq[off[i]:off[i + 1]] = lq
        k = rng.integers(0, nt[i])
        cp[i] = t[k] if rng.random() < 0.3 else t[k] * rng.uniform(0.5, 1.0)   # on a boundary 30 % of the time
    γ = rng.choice([1.0, 0.997, 0.3], m)
    b = cr.UniV3.batch(cp, off, ticks, liq, γ, synth.token_pairs(5, 1, m, n))
    for spread in (0.05, 3.0):
        v = synth.sweep_prices(n, seed=int(spread * 100), spread=spread)
        D, L, psi, acc = device_sweep([b], n, v)
        Do, Lo, psio, acco = oracle_sweep([b], n, v)
        np.testing.assert_array_equal(D, Do)
        np.testing.assert_array_equal(L, Lo)
        assert rel_to_max(psi, psio) <= 1e-11


def test_univ3_walk_decisions_at_tick_boundaries():
    """The walk's drain decisions are taken on thresholds prepared at upload (UniV3Ops::solve_dir): `price <= T_j` must be
    the reference's floating-point test `sqrt(k/price) − s_in >= δmax`, also when the price sits ON a tick boundary or a
    few ulps beside it.  3000 two-token pools (γ = 1, so the walk's internal price is v₁/v₂ itself in one direction and
    its reciprocal in the other) whose ladders put a boundary exactly at, and −3..+3 ulps around, the swept price."""
    n, m = 2, 3000
    rng = np.random.default_rng(7)
    price = 1.37
    nt = 8
    off = nt * np.arange(m + 1, dtype=np.int64)
    ticks, liq, cp = np.empty(m * nt), np.empty(m * nt), np.empty(m)
    for i in range(m):
        e = int(rng.integers(-3, 4))
        hit = np.nextafter(price, np.inf) if e > 0 else (np.nextafter(price, 0) if e < 0 else price)
        for _ in range(abs(e) - 1):
            hit = np.nextafter(hit, np.inf if e > 0 else 0)
        hit = hit if rng.random() < 0.5 else 1.0 / hit             # falling-price walks and rising-price walks
        j = int(rng.integers(1, nt - 1))                           # which boundary carries the hit
        lad = hit * 1.05 ** (j - np.arange(nt))                    # strictly descending, lad[j] == hit
        ticks[off[i]:off[i + 1]] = lad
        lq = 10.0 ** rng.uniform(2, 6, nt)
        lq[rng.random(nt) < 0.15] = 0.0
        liq[off[i]:off[i + 1]] = lq
        k = int(rng.integers(0, nt))                               # current price somewhere in the ladder
        cp[i] = lad[k] if rng.random() < 0.3 else lad[k] * rng.uniform(0.96, 1.0)
    Ai = np.tile(np.array([[1, 2]]), (m, 1))
    Ai[rng.random(m) < 0.5] = [2, 1]
    b = cr.UniV3.batch(cp, off, ticks, liq, np.ones(m), Ai)
    for v in ([price, 1.0], [1.0, price], [np.nextafter(price, 2), 1.0], [1.0, np.nextafter(price, 0)]):
        v = np.array(v)
        for fast in (1, 0):
            D, L, psi, acc = device_sweep([b], n, v, fast_math=fast)
            Do, Lo, psio, acco = oracle_sweep([b], n, v)
            np.testing.assert_array_equal(D, Do)
            np.testing.assert_array_equal(L, Lo)


@pytest.mark.parametrize("t", [3, 64, 65, 200])
def test_univ3_deep_walks(t):
    """Ladders of t ticks with the external price far outside: walks cross up to ~t ticks -- the drained ticks through
    the prefix sums of the records (threshold scan over 1 .. 25 lines), the last one tick by tick.  Bit-exact."""
    m, n = 3000, 8
    b = synth.univ3_pools(m, n, t, seed=t)
    for spread in (0.3, 8.0):                       # e^±8: every pool is drained in one direction
        v = synth.sweep_prices(n, seed=t, spread=spread)
        D, L, psi, acc = device_sweep([b], n, v)
        Do, Lo, psio, acco = oracle_sweep([b], n, v)
        np.testing.assert_array_equal(D, Do)
        np.testing.assert_array_equal(L, Lo)
        assert rel_to_max(psi, psio) <= 1e-11
    # mixed launch (sweep_multi): the same walk next to the other families
    D, L, psi, acc = device_sweep([synth.product_pools(5000, n, 1), b], n, v)
    Do, Lo, psio, acco = oracle_sweep([synth.product_pools(5000, n, 1), b], n, v)
    np.testing.assert_array_equal(D, Do)
    np.testing.assert_array_equal(L, Lo)


@pytest.mark.parametrize("fast", [1, 0])
@pytest.mark.parametrize("copies", [1, 2])
@pytest.mark.parametrize("block", [512, 1024])
def test_product_launch_variants(fast, copies, block):
    m, n = 70_001, 96
    b = synth.product_pools(m, n, seed=3)
    v = synth.sweep_prices(n, seed=4)
    D, L, psi, acc = device_sweep([b], n, v, fast_math=fast, bin_copies=copies, max_grid=7, block=block)
    Do, Lo, psio, acco = oracle_sweep([b], n, v)
    np.testing.assert_array_equal(D, Do)
    np.testing.assert_array_equal(L, Lo)
    assert rel_to_max(psi, psio) <= REDUCE_TOL


@pytest.mark.parametrize("exact", [0, 1])   # 0: log-space forms (default), 1: pow in reference order
@pytest.mark.parametrize("m,n", [(1, 2), (513, 16), (50_000, 256)])
def test_geomean_sweep(m, n, exact):
    b = synth.geomean_pools(m, n, seed=m)
    v = synth.sweep_prices(n, seed=n)
    D, L, psi, acc = device_sweep([b], n, v, geomean_exact=exact)
    Do, Lo, psio, acco = oracle_sweep([b], n, v)
    scale = np.max(b.R, axis=1, keepdims=True)
    assert np.max(np.abs(D - Do) / scale) <= GEOM_RTOL
    assert np.max(np.abs(L - Lo) / scale) <= GEOM_RTOL
    assert np.array_equal(D > 0, Do > 0) or np.max(np.abs(D - Do)[(D > 0) != (Do > 0)] / scale.repeat(2, 1)[(D > 0) != (Do > 0)]) <= 1e-14
    assert rel_to_max(psi, psio) <= 1e-11
    assert abs(acc - acco) <= 1e-10 * max(abs(acco), 1.0)


@pytest.mark.parametrize("m,n,t", [(1, 2, 1), (300, 8, 2), (4097, 64, 5), (20_000, 128, 17)])
def test_univ3_sweep_bit_exact(m, n, t):
    b = synth.univ3_pools(m, n, t, seed=m) if t != 2 else synth.bounded_product_pools(m, n, seed=m)
    v = synth.sweep_prices(n, seed=t, spread=1.0)
    D, L, psi, acc = device_sweep([b], n, v)
    Do, Lo, psio, acco = oracle_sweep([b], n, v)
    np.testing.assert_array_equal(D, Do)
    np.testing.assert_array_equal(L, Lo)
    assert m == 1 or np.count_nonzero(D) > 0
    assert rel_to_max(psi, psio) <= REDUCE_TOL


def test_geomean_extreme_weights_and_fees():
    """η = w₁/w₂ from 1/19 to 19, γ down to 0.5, reserves over 8 decades: log-space vs reference order.
    (Beyond η·|log₁₀R| ≈ 308 the reference's own r2^η overflows to Inf; the log-space form does not.)"""
    rng = np.random.default_rng(11)
    m, n = 20_000, 12
    w1 = rng.uniform(0.05, 0.95, m)
    R = 10.0 ** rng.uniform(-4, 4, (m, 2))
    γ = rng.choice([0.5, 0.9, 0.997, 1.0], m)
    b = cr.GeometricMeanTwoCoin.batch(R, np.stack([w1, 1 - w1], 1), γ, synth.token_pairs(3, 1, m, n))
    v = 10.0 ** rng.uniform(-3, 3, n)
    Do, Lo, psio, _ = oracle_sweep([b], n, v)
    scale = np.max(R, axis=1, keepdims=True)
    for exact in (0, 1):
        D, L, psi, _ = device_sweep([b], n, v, geomean_exact=exact)
        assert np.max(np.abs(D - Do) / np.maximum(scale, Do)) <= 1e-11
        assert np.max(np.abs(L - Lo) / scale) <= 1e-11
        assert rel_to_max(psi, psio) <= 1e-10


def test_mixed_segments_and_router_order():
    """Config-3 shape in miniature + a router whose cfmms vector interleaves the families."""
    n = 32
    bp, bg, bu = synth.product_pools(700, n, 1), synth.geomean_pools(300, n, 2), synth.univ3_pools(100, n, 4, 3)
    v = synth.sweep_prices(n, seed=9, spread=0.5)
    D, L, psi, acc = device_sweep([bp, bg, bu], n, v)
    Do, Lo, psio, acco = oracle_sweep([bp, bg, bu], n, v)
    np.testing.assert_array_equal(D[:700], Do[:700])
    np.testing.assert_array_equal(D[1000:], Do[1000:])
    assert np.max(np.abs(D[700:1000] - Do[700:1000])) <= 1e-9
    assert rel_to_max(psi, psio) <= 1e-11
    # interleaved object list -> packed by family, results un-permuted to router order
    pools = []
    for i in range(100):
        pools += [bp[i], bg[i], bu[i]]
    r = cr.Router(cr.LinearNonnegative(np.ones(n)), pools, n)
    cr.find_arb_(r, v)
    for i in range(100):
        np.testing.assert_array_equal(r.Δs[3 * i], Do[i])
        np.testing.assert_array_equal(r.Δs[3 * i + 2], Do[1000 + i])
        assert np.max(np.abs(r.Λs[3 * i + 1] - Lo[700 + i])) <= 1e-9
    r.close()


@pytest.mark.parametrize("fuse", [0, 1])
def test_many_segments_one_launch(fuse):
    """6 segments (> kMaxMulti = 4 per launch) of all three families, fused and per-segment launches."""
    n = 40
    segs = [synth.product_pools(3000, n, 1), synth.geomean_pools(1500, n, 2), synth.univ3_pools(700, n, 6, 3),
            synth.product_pools(10, n, 4), synth.bounded_product_pools(2500, n, 5), synth.geomean_pools(1, n, 6)]
    v = synth.sweep_prices(n, seed=2, spread=0.7)
    D, L, psi, acc = device_sweep(segs, n, v, fuse_segments=fuse)
    Do, Lo, psio, acco = oracle_sweep(segs, n, v)
    scale = 1e3
    assert np.max(np.abs(D - Do)) <= 1e-12 * max(scale, np.max(Do))
    assert np.max(np.abs(L - Lo)) <= 1e-12 * max(scale, np.max(Lo))
    np.testing.assert_array_equal(D[:3000], Do[:3000])                 # product rows stay bit-exact
    np.testing.assert_array_equal(D[4500:5200], Do[4500:5200])         # univ3 rows stay bit-exact
    assert rel_to_max(psi, psio) <= 1e-11
    assert abs(acc - acco) <= 1e-10 * max(abs(acco), 1.0)
    D2, L2, psi2, _ = device_sweep(segs, n, v, materialize=False, fuse_segments=fuse)
    np.testing.assert_array_equal(psi2, psi)


def test_fused_equals_materialised_and_deterministic():
    m, n = 200_000, 64
    b = synth.product_pools(m, n, seed=5)
    v = synth.sweep_prices(n, seed=6)
    be = cr.DeviceBackend(n, [b])
    psi1, acc1 = be.find_arb(v)
    psi2, acc2 = be.eval(v)
    psi3, acc3 = be.find_arb(v)
    np.testing.assert_array_equal(psi1, psi2)   # same kernel body, same launch geometry
    np.testing.assert_array_equal(psi1, psi3)   # no float atomics in global memory: reproducible
    assert acc1 == acc2 == acc3
    be.close()


def test_zero_copy_and_copy_paths_agree():
    n = 100
    b = synth.product_pools(40_000, n, seed=4)
    be = cr.DeviceBackend(n, [b])
    for seed in (1, 2, 3):          # fresh v every call: the kernels must see the new host values
        v = synth.sweep_prices(n, seed=seed)
        psi_z, acc_z = be.eval(v)
        be.ctx.set_option("zero_copy", 0)
        psi_c, acc_c = be.eval(v)
        be.ctx.set_option("zero_copy", 1)
        np.testing.assert_array_equal(psi_z, psi_c)
        assert acc_z == acc_c
        _, _, psio, _ = oracle_sweep([b], n, v)
        assert rel_to_max(psi_z, psio) <= REDUCE_TOL
    be.close()


def test_empty_router_and_bad_inputs():
    be = cr.DeviceBackend(4, [])
    psi, acc = be.find_arb(np.ones(4))
    assert np.all(psi == 0) and acc == 0
    with pytest.raises(cr.ArgumentError):
        be.ctx.add_product([[1.0, 2.0]], [1.0], [[0, 0]])         # same token twice
    with pytest.raises(cr.ArgumentError):
        be.ctx.add_product([[1.0, 2.0]], [1.0], [[0, 4]])         # index out of range
    with pytest.raises(cr.ArgumentError):
        be.ctx.add_product([[0.0, 2.0]], [1.0], [[0, 1]])         # empty reserve
    with pytest.raises(cr.ArgumentError):
        be.ctx.add_product([[1.0, 2.0]], [0.0], [[0, 1]])         # γ = 0
    with pytest.raises(cr.ArgumentError):
        be.ctx.add_univ3([1.0], [1.0], [[0, 1]], [0, 2], [1.0, 2.0], [1.0, 1.0])  # ascending ticks
    with pytest.raises(cr.ArgumentError):
        be.ctx.add_univ3([3.0], [1.0], [[0, 1]], [0, 2], [2.0, 1.0], [1.0, 1.0])  # price above first tick
    with pytest.raises(cr.ArgumentError):
        be.find_arb(np.array([1.0, 0.0, 1.0, 1.0]))               # v must be > 0
    with pytest.raises(RuntimeError):
        cr.DeviceBackend(4, []).trades()                           # no sweep yet
    be.close()


def test_max_tokens():
    n = 8192
    b = synth.product_pools(30_000, n, seed=8)
    v = synth.sweep_prices(n, seed=8)
    D, L, psi, acc = device_sweep([b], n, v)
    Do, Lo, psio, acco = oracle_sweep([b], n, v)
    np.testing.assert_array_equal(D, Do)
    assert rel_to_max(psi, psio) <= REDUCE_TOL


@pytest.mark.parametrize("n", [8193, 50_000])
def test_large_market_global_bins(n):
    """n_tokens > 8192: v gathered from global memory, flows added with global f64 atomics."""
    segs = [synth.product_pools(120_000, n, seed=1), synth.geomean_pools(30_000, n, seed=2),
            synth.univ3_pools(10_000, n, 3, seed=3)]
    v = synth.sweep_prices(n, seed=5, spread=0.6)
    for fuse in (1, 0):
        D, L, psi, acc = device_sweep(segs, n, v, fuse_segments=fuse)
        Do, Lo, psio, acco = oracle_sweep(segs, n, v)
        np.testing.assert_array_equal(D[:120_000], Do[:120_000])
        np.testing.assert_array_equal(D[150_000:], Do[150_000:])
        assert np.max(np.abs(D - Do)) <= 1e-9 and np.max(np.abs(L - Lo)) <= 1e-9
        assert rel_to_max(psi, psio) <= 1e-12
        assert abs(acc - acco) <= 1e-11 * max(abs(acco), 1.0)
    psi2 = device_sweep(segs, n, v, materialize=False)[2]
    assert rel_to_max(psi2, psio) <= 1e-12
    np.testing.assert_array_equal(psi2, psi)      # pull-based gather: fixed summation order, reproducible


def test_large_market_hub_token():
    """a numeraire that sits in every pool (degree = m): its incidence list is cut into many chunks"""
    n, m = 9000, 200_000
    rng = np.random.default_rng(3)
    other = rng.integers(2, n + 1, m)
    Ai = np.stack([np.ones(m, dtype=np.int64), other], 1)
    flip = rng.random(m) < 0.5
    Ai[flip] = Ai[flip][:, ::-1]
    b = cr.ProductTwoCoin.batch(1000 * rng.random((m, 2)) + 1, rng.choice([0.997, 1.0], m), Ai)
    v = synth.sweep_prices(n, seed=1, spread=0.3)
    D, L, psi, acc = device_sweep([b], n, v)
    Do, Lo, psio, acco = oracle_sweep([b], n, v)
    np.testing.assert_array_equal(D, Do)
    assert rel_to_max(psi, psio) <= 1e-12
    assert abs(psi[0] - psio[0]) <= 1e-11 * abs(psio[0])


def test_contexts_with_different_token_counts_coexist():
    """the dynamic-LDS attribute is process-wide: a small context must not shrink a big one's limit"""
    big = cr.DeviceBackend(8192, [synth.product_pools(5000, 8192, seed=1)])
    small = cr.DeviceBackend(4, [synth.product_pools(50, 4, seed=2)])
    huge = cr.DeviceBackend(20_000, [synth.product_pools(500, 20_000, seed=3)])
    for be, n in ((small, 4), (huge, 20_000), (big, 8192), (small, 4)):
        psi, _ = be.find_arb(synth.sweep_prices(n, seed=n))
        assert np.all(np.isfinite(psi))
    for be in (big, small, huge):
        be.close()


# ---- route! ---------------------------------------------------------------------------------------------

def route_both(objective, batches_or_pools, n, v0=None):
    r = cr.Router(objective, batches_or_pools, n)
    cr.route_(r, v=v0)
    ref = orc.route_oracle(oracle_objective(objective), oracle_poolset(r._batches, n), v0=v0)
    return r, ref


def test_route_readme_example():
    """README.md:27-38 (config 1)."""
    pools = [cr.ProductTwoCoin([1e6, 1e6], 1, [1, 2]), cr.ProductTwoCoin([1e3, 2e3], 1, [1, 2])]
    r, ref = route_both(cr.LinearNonnegative(np.ones(2)), pools, 2)
    Ψ = cr.netflows(r)
    assert rel_to_max(Ψ, ref["psi"]) <= ROUTE_TOL
    assert abs(Ψ[1] - 171.4) < 0.1 and abs(Ψ[0]) < 1e-3     # SURVEY §8c analytic cross-check
    np.testing.assert_allclose(r.v, ref["v"], rtol=1e-9)
    r.close()


@pytest.mark.parametrize("m,n", [(100, 10), (10_000, 100), (100_000, 64)])
def test_route_arbitrage_parity(m, n):
    """test/arb.jl:61-86 shape (m=100) up to config 2 (100k pools, 64 tokens)."""
    b = synth.product_pools(m, n, seed=1234)
    obj = cr.LinearNonnegative(synth.linear_prices(n, seed=1234))
    r, ref = route_both(obj, b, n, v0=np.ones(n))
    assert rel_to_max(cr.netflows(r), ref["psi"]) <= ROUTE_TOL
    # feasibility as in test/arb.jl:22 (all_flows >= -TOL), with the CPU restatement's own worst component as the
    # yardstick: L-BFGS-B stops on factr, which leaves constraint-active netflows slightly negative on BOTH sides
    # (how negative depends on the last iterate, i.e. on summation-order rounding of Ψ), never more than that
    floor = min(float(np.min(ref["psi"])), 0.0)
    assert np.all(cr.netflows(r) >= 3 * floor - 1e-3 - 1e-8 * np.max(np.abs(ref["psi"])))
    assert np.all(r.v >= cr.lower_limit(obj) - 1e-4)
    r.close()


def test_netflows_exact_is_the_reference_loop_bit_for_bit():
    """VERDICT r5 missing #4 / test/arb.jl:16: `all(all_flows .== netflows(r))` -- exact equality with the serial pool-order
    sum over r.Δs / r.Λs.  netflows(r, exact=True) IS that loop (src/router.jl:113-116) over the fetched rows; the default
    stays the device's reduction (<= 1e-12 of it).  A router built from pool objects in SHUFFLED family order (the packing
    permutes them; router order is the caller's), after route!, and a 300k-pool batch router after find_arb!."""
    n = 12
    bp, bg = synth.product_pools(3000, n, seed=51), synth.geomean_pools(2000, n, seed=52)
    bu = synth.univ3_pools(500, n, 5, seed=53)
    pools = [bp[i] for i in range(len(bp))] + [bg[i] for i in range(len(bg))] + [bu[i] for i in range(len(bu))]
    np.random.default_rng(5).shuffle(pools)
    r = cr.Router(cr.LinearNonnegative(synth.linear_prices(n, seed=54)), pools, n)
    try:
        cr.route_(r, v=np.ones(n))
        all_flows = np.zeros(n)
        for Δ, Λ, c in zip(r.Δs, r.Λs, r.cfmms):                       # test/arb.jl:8-14
            all_flows[c.Ai - 1] += Λ - Δ
        assert np.all(all_flows == cr.netflows(r, exact=True))          # test/arb.jl:16
        ψ = np.empty(n)
        cr.netflows_(ψ, r, exact=True)
        assert np.array_equal(ψ, all_flows)
        assert rel_to_max(cr.netflows(r), all_flows) <= 1e-12           # the device's reduction of the same sweep
    finally:
        r.close()
    n, m = 64, 300_000
    b = synth.product_pools(m, n, seed=55)
    r = cr.Router(cr.LinearNonnegative(np.ones(n)), b, n)
    try:
        cr.find_arb_(r, synth.sweep_prices(n, seed=56))
        ref = orc.netflows(r.Δs, r.Λs, (b.Ai - 1).astype(np.int32), n)  # the restatement's serial loop (oracle/cfmm_oracle.c)
        assert np.array_equal(cr.netflows(r, exact=True), ref)
        assert rel_to_max(cr.netflows(r), ref) <= 1e-12
    finally:
        r.close()


def test_route_basket_liquidation_parity():
    """test/swap.jl:20-46 shape + examples/liquidate.jl."""
    n, m = 10, 100
    b = synth.product_pools(m, n, seed=77)
    obj = cr.BasketLiquidation(1, synth.basket(n, seed=77))
    r, ref = route_both(obj, b, n)
    assert rel_to_max(cr.netflows(r), ref["psi"]) <= ROUTE_TOL
    r.close()
    cfmms = [cr.ProductTwoCoin([1e3, 1e4], 0.997, [1, 2]), cr.ProductTwoCoin([1e3, 1e2], 0.997, [2, 3]),
             cr.ProductTwoCoin([1e3, 2e4], 0.997, [1, 3])]
    r, ref = route_both(cr.BasketLiquidation(1, [0, 1e1, 1e2]), cfmms, 3)
    assert rel_to_max(cr.netflows(r), ref["psi"]) <= ROUTE_TOL
    assert cr.netflows(r)[0] > 0
    r.close()


@pytest.mark.parametrize("solver", ["scipy", "native"])
@pytest.mark.parametrize("shape", ["config3", "config5"])
def test_route_named_config_shapes(shape, solver):
    """BASELINE configs 3 and 5 in miniature (the reference has no router-level test with
    GeometricMeanTwoCoin or UniV3 pools): route! parity against the CPU restatement."""
    n = 64
    if shape == "config3":   # mixed ProductTwoCoin + GeometricMeanTwoCoin, LinearNonnegative arbitrage
        market = [synth.product_pools(20_000, n, seed=31), synth.geomean_pools(20_000, n, seed=32)]
        obj, v0 = cr.LinearNonnegative(synth.linear_prices(n, seed=31)), np.ones(n)
    else:                    # BoundedProduct (2-tick UniV3) pools, BasketLiquidation
        market = [synth.bounded_product_pools(40_000, n, seed=33)]
        obj, v0 = cr.BasketLiquidation(1, synth.basket(n, seed=33)), None
    r = cr.Router(obj, market, n)
    cr.route_(r, v=v0, solver=solver)
    ref = orc.route_oracle(oracle_objective(obj), oracle_poolset(r._batches, n), v0=v0, nthreads=8)
    assert rel_to_max(cr.netflows(r), ref["psi"]) <= ROUTE_TOL
    assert abs(r.info["f"] - ref["f"]) <= 1e-9 * max(1.0, abs(ref["f"]))
    if shape == "config5":
        Ψ = cr.netflows(r)
        assert Ψ[0] > 0                                   # something was received in the output token
        assert np.all(Ψ[1:] + obj.Δin[1:] >= -1e-3 - 1e-8 * np.max(np.abs(Ψ)))   # basket constraint Ψ₋ᵢ + Δin₋ᵢ ≥ 0
    r.close()


@pytest.mark.parametrize("solver", ["scipy", "native"])
def test_route_config5_interior_optimum(solver):
    """BASELINE config 5 on a market that is close to no-arbitrage (BoundedProduct pools quoted around
    one token price vector, 1 % noise): the BasketLiquidation dual has an INTERIOR optimum, route! has
    to discover the price vector from v = 1/n, and L-BFGS-B runs for tens of evaluations (on the
    arbitrage-rich random market it stops after 2, at the box corner).  300k pools, 256 tokens."""
    n, m = 256, 300_000
    market = [synth.bounded_product_pools(m, n, seed=1234, consistent=True)]
    obj = cr.BasketLiquidation(1, synth.basket(n, seed=1234))
    r = cr.Router(obj, market, n)
    cr.route_(r, solver=solver)
    ref = orc.route_oracle(oracle_objective(obj), oracle_poolset(market, n), v0=None, nthreads=_threads())
    assert r.info["funcalls"] >= 10 and ref["info"]["funcalls"] >= 10
    # (1) The KERNEL, isolated from the solver: the HIP sweep evaluated AT THE ORACLE'S v* returns the oracle's Ψ*
    #     to summation-order rounding (and its trades bit for bit).
    be = cr.DeviceBackend(n, market)
    psi_at_ref, _ = be.find_arb(ref["v"])
    D_at_ref, L_at_ref = be.trades()
    be.close()
    assert rel_to_max(psi_at_ref, ref["psi"]) <= REDUCE_TOL
    np.testing.assert_array_equal(D_at_ref, ref["Delta"])
    np.testing.assert_array_equal(L_at_ref, ref["Lambda"])
    # (2) At default tolerances neither side pins Ψ* to 1e-6 here: both stop on factr = 1e1 (relative decrease of the
    #     dual <= 10 eps) inside the rounding noise of the dual VALUE with a stationarity residual of ~1e-6 max|Ψ| left.
    #     Sanity bound only; the parity statement is test_route_parity_by_convergence below.
    Ψ, scale = cr.netflows(r), np.max(np.abs(ref["psi"]))
    dev = np.max(np.abs(Ψ[1:] - ref["psi"][1:])) / scale
    print(f"config5 interior ({solver}): device vs oracle at default tolerances {dev:.2e}, "
          f"|v*-v*ref|/|v*ref| {np.max(np.abs(r.v - ref['v']) / ref['v']):.2e}")
    assert dev <= 1e-5
    res_dev = np.max(np.abs(Ψ[1:] + obj.Δin[1:])) / scale
    res_ref = np.max(np.abs(ref["psi"][1:] + obj.Δin[1:])) / scale
    assert res_dev <= 1e-5 and res_ref <= 1e-5
    assert abs(r.info["f"] - ref["f"]) <= 1e-12 * max(1.0, abs(ref["f"]))
    # Ψ[0], the amount of the output token received, is the PRIMAL objective; its price sits on its bound, so
    # no stationarity condition pins it and it inherits the residuals amplified by the dual's curvature
    # (observed: 3.6e-6).  It is certified instead by the duality gap: primal value == dual value to 1e-5.
    assert abs(Ψ[0] - r.info["f"]) <= 1e-5 * abs(r.info["f"]) and abs(ref["psi"][0] - ref["f"]) <= 1e-5 * abs(ref["f"])
    assert abs(Ψ[0] - ref["psi"][0]) <= 1e-5 * scale
    lo = cr.lower_limit(obj)
    assert np.all(r.v[1:] > lo[1:] * 1e3)                 # interior: only the output token's price sits on its bound ...
    π = synth.token_price_vector(n, seed=1234)
    assert np.std(np.log(r.v / π)) < 0.02                 # ... and v* is the market's price vector up to scale
    assert Ψ[0] > 0 and np.all(Ψ[1:] + obj.Δin[1:] >= -1e-5 * scale)
    # trades at v*: every row equals a plain oracle sweep at the same prices
    Do, Lo, _, _ = oracle_sweep(market, n, r.v, nthreads=_threads())
    np.testing.assert_array_equal(r.Δs, Do)
    np.testing.assert_array_equal(r.Λs, Lo)
    r.close()


CONVERGED_TOL = 1e-8   # of max|Ψ|: the two converged netflow vectors (measured: 1e-14 .. 1e-12)


@pytest.mark.parametrize("solver", ["native", "scipy"])
@pytest.mark.parametrize("name", ["config5_300k", "config5_1m", "univ3_ticks_1m", "config4shard", "config3_1m"])
def test_route_parity_by_convergence(name, solver):
    """Route-level parity proven by convergence (src/router.jl:58,105, src/objectives.jl:92-129).
    route! on the HIP path and on the CPU restatement end up to 2e-6·max|Ψ| apart at the reference's tolerances on the
    interior optima (config 5, multi-tick UniV3) and up to 8e-7 on the arbitrage configs -- L-BFGS-B's stopping slack on
    each side, not a difference of the evaluations: tightening BOTH sides with the same gradient-only polish
    (cfmmrouter.jl_amd/router.py::polish_) makes the two Ψ* agree to <= 1e-8 (measured 1e-14 .. 1e-12).  factr -> 0 /
    pgtol -> 1e-9 do not do that: L-BFGS-B's line search compares dual VALUES, whose rounding noise (~1e-15 relative, a
    sum over 10^6 pools) hides the remaining decrease, and the run ends in ABNORMAL_TERMINATION_IN_LNSRCH at the same
    residual.  At default tolerances only a sanity bound is asserted."""
    if solver == "scipy" and name not in ("config5_300k", "config4shard"):
        pytest.skip("the SciPy-driven loop is exercised on two shapes; the library's own solver on all")
    n = 512 if name == "config4shard" else 256
    if name.startswith("config5"):
        market = [synth.bounded_product_pools(300_000 if name.endswith("300k") else 1_000_000, n, seed=1234, consistent=True)]
    elif name == "univ3_ticks_1m":
        market = [synth.univ3_ragged_pools(1_000_000, n, seed=1234)]
    elif name == "config4shard":
        market = [synth.product_pools(500_000, n, seed=1234)]
    else:
        market = [synth.product_pools(500_000, n, seed=1234), synth.geomean_pools(500_000, n, seed=1234)]
    basket = name.startswith("config5") or name.startswith("univ3")
    obj = cr.BasketLiquidation(1, synth.basket(n, seed=1234)) if basket else cr.LinearNonnegative(synth.linear_prices(n, seed=1234))
    out = route_converged(obj, market, n, v0=None if basket else np.ones(n), nthreads=_threads(), solver=solver)
    print(f"{name} ({solver}): default {out['default']:.2e} -> converged {out['converged']:.2e} (v* {out['v_converged']:.1e}); "
          f"route! ended {out['device_moved']:.1e} (device) / {out['oracle_moved']:.1e} (CPU restatement) from its own converged "
          f"point; evaluations {out['evaluations_device']} / {out['evaluations_oracle']}; residuals "
          f"{out['polish_device']['residual0']:.1e} -> {out['polish_device']['residual']:.1e} / "
          f"{out['polish_oracle']['residual0']:.1e} -> {out['polish_oracle']['residual']:.1e}")
    assert out["converged"] <= CONVERGED_TOL
    assert out["default"] <= 1e-5
    # each side's route! result lies within ITS OWN stopping slack of the common converged point
    assert out["default"] <= 1.5 * (out["device_moved"] + out["oracle_moved"]) + CONVERGED_TOL
    for pol in (out["polish_device"], out["polish_oracle"]):
        assert pol["residual"] <= 1e-9 * out["psi_scale"]


@pytest.mark.parametrize("kind", ["basket", "arb"])
def test_polish_inside_the_library_equals_the_python_mirror(kind):
    """cfmm_polish (one C-ABI call: what a Julia or C caller uses) and router.py::polish_ are the same iteration: from the
    same route! result both reach the same point (<= 1e-12 of max|Ψ|), with the optimality conditions satisfied to 1e-9
    and r.Δs / r.Λs describing the polished point."""
    n = 96
    if kind == "basket":
        market = [synth.bounded_product_pools(120_000, n, seed=91, consistent=True)]
        obj, v0 = cr.BasketLiquidation(1, synth.basket(n, seed=91)), None
    else:
        market = [synth.product_pools(100_000, n, seed=92), synth.geomean_pools(50_000, n, seed=93)]
        obj, v0 = cr.LinearNonnegative(synth.linear_prices(n, seed=92)), np.ones(n)
    res = {}
    for native in (True, False):
        r = cr.Router(obj, market, n)
        cr.route_(r, v=v0, solver="native")
        before = cr.netflows(r).copy()
        cr.polish_(r, native=native)
        res[native] = (cr.netflows(r).copy(), r.v.copy(), dict(r.info["polish"]), r.Δs.copy(), r.Λs.copy(), before)
        r.close()
    (pn, vn, infn, Dn, Ln, before), (pp, vp, infp, _, _, _) = res[True], res[False]
    scale = np.max(np.abs(pp))
    assert rel_to_max(pn, pp) <= 1e-12 and np.max(np.abs(vn - vp) / vp) <= 1e-12
    for inf in (infn, infp):
        assert inf["residual"] <= 1e-9 * scale and inf["residual"] < inf["residual0"] and inf["sweeps"] <= n + 1 + 8 * 7 + 2
    assert "native_seconds" in infn
    assert rel_to_max(before, pn) <= 1e-5              # polish moved the route!'s result by its stopping slack only
    Do, Lo, psi_o, _ = oracle_sweep(market, n, vn, nthreads=8)    # the trades left behind are those of a sweep at the polished prices
    assert rel_to_max(pn, psi_o) <= 1e-12
    np.testing.assert_allclose(Dn, Do, rtol=0, atol=1e-9 * np.max(Do))
    np.testing.assert_allclose(Ln, Lo, rtol=0, atol=1e-9 * np.max(Lo))
    lo = cr.lower_limit(obj)
    G = np.zeros(n)
    cr.grad_(G, obj, vn)
    G += pn
    on = vn <= lo
    assert np.max(np.abs(G[~on])) <= 1e-9 * scale and np.all(G[on] >= -1e-9 * scale)


@pytest.mark.parametrize("kind", ["arb", "basket", "mixed"])
def test_route_native_solver_one_call(kind):
    """cfmm_route: the whole of route! inside the library (own L-BFGS-B) vs the CPU restatement."""
    n = 48
    if kind == "mixed":
        market = [synth.product_pools(20_000, n, seed=3), synth.geomean_pools(10_000, n, seed=4),
                  synth.bounded_product_pools(10_000, n, seed=5)]
    else:
        market = [synth.product_pools(30_000, n, seed=9)]
    obj = cr.BasketLiquidation(2, synth.basket(n, seed=2)) if kind == "basket" else \
        cr.LinearNonnegative(synth.linear_prices(n, seed=3))
    v0 = None if kind == "basket" else np.ones(n)
    r = cr.Router(obj, market, n)
    cr.route_(r, v=v0, solver="native")
    ref = orc.route_oracle(oracle_objective(obj), oracle_poolset(r._batches, n), v0=v0)
    assert rel_to_max(cr.netflows(r), ref["psi"]) <= ROUTE_TOL
    assert abs(r.info["f"] - ref["f"]) <= 1e-9 * max(1.0, abs(ref["f"]))
    assert np.all(r.v >= cr.lower_limit(obj) - 1e-12)
    D, L = r.Δs, r.Λs                      # trades were materialised at v* by the same call
    assert D.shape == (sum(len(b) for b in market), 2) and np.all(D >= 0) and np.all(L >= 0)
    r.close()


@pytest.mark.parametrize("solver", ["scipy", "native"])
def test_no_fee_optimality_after_update_reserves(solver):
    """The reference's disabled check_opt_conditions_no_fee! (test/arb.jl:30-39) on the device path:
    route!, update_reserves!, then ∇φ(R) ∥ v[Ai] for every pool and nothing left to arbitrage."""
    n, m = 24, 20_000
    b = synth.product_pools(m, n, seed=42)
    b.γ[:] = 1.0
    k0 = b.R[:, 0] * b.R[:, 1]
    r = cr.Router(cr.LinearNonnegative(synth.linear_prices(n, seed=42)), b, n)
    cr.route_(r, solver=solver)
    cr.update_reserves_(r)
    R = r._batches[0].R
    vv = r.v[b.Ai - 1]
    cos = (R[:, 1] * vv[:, 0] + R[:, 0] * vv[:, 1]) / (np.hypot(R[:, 0], R[:, 1]) * np.hypot(vv[:, 0], vv[:, 1]))
    assert np.max(np.abs(cos - 1.0)) < 1e-9
    assert np.max(np.abs(R[:, 0] * R[:, 1] - k0) / k0) < 1e-9
    cr.route_(r, solver=solver)                       # the updated pools sit on the device
    assert np.max(np.abs(cr.netflows(r))) < 1e-2
    r.close()


# ---- BASELINE sizes: EVERY row against the threaded oracle (it sweeps 1M pools in well under a second) ----

NT = None


def _threads():
    global NT
    if NT is None:
        NT = max(1, int(orc.lib().oracle_max_threads()))
    return NT


def _size_independent_checks(b_list, D, L, psi, acc, v, n):
    """Properties that hold at any size: feasibility, one direction per pool, the device's Ψ / dual
    scalar are the sums of its own trades, no pool trades at a loss."""
    assert np.all(D >= 0) and np.all(L >= 0)
    assert not np.any((D[:, 0] > 0) & (D[:, 1] > 0)) and not np.any((L[:, 0] > 0) & (L[:, 1] > 0))
    Ai0 = np.concatenate([b.Ai for b in b_list]).astype(np.int32) - 1
    assert rel_to_max(psi, orc.netflows(D, L, Ai0, n)) <= REDUCE_TOL
    profit = (L * v[Ai0]).sum(1) - (D * v[Ai0]).sum(1)
    assert np.all(profit >= -1e-9) and abs(profit.sum() - acc) <= 1e-9 * abs(acc)


def test_full_size_product1m_all_rows():
    """1M ProductTwoCoin pools, 256 tokens (the north_star's ProductTwoCoin instance): all rows bit-exact."""
    m, n = 1_000_000, 256
    b = synth.product_pools(m, n, seed=1234)
    v = synth.sweep_prices(n, seed=1234)
    be = cr.DeviceBackend(n, [b])
    psi, acc = be.find_arb(v)
    D, L = be.trades()
    be.close()
    Do, Lo, psi_o, acc_o = oracle_sweep([b], n, v, nthreads=_threads())
    np.testing.assert_array_equal(D, Do)
    np.testing.assert_array_equal(L, Lo)
    assert rel_to_max(psi, psi_o) <= REDUCE_TOL and abs(acc - acc_o) <= REDUCE_TOL * abs(acc_o)
    _size_independent_checks([b], D, L, psi, acc, v, n)
    Rn = b.R + b.γ[:, None] * D - L
    assert np.all(Rn[:, 0] * Rn[:, 1] >= b.R[:, 0] * b.R[:, 1] * (1 - 1e-12))   # trading function not decreased


def test_full_size_config3_all_rows():
    """BASELINE config 3 at full size (500k ProductTwoCoin + 500k GeometricMeanTwoCoin, 256 tokens)."""
    n, h = 256, 500_000
    v = synth.sweep_prices(n, seed=1234)
    bp, bg = synth.product_pools(h, n, seed=1234), synth.geomean_pools(h, n, seed=1234)
    be = cr.DeviceBackend(n, [bp, bg])
    psi, acc = be.find_arb(v)
    D, L = be.trades()
    be.eval(v)                                                          # (tile direction alternates between sweeps)
    psi_f, acc_f = be.eval(v)                                           # the fused evaluation route! uses, same direction
    be.close()
    Do, Lo, psi_o, acc_o = oracle_sweep([bp, bg], n, v, nthreads=_threads())
    np.testing.assert_array_equal(D[:h], Do[:h])                       # ProductTwoCoin half: bit-exact
    np.testing.assert_array_equal(L[:h], Lo[:h])
    scale = np.maximum(bg.R.max(axis=1), 1.0)[:, None]                 # GeometricMean half: log-space forms,
    assert np.max(np.abs(D[h:] - Do[h:]) / scale) <= GEOM_RTOL         # <= 1e-12 of the reserve scale, every row
    assert np.max(np.abs(L[h:] - Lo[h:]) / scale) <= GEOM_RTOL
    assert np.array_equal(D[h:] > 0, Do[h:] > 0) and np.array_equal(L[h:] > 0, Lo[h:] > 0)   # same pools trade
    assert rel_to_max(psi, psi_o) <= REDUCE_TOL and abs(acc - acc_o) <= REDUCE_TOL * abs(acc_o)
    np.testing.assert_array_equal(psi_f, psi)
    assert acc_f == acc
    _size_independent_checks([bp, bg], D, L, psi, acc, v, n)
    Rn = bg.R + bg.γ[:, None] * D[h:] - L[h:]
    phi0 = bg.w[:, 0] * np.log(bg.R[:, 0]) + bg.w[:, 1] * np.log(bg.R[:, 1])
    phi1 = bg.w[:, 0] * np.log(Rn[:, 0]) + bg.w[:, 1] * np.log(Rn[:, 1])
    assert np.all(phi1 >= phi0 - 1e-11)                                # weighted invariant not decreased


def test_full_size_config4_shard_all_rows():
    """One GPU's share of BASELINE config 4: 500k ProductTwoCoin pools, 512 tokens."""
    m, n = 500_000, 512
    b = synth.product_pools(m, n, seed=1234)
    v = synth.sweep_prices(n, seed=1234)
    be = cr.DeviceBackend(n, [b])
    psi, acc = be.find_arb(v)
    D, L = be.trades()
    be.close()
    Do, Lo, psi_o, acc_o = oracle_sweep([b], n, v, nthreads=_threads())
    np.testing.assert_array_equal(D, Do)
    np.testing.assert_array_equal(L, Lo)
    assert rel_to_max(psi, psi_o) <= REDUCE_TOL and abs(acc - acc_o) <= REDUCE_TOL * abs(acc_o)
    _size_independent_checks([b], D, L, psi, acc, v, n)


@pytest.mark.parametrize("consistent", [True, False])
def test_full_size_config5_all_rows(consistent):
    """BASELINE config 5 at full size: 1M BoundedProduct pools (2-tick UniV3), all rows bit-exact -- on the market
    bench.py's config5 times (pools quoted around one token price vector: consistent = True) and on the arbitrage-rich
    one with independent random prices (config5corner), at prices off the no-arbitrage manifold."""
    n = 256
    v = synth.sweep_prices(n, seed=1234)
    bu = synth.bounded_product_pools(1_000_000, n, seed=1234, consistent=consistent)
    be = cr.DeviceBackend(n, [bu])
    psi, acc = be.find_arb(v)
    D, L = be.trades()
    be.close()
    Do, Lo, psi_o, acc_o = oracle_sweep([bu], n, v, nthreads=_threads())
    np.testing.assert_array_equal(D, Do)
    np.testing.assert_array_equal(L, Lo)
    assert rel_to_max(psi, psi_o) <= REDUCE_TOL and abs(acc - acc_o) <= REDUCE_TOL * abs(acc_o)
    _size_independent_checks([bu], D, L, psi, acc, v, n)


def test_full_size_univ3_ticks_all_rows():
    """The multi-tick workload of bench.py (--workload univ3_ticks) at full size: 1M UniV3 pools with ragged ladders of
    2..64 initialised ticks (17M ticks), swept a few per cent off the price vector they are quoted around, so that most
    pools walk through several ticks (src/cfmms.jl:339-395): every trade row bit-exact against the CPU restatement."""
    n = 256
    bu = synth.univ3_ragged_pools(1_000_000, n, seed=1234)
    v = synth.sweep_prices(n, seed=1234) * synth.token_price_vector(n, seed=1234)
    visited = synth.univ3_ticks_visited(bu, v)
    assert np.mean(visited > 1) > 0.5 and visited.max() >= 16          # a real multi-tick walk, not the 2-tick degenerate case
    be = cr.DeviceBackend(n, [bu])
    psi, acc = be.find_arb(v)
    D, L = be.trades()
    psi_f, acc_f = be.eval(v)
    psi_f, acc_f = be.eval(v)                                           # (same tile direction as the materialising sweep)
    be.close()
    Do, Lo, psi_o, acc_o = oracle_sweep([bu], n, v, nthreads=_threads())
    np.testing.assert_array_equal(D, Do)
    np.testing.assert_array_equal(L, Lo)
    assert rel_to_max(psi, psi_o) <= REDUCE_TOL and abs(acc - acc_o) <= REDUCE_TOL * abs(acc_o)
    np.testing.assert_array_equal(psi_f, psi)
    assert np.all(D[visited == 0] == 0) and np.all(L[visited == 0] == 0)   # pools inside their no-arbitrage band
    assert np.count_nonzero(D.sum(axis=1) > 0) > 0.9 * np.count_nonzero(visited > 0)


# ---- the reference's optimality predicate (test/cfmms.jl:3-22) on DEVICE trades ---------------------

SQRT_EPS = math.sqrt(np.finfo(float).eps)


def _isapprox(a, b):   # Julia's isapprox with its default rtol = sqrt(eps)
    return abs(a - b) <= SQRT_EPS * max(abs(a), abs(b))


def _optimality_conditions_met(c, Δ, Λ, cfmm):
    """test_optimality_conditions_met(c, Δ, Λ, cfmm) -- test/cfmms.jl:3-22, with the host mirror's ϕ / ∇ϕ!"""
    R, γ = cfmm.R, cfmm.γ
    Rp = R + γ * Δ - Λ
    pfeas = np.all(Δ >= 0) and np.all(Λ >= 0)
    ϕR, ϕRp = cr.ϕ(cfmm), cr.ϕ(cfmm, R=Rp)
    g = np.zeros(2)
    cr.ϕ_grad_(g, cfmm, R=Rp)
    cfmm_sat = _isapprox(ϕR, ϕRp) and ϕRp >= ϕR - SQRT_EPS
    opt = max(γ * g[i] / c[i] for i in range(2)) <= min(g[i] / c[i] for i in range(2)) + SQRT_EPS
    return pfeas and cfmm_sat and opt


def test_reference_optimality_predicate_on_device_trades():
    """test/cfmms.jl:63-67, :92-96 (product) and :100-107 (geo mean): 3 reserves x 3 fees x 3 price
    vectors (x 3 weights), find_arb! on the MI355X, the reference's predicate on what comes back."""
    rng = np.random.default_rng(1234)
    γs, Rs, νs = rng.random(3), rng.random((3, 2)) * 10, rng.random((3, 2))
    ws = [np.array([w1, 1 - w1]) for w1 in rng.random(3)]
    Δ, Λ = np.zeros(2), np.zeros(2)
    for R in Rs:
        for γ in γs:
            for ν in νs:
                cfmm = cr.ProductTwoCoin(R, γ, [1, 2])
                cr.find_arb_(Δ, Λ, cfmm, ν)
                assert _optimality_conditions_met(ν, Δ, Λ, cfmm)
                for w in ws:
                    cfmm = cr.GeometricMeanTwoCoin(R, w, γ, [1, 2])
                    cr.find_arb_(Δ, Λ, cfmm, ν)
                    assert _optimality_conditions_met(ν, Δ, Λ, cfmm)


def test_reference_optimality_predicate_on_a_swept_market():
    """The same predicate on every pool of a 20k-pool mixed sweep (batch path, fused launch)."""
    n = 32
    bp, bg = synth.product_pools(10_000, n, seed=77), synth.geomean_pools(10_000, n, seed=78)
    v = synth.sweep_prices(n, seed=79, spread=0.5)
    be = cr.DeviceBackend(n, [bp, bg])
    be.find_arb(v)
    D, L = be.trades()
    be.close()
    for k in range(0, 10_000, 7):
        assert _optimality_conditions_met(v[bp.Ai[k] - 1], D[k], L[k], bp[k])
        assert _optimality_conditions_met(v[bg.Ai[k] - 1], D[10_000 + k], L[10_000 + k], bg[k])


@pytest.mark.parametrize("seed", range(int(os.environ.get("CFMM_FUZZ_SEEDS", "12"))))
def test_randomized_markets_differential(seed):
    """Random market shapes / families / token counts / launch options: device vs oracle."""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.choice([2, 3, 17, 64, 300, 1500, 9000]))
    segs = []
    for _ in range(int(rng.integers(1, 6))):
        kind = int(rng.integers(0, 4))
        m = int(rng.choice([1, 63, 64, 65, 1000, 5000, 40_000]))
        sd = int(rng.integers(1, 10_000))
        if kind == 0:
            segs.append(synth.product_pools(m, n, seed=sd))
        elif kind == 1:
            segs.append(synth.geomean_pools(m, n, seed=sd))
        elif kind == 2:
            segs.append(synth.bounded_product_pools(m, n, seed=sd))
        else:
            segs.append(synth.univ3_pools(min(m, 5000), n, int(rng.integers(1, 30)), seed=sd))
    v = synth.sweep_prices(n, seed=seed, spread=float(rng.choice([0.01, 0.3, 2.0])))
    opts = {}
    if rng.random() < 0.5:
        opts["fuse_segments"] = int(rng.integers(0, 2))
    if rng.random() < 0.3:
        opts["block"] = int(rng.choice([256, 512, 1024]))
    if rng.random() < 0.3:
        opts["max_grid"] = int(rng.choice([1, 3, 64]))
    if rng.random() < 0.3:
        opts["bin_copies"] = int(rng.choice([1, 2]))
    mat = bool(rng.random() < 0.7)
    D, L, psi, acc = device_sweep(segs, n, v, materialize=mat, **opts)
    Do, Lo, psio, acco = oracle_sweep(segs, n, v)
    if mat:
        off = 0
        for b in segs:
            sl = slice(off, off + len(b))
            if b.kind == 1:      # geometric mean: log-space forms
                scale = np.max(b.R, axis=1, keepdims=True)
                assert np.max(np.abs(D[sl] - Do[sl]) / scale) <= GEOM_RTOL
                assert np.max(np.abs(L[sl] - Lo[sl]) / scale) <= GEOM_RTOL
            else:                # product / univ3: bit-exact
                np.testing.assert_array_equal(D[sl], Do[sl])
                np.testing.assert_array_equal(L[sl], Lo[sl])
            off += len(b)
    assert rel_to_max(psi, psio) <= 1e-10
    assert abs(acc - acco) <= 1e-9 * max(abs(acco), 1.0)


def test_context_lifecycle_stress():
    """create / upload / sweep / destroy many contexts: no leak-induced failure, results stay identical."""
    b = synth.product_pools(2000, 8, seed=1)
    v = synth.sweep_prices(8, seed=1)
    first = None
    for k in range(300):
        be = cr.DeviceBackend(8, [b])
        psi, acc = be.eval(v) if k % 2 else be.find_arb(v)
        be.ctx.clear()
        be.ctx.add_product(b.R, b.γ, (b.Ai - 1).astype(np.int32))      # pools can be replaced in place
        psi2, _ = be.eval(v)
        be.close()
        np.testing.assert_array_equal(psi, psi2)
        if first is None:
            first = psi
        np.testing.assert_array_equal(psi, first)


def test_eight_shards_summed_equal_the_unsharded_sweep():
    """shard-correctness on one GPU -- config 4's layout in miniature (contiguous blocks
    of every family per rank): the 8 shards' {Ψ, acc}, summed in rank order, equal the whole market's."""
    from cfmmrouter_amd.dist import shard_batches, shard_range
    n = 512
    market = [synth.product_pools(400_003, n, seed=4), synth.geomean_pools(50_001, n, seed=5),
              synth.univ3_pools(20_005, n, 4, seed=6)]
    v = synth.sweep_prices(n, seed=4)
    be = cr.DeviceBackend(n, market)
    psi_all, acc_all = be.find_arb(v)
    D_all, L_all = be.trades()
    be.close()
    psi_sum, acc_sum, rows = np.zeros(n), 0.0, 0
    for rank in range(8):
        shard = shard_batches(market, rank, 8)
        be = cr.DeviceBackend(n, shard)
        psi, acc = be.find_arb(v)
        D, L = be.trades()
        be.close()
        psi_sum += psi
        acc_sum += acc
        off_all = 0
        off = 0
        for b, full in zip(shard, market):          # every shard row is the unsharded row, bit for bit
            lo, hi = shard_range(len(full), rank, 8)
            np.testing.assert_array_equal(D[off:off + len(b)], D_all[off_all + lo:off_all + hi])
            np.testing.assert_array_equal(L[off:off + len(b)], L_all[off_all + lo:off_all + hi])
            off += len(b)
            off_all += len(full)
        rows += off
    assert rows == sum(len(b) for b in market)
    assert rel_to_max(psi_sum, psi_all) <= 1e-12
    assert abs(acc_sum - acc_all) <= 1e-11 * abs(acc_all)


def test_small_market_fuzz_against_the_oracle():
    """150 seeded random small routers -- 1 .. 5 000 pools, 2 .. 48 tokens, one to three pool families in random order and sizes
    (so: single-block launches, one-tile-per-block launches and fused multi-family launches of odd grids), fees on both sides
    of 1, reserves over twelve decades, host-pointer and fused evaluations -- every ProductTwoCoin / UniV3 row bit-equal to the
    CPU restatement of the reference, GeometricMean within 1e-12 of the reserve scale, psi and the dual value within 1e-12."""
    rng = np.random.default_rng(20261001)
    for case in range(150):
        n = int(rng.integers(2, 49))
        fams = list(rng.permutation(["product", "geomean", "univ3"])[: int(rng.integers(1, 4))])
        batches = []
        for f in fams:
            m = int(rng.choice([1, 2, 63, 64, 65, 511, 512, 513, 1023, 1025, 2047, 2048, 2049, int(rng.integers(1, 5000))]))
            seed = int(rng.integers(1, 1 << 30))
            if f == "product":
                b = synth.product_pools(m, n, seed=seed)
                b.R[:] *= 10.0 ** rng.uniform(-6, 6, size=(m, 1))
                b.γ[::5] = 1.0 + 1e-3                                      # fee above 1: both directions can trade
            elif f == "geomean":
                b = synth.geomean_pools(m, n, seed=seed)
                b.R[:] *= 10.0 ** rng.uniform(-3, 3, size=(m, 1))
            else:
                b = synth.univ3_pools(m, n, int(rng.integers(2, 9)), seed=seed)
            batches.append(b)
        v = synth.sweep_prices(n, seed=int(rng.integers(1, 1 << 30)), spread=float(rng.uniform(0.01, 0.8)))
        Do, Lo, psi_o, acc_o = oracle_sweep(batches, n, v)
        be = cr.DeviceBackend(n, batches)
        try:
            psi, acc = be.find_arb(v)
            D, L = be.trades()
            psi_f, acc_f = be.eval(v)
        finally:
            be.close()
        off = 0
        for f, b in zip(fams, batches):
            sl = slice(off, off + len(b))
            if f == "geomean":
                scale = np.maximum(b.R.max(axis=1), 1.0)[:, None]
                assert np.max(np.abs(D[sl] - Do[sl]) / scale) <= 1e-12 and np.max(np.abs(L[sl] - Lo[sl]) / scale) <= 1e-12, (case, f)
            else:
                assert np.array_equal(D[sl], Do[sl]) and np.array_equal(L[sl], Lo[sl]), (case, f, len(b), n)
            off += len(b)
        scale = max(float(np.max(np.abs(psi_o))), 1e-300)
        assert np.max(np.abs(psi - psi_o)) <= 1e-12 * scale and np.max(np.abs(psi_f - psi_o)) <= 1e-12 * scale, (case, fams)
        assert abs(acc - acc_o) <= 1e-12 * max(abs(acc_o), scale) and abs(acc_f - acc_o) <= 1e-12 * max(abs(acc_o), scale), (case, fams)
