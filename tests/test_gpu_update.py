"""update_reserves!(r) on the device (cfmm_update_reserves; src/router.jl:127-132, the update
prescribed by src/cfmms.jl:26-31) -- the reference's own version is unimplemented and its test
disabled (test/arb.jl:30-39), so what is pinned here is the mathematics: the pool after the update IS
R + γΔ − Λ, nothing is left to arbitrage at the same prices, and sequential routing works without any
per-pool host traffic."""
import numpy as np
import pytest

import cfmmrouter_amd as cr
from cfmmrouter_amd import synth
from helpers import oracle_sweep, rel_to_max

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("device", [0, [0, 0, 0]])
def test_two_coin_update_is_exact_and_leaves_no_arbitrage(device):
    n = 40
    bp, bg = synth.product_pools(50_001, n, seed=1), synth.geomean_pools(30_000, n, seed=2)
    v = synth.sweep_prices(n, seed=3, spread=0.4)
    be = cr.DeviceBackend(n, [bp, bg], device=device)
    try:
        be.find_arb(v)
        D, L = be.trades()
        be.ctx.update_reserves()
        Rp, Rg = be.ctx.reserves(0, len(bp)), be.ctx.reserves(1, len(bg))
        np.testing.assert_array_equal(Rp, (bp.R + bp.γ[:, None] * D[:len(bp)]) - L[:len(bp)])    # same operation order: same bits
        np.testing.assert_array_equal(Rg, (bg.R + bg.γ[:, None] * D[len(bp):]) - L[len(bp):])
        with pytest.raises(RuntimeError):                 # the trades were consumed
            be.trades()
        with pytest.raises(RuntimeError):
            be.ctx.update_reserves()
        psi, acc = be.find_arb(v)                         # nothing left at the same prices
        D2, L2 = be.trades()
        scale = np.maximum(np.concatenate([Rp, Rg]).max(axis=1), 1.0)[:, None]
        assert np.max(D2 / scale) <= 1e-9 and np.max(L2 / scale) <= 1e-9
        # ... and the device's updated store behaves as a freshly uploaded one with those reserves
        v2 = synth.sweep_prices(n, seed=4, spread=0.4)
        psi_dev, _ = be.find_arb(v2)
        bp2 = cr.PoolBatch(bp.kind, R=Rp, γ=bp.γ, Ai=bp.Ai)
        bg2 = cr.PoolBatch(bg.kind, R=Rg, w=bg.w, γ=bg.γ, Ai=bg.Ai)
        Do, Lo, psi_o, _ = oracle_sweep([bp2, bg2], n, v2, nthreads=8)
        assert rel_to_max(psi_dev, psi_o) <= 1e-12
        np.testing.assert_array_equal(be.trades()[0][:len(bp)], Do[:len(bp)])
    finally:
        be.close()


def _tick_reserves(cp, lt, liq):
    """Real reserves (Σ over ticks of R₁, R₂) of one UniV3 pool at price cp -- compute_at_tick,
    src/cfmms.jl:294-313, in numpy."""
    nt = lt.size
    ct = int(np.count_nonzero(lt >= cp))
    tot = np.zeros(2)
    for idx in range(1, nt + 1):
        k, pplus = liq[idx - 1], lt[idx - 1]
        pminus = lt[idx] if idx < nt else 0.0
        p = pplus if idx > ct else (pminus if idx < ct else cp)
        if k == 0:
            continue
        tot[0] += np.sqrt(k / p) - np.sqrt(k / pplus) if p > 0 else 0.0
        tot[1] += np.sqrt(k * p) - np.sqrt(k * pminus)
    return tot


@pytest.mark.parametrize("ticks", [2, 9])
def test_univ3_update_moves_the_price_and_conserves_tick_reserves(ticks):
    n, m = 24, 6000
    b = synth.bounded_product_pools(m, n, seed=5, consistent=True, noise=0.05) if ticks == 2 else \
        synth.univ3_pools(m, n, ticks, seed=6)
    v = synth.token_price_vector(n, seed=5) * np.exp(0.05 * (2 * synth.uniform(7, 1, n) - 1)) if ticks == 2 else \
        synth.sweep_prices(n, seed=8, spread=0.3)
    r = cr.Router(cr.LinearNonnegative(np.ones(n)), [b], n)
    try:
        cr.find_arb_(r, v)
        D, L = r.Δs.copy(), r.Λs.copy()
        cp0 = b.current_price.copy()
        traded = (D.sum(1) > 0)
        assert traded.sum() > m // 10 and (~traded).sum() > 0
        cr.update_reserves_(r)                            # also refreshes b.current_price
        cp1 = b.current_price
        lt, lq = b.lower_ticks.reshape(m, ticks), b.liquidity.reshape(m, ticks)
        for i in np.flatnonzero(~traded & (L.sum(1) == 0))[:200]:
            # a pool that did not trade keeps its price, or moved through ticks WITHOUT liquidity (any price in
            # such a gap describes the same pool): its real reserves are untouched either way
            np.testing.assert_allclose(_tick_reserves(cp1[i], lt[i], lq[i]), _tick_reserves(cp0[i], lt[i], lq[i]),
                                       rtol=1e-12, atol=0)
        checked = 0
        for i in np.flatnonzero(traded)[:400]:
            before, after = _tick_reserves(cp0[i], lt[i], lq[i]), _tick_reserves(cp1[i], lt[i], lq[i])
            want = before + b.γ[i] * D[i] - L[i]          # the pool the routing problem prescribes
            if ticks > 2 and cp1[i] <= lt[i, -1] and lq[i, -1] > 0:
                continue                                   # last tick reaches price 0: R₁ unbounded, skip the identity there
            assert np.allclose(after, want, rtol=1e-9, atol=1e-9 * max(1.0, np.max(np.abs(before))))
            checked += 1
        assert checked > 100
        cr.find_arb_(r, v)                                # no pool has anything left at the same prices
        assert np.max(r.Δs) <= 1e-9 * np.max(D) and np.max(r.Λs) <= 1e-9 * np.max(L)
        # the updated store equals a fresh upload at the new prices (bit for bit)
        v2 = v * np.exp(0.1 * (2 * synth.uniform(9, 2, n) - 1))
        cr.find_arb_(r, v2)
        Do, Lo, psi_o, _ = oracle_sweep([b], n, v2, nthreads=8)
        np.testing.assert_array_equal(r.Δs, Do)
        np.testing.assert_array_equal(r.Λs, Lo)
    finally:
        r.close()


def test_sequential_routing_full_size_without_host_trade_traffic():
    """route! -> update_reserves! -> route! on 1M no-fee ProductTwoCoin pools: the second route! finds the
    market already cleared (test/arb.jl:30-39's check_opt_conditions_no_fee! shape), and update_reserves!
    moved nothing per pool over PCIe (sync_host=False)."""
    n, m = 256, 1_000_000
    b = synth.product_pools(m, n, seed=42)
    b.γ[:] = 1.0
    k0 = b.R[:, 0] * b.R[:, 1]
    r = cr.Router(cr.LinearNonnegative(synth.linear_prices(n, seed=42)), b, n)
    try:
        cr.route_(r, v=np.ones(n), solver="native")
        first = np.max(np.abs(cr.netflows(r)))
        cr.update_reserves_(r, sync_host=False)
        cr.route_(r, v=r.v.copy(), solver="native")
        assert np.max(np.abs(cr.netflows(r))) <= 1e-6 * first     # nothing left to arbitrage
        R = r._backend.ctx.reserves(0, m)
        vv = r.v[b.Ai - 1]
        cos = (R[:, 1] * vv[:, 0] + R[:, 0] * vv[:, 1]) / (np.hypot(R[:, 0], R[:, 1]) * np.hypot(vv[:, 0], vv[:, 1]))
        assert np.max(np.abs(cos - 1.0)) < 1e-9                    # ∇φ(R) ∥ v[Ai] for every pool
        assert np.max(np.abs(R[:, 0] * R[:, 1] - k0) / k0) < 1e-9   # the invariant is unchanged
    finally:
        r.close()


def test_nan_prices_propagate_through_device_pointer_sweeps():
    """cfmm_sweep_dev takes v from device memory and cannot validate it; a NaN price must come
    back as NaN trades / netflows (the reference propagates NaN), never as a plausible-looking zero."""
    import torch
    n = 12
    batches = [synth.product_pools(3000, n, seed=1), synth.geomean_pools(2000, n, seed=2),
               synth.univ3_pools(1000, n, 4, seed=3)]
    be = cr.DeviceBackend(n, batches)
    try:
        v = synth.sweep_prices(n, seed=4)
        v[5] = np.nan
        vt = torch.from_numpy(v).cuda()
        out = torch.zeros(n + 1, dtype=torch.float64, device="cuda")
        be.ctx.sweep_dev(vt.data_ptr(), out.data_ptr(), True)        # round 5: the very first sweep propagates it per pool (the
        torch.cuda.synchronize()                                     # launch carries both arithmetics; round 4 refused it once)
        o = out.cpu().numpy()
        assert np.isnan(o[5]) and np.isnan(o[n])
        D, L = be.trades()
        Ai = np.concatenate([b.Ai for b in batches])
        touches = np.any(Ai == 6, axis=1)                 # 1-based token 6 == index 5
        assert np.all(np.isnan(D[touches]).any(axis=1) | np.isnan(L[touches]).any(axis=1))
        assert not np.isnan(D[~touches]).any() and not np.isnan(L[~touches]).any()
        with pytest.raises(cr.ArgumentError):             # host-pointer calls validate v instead
            be.find_arb(v)
    finally:
        be.close()
