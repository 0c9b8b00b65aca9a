"""GPU tests of the in-launch row fold (option "inline_fold") and the host-visible completion flag
(option "host_flag"): the partial rows of a sweep are published with write-through stores and
folded by extra blocks of the SAME launch (src/router.jl:81-83, :98-100 summed over blocks).

The separate-launch fold (reduce_partials) sums the rows in exactly the same order, so the two
forms must agree BIT FOR BIT; a stale or torn row read inside the launch shows up as a mismatch.
Per MI355X_MICROARCH.md the hand-off is exercised back to back (the fold blocks' caches are warm
with the previous sweep's rows), with different prices per sweep and with uneven per-block load.
"""
import numpy as np
import pytest

import cfmmrouter_amd as cr
from cfmmrouter_amd import synth
from helpers import oracle_sweep, rel_to_max

pytestmark = pytest.mark.gpu


def _pair(n, batches, mode=1, **opts):
    a, b = cr.DeviceBackend(n, batches), cr.DeviceBackend(n, batches)
    a.ctx.set_option("inline_fold", mode)
    b.ctx.set_option("inline_fold", 0)
    for k, v in opts.items():
        a.ctx.set_option(k, v)
        b.ctx.set_option(k, v)
    return a, b


@pytest.mark.parametrize("shape", ["config2", "config3mini", "config4mini", "bounded", "tiny", "n1024"])
def test_inline_fold_bitwise_equals_separate_fold(shape):
    if shape == "config2":
        n, batches = 64, [synth.product_pools(100_000, 64, seed=3)]
    elif shape == "config3mini":
        n, batches = 256, [synth.product_pools(200_000, 256, seed=3), synth.geomean_pools(150_001, 256, seed=4)]
    elif shape == "config4mini":
        n, batches = 512, [synth.product_pools(500_000, 512, seed=5)]
    elif shape == "bounded":
        n, batches = 256, [synth.bounded_product_pools(120_000, 256, seed=6)]
    elif shape == "tiny":
        n, batches = 2, [synth.product_pools(3, 2, seed=7)]
    else:
        n, batches = 1024, [synth.product_pools(70_000, 1024, seed=8)]
    a, b = _pair(n, batches)
    try:
        rng = np.random.default_rng(11)
        for it in range(40):   # back to back: the fold blocks re-read row addresses they read one sweep ago
            v = synth.sweep_prices(n, seed=100 + it) * rng.uniform(0.5, 2.0)
            for mat in (False, True):
                pa = (a.find_arb if mat else a.eval)(v)
                pb = (b.find_arb if mat else b.eval)(v)
                if shape == "n1024":   # one LDS bin copy shared by the wavefronts: the sum order is not fixed
                    assert rel_to_max(pa[0], pb[0]) <= 1e-14 and abs(pa[1] - pb[1]) <= 1e-12 * abs(pb[1])
                else:
                    np.testing.assert_array_equal(pa[0], pb[0])
                    assert pa[1] == pb[1]
        D, L, psi, acc = oracle_sweep(batches, n, v)
        assert rel_to_max(pa[0], psi) <= 1e-12
        Da, La = a.trades()
        if shape != "config3mini":   # geomean trades differ from the oracle by a few ulp (tested elsewhere)
            np.testing.assert_array_equal(Da, D)
            np.testing.assert_array_equal(La, L)
    finally:
        a.close()
        b.close()


def test_inline_fold_under_uneven_load():
    """A deep-walk UniV3 segment next to a ProductTwoCoin segment: blocks of one launch finish at very
    different times, and a second context streams on another stream meanwhile."""
    n = 128
    batches = [synth.product_pools(300_000, n, seed=21), synth.univ3_pools(40_000, n, 40, seed=22)]
    a, b = _pair(n, batches)
    noise = cr.DeviceBackend(256, [synth.product_pools(1_000_000, 256, seed=23)])
    try:
        import torch
        vt = torch.from_numpy(synth.sweep_prices(256, seed=1)).cuda()
        ot = torch.zeros(257, dtype=torch.float64, device="cuda")
        for it in range(25):
            for _ in range(4):   # asynchronous streaming load on the context's own stream
                noise.ctx.sweep_dev(vt.data_ptr(), ot.data_ptr(), True)
            v = synth.sweep_prices(n, seed=300 + it)
            pa, pb = a.eval(v), b.eval(v)
            np.testing.assert_array_equal(pa[0], pb[0])
            assert pa[1] == pb[1]
        torch.cuda.synchronize()
    finally:
        a.close()
        b.close()
        noise.close()


def test_device_resident_sweeps_with_inline_fold():
    """cfmm_sweep_dev (the bench's timed path): many launches in flight on one stream, each leaving the
    arrival counters zero for the next."""
    import torch
    n = 256
    batches = [synth.product_pools(500_000, n, seed=31), synth.geomean_pools(500_000, n, seed=32)]
    a, b = _pair(n, batches)
    try:
        st = torch.cuda.Stream()
        a.ctx.set_stream(st.cuda_stream)
        b.ctx.set_stream(st.cuda_stream)
        vs = [torch.from_numpy(synth.sweep_prices(n, seed=400 + k)).cuda() for k in range(8)]
        oa = [torch.zeros(n + 1, dtype=torch.float64, device="cuda") for _ in vs]
        ob = [torch.zeros(n + 1, dtype=torch.float64, device="cuda") for _ in vs]
        torch.cuda.synchronize()
        for rep in range(6):
            for k, vt in enumerate(vs):
                a.ctx.sweep_dev(vt.data_ptr(), oa[k].data_ptr(), rep % 2 == 0)
            for k, vt in enumerate(vs):
                b.ctx.sweep_dev(vt.data_ptr(), ob[k].data_ptr(), rep % 2 == 0)
            st.synchronize()
            for k in range(len(vs)):
                assert torch.equal(oa[k], ob[k])
        a.ctx.reset_stream()
        b.ctx.reset_stream()
    finally:
        a.close()
        b.close()


@pytest.mark.parametrize("inline", [0, 1])
def test_host_flag_and_stream_wait_agree(inline):
    """Zero-copy host-pointer sweeps end when the last fold block raises a flag in mapped host memory
    (option "host_flag", default on); waiting for the stream instead must give the same bits."""
    n = 256
    batches = [synth.product_pools(250_000, n, seed=41), synth.geomean_pools(250_000, n, seed=42)]
    a = cr.DeviceBackend(n, batches)
    b = cr.DeviceBackend(n, batches)
    a.ctx.set_option("inline_fold", inline)
    b.ctx.set_option("inline_fold", inline)
    b.ctx.set_option("host_flag", 0)
    try:
        assert a.ctx.get_option("host_flag") == 1
        for it in range(200):
            v = synth.sweep_prices(n, seed=500 + it)
            pa, pb = a.eval(v), b.eval(v)
            np.testing.assert_array_equal(pa[0], pb[0])
            assert pa[1] == pb[1]
        # trades fetched right after a flagged sweep wait for the kernel itself
        pa = a.find_arb(v)
        pb = b.find_arb(v)
        Da, La = a.trades()
        Db, Lb = b.trades()
        np.testing.assert_array_equal(Da, Db)
        np.testing.assert_array_equal(La, Lb)
    finally:
        a.close()
        b.close()


def test_empty_segment_through_the_abi_is_ignored():
    """an m == 0 batch added through the C ABI must not leave an unwritten partial row."""
    n = 16
    for fuse, inline in ((0, 0), (0, 1), (1, 1)):
        be = cr.DeviceBackend(n, [])
        be.ctx.set_option("fuse_segments", fuse)
        be.ctx.set_option("inline_fold", inline)
        try:
            pb = synth.product_pools(1000, n, seed=51)
            empty = pb.slice(0, 0)
            from cfmmrouter_amd.cfmms import _upload
            _upload(be.ctx, empty)
            _upload(be.ctx, pb)
            _upload(be.ctx, empty)
            assert be.ctx.pool_count == 1000 and len(be.ctx.segments()) == 1
            v = synth.sweep_prices(n, seed=52)
            psi, acc = be.find_arb(v)
            D, L, psi_o, acc_o = oracle_sweep([pb], n, v)
            assert rel_to_max(psi, psi_o) <= 1e-12
            Dd, Ld = be.trades()
            np.testing.assert_array_equal(Dd, D)
        finally:
            be.close()


@pytest.mark.parametrize("opts", [dict(xcd_map=0), dict(xcd_map=1), dict(wave_split=1), dict(wave_split=1, inline_fold=1)])
@pytest.mark.parametrize("families", ["pg", "pgu", "pgup"])
def test_fused_launch_block_maps_agree_with_oracle(opts, families):
    """The fused multi-family launch under its block -> segment maps (plain b % nseg, XCD-aware, wavefronts
    dealt to the families): same trades bit for bit, same Ψ to summation-order rounding."""
    n = 200
    batches = [synth.product_pools(300_000, n, seed=61), synth.geomean_pools(200_001, n, seed=62)]
    if "u" in families:
        batches.append(synth.univ3_pools(50_000, n, 5, seed=63))
    if families == "pgup":
        batches.append(synth.product_pools(70_000, n, seed=64))
    v = synth.sweep_prices(n, seed=65)
    be = cr.DeviceBackend(n, batches)
    for k, val in opts.items():
        be.ctx.set_option(k, val)
    try:
        psi, acc = be.find_arb(v)
        D, L = be.trades()
        psi_b, acc_b = be.eval(v)     # consecutive sweeps walk the tiles in alternating directions (option "alternate"):
        psi_f, acc_f = be.eval(v)     # the next one is backwards, the one after it forwards again
        Do, Lo, psi_o, acc_o = oracle_sweep(batches, n, v, nthreads=8)
        assert rel_to_max(psi, psi_o) <= 1e-12 and abs(acc - acc_o) <= 1e-12 * abs(acc_o)
        np.testing.assert_array_equal(psi_f, psi)                         # same direction: fused == materialising, bit for bit
        assert acc_f == acc
        assert rel_to_max(psi_b, psi) <= 1e-14 and abs(acc_b - acc) <= 1e-13 * abs(acc)   # other direction: rounding only
        g0, g1 = 300_000, 500_001
        np.testing.assert_array_equal(D[:g0], Do[:g0])
        np.testing.assert_array_equal(L[:g0], Lo[:g0])
        np.testing.assert_array_equal(D[g1:], Do[g1:])
        np.testing.assert_array_equal(L[g1:], Lo[g1:])
        scale = np.maximum(batches[1].R.max(axis=1), 1.0)[:, None]
        assert np.max(np.abs(D[g0:g1] - Do[g0:g1]) / scale) <= 1e-12
    finally:
        be.close()


def test_determinism_claim_is_conditional_on_private_bin_copies():
    """sweeps are bit-reproducible when every wavefront owns a private LDS bin copy (n_tokens up to
    ~600 with 1024-thread blocks); above that the wavefronts of a block share one copy and the cross-wavefront
    ds_add_f64 order is not fixed -- results then agree to summation-order rounding only."""
    m = 200_000
    for n, exact in ((256, True), (4096, False)):
        be = cr.DeviceBackend(n, [synth.product_pools(m, n, seed=91)])
        try:
            v = synth.sweep_prices(n, seed=92)
            runs = [be.eval(v) for _ in range(6)]
            for psi, acc in runs[1:]:
                if exact:
                    np.testing.assert_array_equal(psi, runs[0][0])
                    assert acc == runs[0][1]
                else:
                    assert rel_to_max(psi, runs[0][0]) <= 1e-14 and abs(acc - runs[0][1]) <= 1e-13 * abs(acc)
        finally:
            be.close()


def test_alternating_tile_direction_is_rounding_only():
    """Option "alternate" (default 1): consecutive sweeps walk every lane's tiles in alternating directions so that a
    sweep starts on the pool data the previous one left in the XCD's L2.  Same direction => same bits; opposite
    direction => summation-order rounding; trades are bit-identical either way; alternate = 0 => every sweep equal."""
    n = 256
    batches = [synth.product_pools(400_000, n, seed=101), synth.geomean_pools(300_000, n, seed=102)]
    v = synth.sweep_prices(n, seed=103)
    be = cr.DeviceBackend(n, batches)
    try:
        runs = []
        for _ in range(4):
            psi, acc = be.find_arb(v)
            runs.append((psi, acc, be.trades()))
        np.testing.assert_array_equal(runs[0][0], runs[2][0])
        np.testing.assert_array_equal(runs[1][0], runs[3][0])
        assert runs[0][1] == runs[2][1] and runs[1][1] == runs[3][1]
        assert rel_to_max(runs[1][0], runs[0][0]) <= 1e-14
        for k in (1, 2, 3):
            np.testing.assert_array_equal(runs[k][2][0], runs[0][2][0])    # Δ
            np.testing.assert_array_equal(runs[k][2][1], runs[0][2][1])    # Λ
        be.ctx.set_option("alternate", 0)
        a, b = be.eval(v), be.eval(v)
        np.testing.assert_array_equal(a[0], b[0])
        assert a[1] == b[1]
    finally:
        be.close()


@pytest.mark.parametrize("n_fees", [1, 2, 256, 257, 5000])
def test_packed_fee_token_records(n_fees):
    """Sweeps read an 8-byte {i1 | i2 << 16, fee-table index} record instead of gamma + Ai when a launch's distinct fees
    fit the 256-entry LDS table (option "pack"); markets with more fee tiers fall back to the plain arrays.  Same bits."""
    n = 300
    rng = np.random.default_rng(n_fees)
    fees = np.sort(rng.uniform(0.9, 1.0, n_fees))
    bp, bg = synth.product_pools(150_000, n, seed=111), synth.geomean_pools(90_000, n, seed=112)
    bp.γ[:] = fees[rng.integers(0, n_fees, len(bp))]
    bg.γ[:] = fees[rng.integers(0, n_fees, len(bg))]
    v = synth.sweep_prices(n, seed=113)
    res = {}
    for pack in (1, 0):
        be = cr.DeviceBackend(n, [bp, bg])
        be.ctx.set_option("pack", pack)
        try:
            psi, acc = be.find_arb(v)
            res[pack] = (psi, acc) + be.trades()
        finally:
            be.close()
    np.testing.assert_array_equal(res[1][0], res[0][0])
    assert res[1][1] == res[0][1]
    np.testing.assert_array_equal(res[1][2], res[0][2])
    np.testing.assert_array_equal(res[1][3], res[0][3])
    Do, Lo, psi_o, acc_o = oracle_sweep([bp, bg], n, v, nthreads=8)
    assert rel_to_max(res[1][0], psi_o) <= 1e-12
    np.testing.assert_array_equal(res[1][2][:len(bp)], Do[:len(bp)])


def test_compact_trade_records_are_lossless():
    """Option "compact_trades" (default 1): a materialising sweep writes ONE 16-byte record per pool ({+Δ₁, Λ₂} or
    {−Δ₂, Λ₁}; both-direction pools -- γ > 1 -- go through overflow rows).  cfmm_get_trades, cfmm_get_trades_range,
    cfmm_trades_dev and update_reserves! must see exactly the trades of the plain 32-byte layout."""
    import ctypes
    n = 48
    bp, bg = synth.product_pools(60_000, n, seed=121), synth.geomean_pools(40_000, n, seed=122)
    bg.γ[::7] = 1.02                      # fees above 1: both directions of such a pool can trade
    bp.γ[::11] = 1.0
    bu = synth.univ3_pools(8_000, n, 6, seed=123)
    v = synth.sweep_prices(n, seed=124, spread=0.02)   # near the no-arbitrage manifold: many idle pools, some γ > 1 pools in both directions
    res = {}
    for compact in (1, 0):
        be = cr.DeviceBackend(n, [bp, bg, bu])
        be.ctx.set_option("compact_trades", compact)
        try:
            psi, acc = be.find_arb(v)
            D, L = be.trades()
            Dw, Lw = be.ctx.trades_range(1, 5, 30_000)
            da, la = be.ctx.trades_dev()
            m = len(D)
            hD, hL = np.empty((m, 2)), np.empty((m, 2))
            import torch
            torch.cuda.synchronize()
            hip = ctypes.CDLL("libamdhip64.so")
            assert hip.hipDeviceSynchronize() == 0
            assert hip.hipMemcpy(hD.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(da), m * 16, 2) == 0
            assert hip.hipMemcpy(hL.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(la), m * 16, 2) == 0
            be.ctx.update_reserves()
            res[compact] = (psi, acc, D, L, Dw, Lw, hD, hL, be.ctx.reserves(0, len(bp)), be.ctx.reserves(1, len(bg)))
        finally:
            be.close()
    both = np.count_nonzero((res[0][2] > 0).all(axis=1) | ((res[0][2][:, 0] > 0) & (res[0][3][:, 0] > 0)))
    assert both > 0, "the market must contain pools that trade in both directions (overflow rows)"
    for a, b in zip(res[1], res[0]):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(res[1][6], res[1][2])      # expanded device arrays == downloaded trades
    np.testing.assert_array_equal(res[1][7], res[1][3])
    np.testing.assert_array_equal(res[1][4], res[1][2][60_005:90_005])
    Do, Lo, psi_o, _ = oracle_sweep([bp, bg, bu], n, v, nthreads=8)
    np.testing.assert_array_equal(res[1][2][:60_000], Do[:60_000])
    np.testing.assert_array_equal(res[1][2][100_000:], Do[100_000:])


@pytest.mark.parametrize("n", [2, 7, 8, 64, 257, 1000])
def test_output_granules_equal_flagged_outputs(n):
    """Option "host_granules" (default 1): a host-pointer sweep returns {Ψ, acc} as self-validating 8-byte granules
    (sequence tag + half of a double, 16 per fold block = two full 64-byte lines per store instruction) that the host
    re-reads until all carry the tag, instead of outputs + drain + ticket + flag word.  Same values, any n_tokens
    (the last fold block pads its lines with zero columns)."""
    m = 40_000 if n > 2 else 300
    batches = [synth.product_pools(m, n, seed=n)] + ([synth.geomean_pools(m // 2, n, seed=n + 1)] if n > 2 else [])
    a, b = cr.DeviceBackend(n, batches), cr.DeviceBackend(n, batches)
    a.ctx.set_option("host_granules", 1)
    b.ctx.set_option("host_granules", 0)
    try:
        for it in range(25):
            v = synth.sweep_prices(n, seed=300 + it)
            for mat in (False, True):
                pa = (a.find_arb if mat else a.eval)(v)
                pb = (b.find_arb if mat else b.eval)(v)
                if n > 600:    # the wavefronts of a block share one LDS bin copy: summation order is not fixed
                    assert rel_to_max(pa[0], pb[0]) <= 1e-14 and abs(pa[1] - pb[1]) <= 1e-12 * abs(pb[1])
                else:
                    np.testing.assert_array_equal(pa[0], pb[0])
                    assert pa[1] == pb[1]
    finally:
        a.close()
        b.close()
