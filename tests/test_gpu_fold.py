"""GPU tests of the launch forms of one evaluation: the row fold and its host-visible outputs (self-validating
granules, option "host_flag"), fused multi-family launches under their block -> segment maps, empty segments,
the determinism claim, alternating tile directions, packed fee + token records, compact trade records, and the
fast arithmetic (option "fast_math": division / square root without range scaffolding) -- every variant must
produce the bits of the plain form (src/router.jl:81-83, :98-100 summed over blocks; src/cfmms.jl:125-140).
"""
import numpy as np
import pytest

import cfmmrouter_amd as cr
from cfmmrouter_amd import synth
from helpers import oracle_sweep, rel_to_max

pytestmark = pytest.mark.gpu


def test_host_flag_and_stream_wait_agree():
    """Zero-copy host-pointer sweeps end when {Ψ, acc} have arrived in mapped host memory as self-validating
    granules (option "host_flag", default on); waiting for the stream + a D2H copy instead must give the same bits."""
    n = 256
    batches = [synth.product_pools(250_000, n, seed=41), synth.geomean_pools(250_000, n, seed=42)]
    a = cr.DeviceBackend(n, batches)
    b = cr.DeviceBackend(n, batches)
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
    for fuse in (0, 1):
        be = cr.DeviceBackend(n, [])
        be.ctx.set_option("fuse_segments", fuse)
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


@pytest.mark.parametrize("opts", [dict(), dict(max_grid=200), dict(fast_math=0), dict(block=1024)])
@pytest.mark.parametrize("families", ["pg", "pgu", "pgup"])
def test_fused_launch_block_maps_agree_with_oracle(opts, families):
    """The fused multi-family launch under its block -> segment maps (XCD-aware when the grid is a multiple of 256
    blocks, plain b % nseg otherwise -- max_grid=200): same trades bit for bit, same Ψ to summation-order rounding."""
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
    """Option "alternate" (default 1): consecutive evaluations walk every lane's tiles in alternating directions so that a
    sweep starts on the pool data the previous one left in the XCD's L2.  find_arb!(r, v) itself always walks forwards
    (a function of v alone, test/arb.jl:16); fused evaluations: same direction => same bits, opposite direction =>
    summation-order rounding; alternate = 0 => every sweep equal."""
    n = 256
    batches = [synth.product_pools(400_000, n, seed=101), synth.geomean_pools(300_000, n, seed=102)]
    v = synth.sweep_prices(n, seed=103)
    be = cr.DeviceBackend(n, batches)
    try:
        runs = []
        for k in range(4):
            for _ in range(k):
                be.eval(v * (1.0 + 0.01 * k))              # whatever ran before
            psi, acc = be.find_arb(v)
            runs.append((psi, acc, be.trades()))
        for k in (1, 2, 3):
            np.testing.assert_array_equal(runs[k][0], runs[0][0])          # Ψ
            assert runs[k][1] == runs[0][1]
            np.testing.assert_array_equal(runs[k][2][0], runs[0][2][0])    # Δ
            np.testing.assert_array_equal(runs[k][2][1], runs[0][2][1])    # Λ
        back, fwd = be.eval(v), be.eval(v)                 # behind a find_arb!: backwards, then forwards again
        np.testing.assert_array_equal(fwd[0], runs[0][0])
        assert fwd[1] == runs[0][1]
        assert rel_to_max(back[0], runs[0][0]) <= 1e-14
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
def test_output_granules_equal_copied_outputs(n):
    """A host-pointer sweep returns {Ψ, acc} as self-validating 8-byte granules (sequence tag + half of a double, 16 per
    fold block = two full 64-byte lines per store instruction) that the host re-reads until all carry the tag.  Same
    values as d_out + a D2H copy (zero_copy = 0), any n_tokens (the last fold block pads its lines with zero columns)."""
    m = 40_000 if n > 2 else 300
    batches = [synth.product_pools(m, n, seed=n)] + ([synth.geomean_pools(m // 2, n, seed=n + 1)] if n > 2 else [])
    a, b = cr.DeviceBackend(n, batches), cr.DeviceBackend(n, batches)
    b.ctx.set_option("zero_copy", 0)
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


def _wide_market(n, seed, scale_exp):
    """Product / GeoMean / UniV3 pools whose reserves, liquidity and prices are scaled by 2^scale_exp."""
    bp, bg = synth.product_pools(60_000, n, seed=seed), synth.geomean_pools(30_000, n, seed=seed + 1)
    bu = synth.univ3_pools(10_000, n, 6, seed=seed + 2)
    f = 2.0 ** scale_exp
    bp.R *= f
    bg.R *= f
    bu.liquidity *= f
    return [bp, bg, bu]


@pytest.mark.parametrize("scale_exp", [0, 100, -100, 140, 200, -200])
def test_fast_math_is_bit_identical_and_falls_back_outside_its_window(scale_exp):
    """Option "fast_math" (default 1): inside the operand window [2^-150, 2^150] divisions and square roots run the
    compiler's own sequences without range scaffolding, with reciprocals refined once per token / fee tier; the trades
    must be THE SAME BITS as with fast_math = 0 and as the CPU restatement's (IEEE / and sqrt).  Pools scaled beyond
    the window (2^±200) make the upload clear the segment's flag: still the same bits, through the compiler's forms."""
    n = 96
    batches = _wide_market(n, 501, scale_exp)
    v = synth.sweep_prices(n, seed=77)
    res = {}
    for fast in (1, 0):
        be = cr.DeviceBackend(n, batches)
        be.ctx.set_option("fast_math", fast)
        try:
            psi, acc = be.find_arb(v)
            res[fast] = (psi, acc) + be.trades()
        finally:
            be.close()
    g0, g1 = 60_000, 90_000     # rows [g0, g1) are GeometricMeanTwoCoin pools
    for k in (2, 3):            # ProductTwoCoin and UniV3 rows: the same bits
        np.testing.assert_array_equal(res[1][k][:g0], res[0][k][:g0])
        np.testing.assert_array_equal(res[1][k][g1:], res[0][k][g1:])
    # GeometricMean (log-space form, within 1e-12 of the reference either way): the fast arithmetic evaluates the one
    # exponential with its own < 1 ulp polynomial instead of the device library's, so trades agree to a few ulp
    scale = np.maximum(batches[1].R.max(axis=1), 1.0)[:, None]
    for k in (2, 3):
        assert np.max(np.abs(res[1][k][g0:g1] - res[0][k][g0:g1]) / scale) <= 1e-14
    assert rel_to_max(res[1][0], res[0][0]) <= 1e-13 and abs(res[1][1] - res[0][1]) <= 1e-13 * abs(res[0][1])
    if abs(scale_exp) > 150:    # outside the window both runs take the compiler's sequences and the library's exp
        for a, b in zip(res[1], res[0]):
            np.testing.assert_array_equal(a, b)
    Do, Lo, psi_o, _ = oracle_sweep(batches, n, v, nthreads=8)
    np.testing.assert_array_equal(res[1][2][:g0], Do[:g0])
    np.testing.assert_array_equal(res[1][3][:g0], Lo[:g0])
    np.testing.assert_array_equal(res[1][2][g1:], Do[g1:])
    np.testing.assert_array_equal(res[1][3][g1:], Lo[g1:])


@pytest.mark.parametrize("vscale", [2.0 ** 160, 2.0 ** -160, 1.0])
def test_fast_math_price_window_is_checked_by_host_and_by_every_block(vscale):
    """Prices outside the window of the fast arithmetic (the arithmetic is chosen per KERNEL).
    Host-pointer calls see the prices and launch the full-range kernels: same bits as fast_math = 0 and as the CPU
    restatement.  Device-pointer sweeps (cfmm_sweep_dev: prices the library cannot see) launch the kernels that carry BOTH
    arithmetics and every block picks from the prices it stages (round 5): the SAME exact result, CFMM_OK, no state change --
    round 4 launched the fast kernels on trust, delivered NaN and reported it on a later call.  The
    blocks' own check of a FAST kernel is still there (dev_prices_in_window = 1, the caller's promise that its device prices
    are inside the window, broken here: the fast kernel refuses such prices -- all NaN, never a wrong number)."""
    import torch
    n = 64
    batches = [synth.product_pools(80_000, n, seed=611), synth.univ3_pools(9_000, n, 4, seed=612)]
    v = synth.sweep_prices(n, seed=613)
    v[::3] *= vscale                      # a third of the tokens far outside (price ratios up to 2^±160)
    Do, Lo, psi_o, _ = oracle_sweep(batches, n, v, nthreads=8)
    res = {}
    for fast in (1, 0):
        be = cr.DeviceBackend(n, batches)
        be.ctx.set_option("fast_math", fast)
        try:
            psi_h, _ = be.find_arb(v)                                  # host pointer: the library picks the kernel
            Dh, Lh = be.trades()
            np.testing.assert_array_equal(Dh, Do)
            np.testing.assert_array_equal(Lh, Lo)
            assert rel_to_max(psi_h, psi_o) <= 1e-12
            vt = torch.from_numpy(v).cuda()
            ot = torch.zeros(n + 1, dtype=torch.float64, device="cuda")
            for _ in range(3):                                         # back-to-back asynchronous sweeps: every one is exact
                be.ctx.sweep_dev(vt.data_ptr(), ot.data_ptr(), True)
            torch.cuda.synchronize()
            assert np.all(np.isfinite(ot.cpu().numpy()))
            res[fast] = (ot.cpu().numpy(),) + be.trades()
            if fast and vscale != 1.0:
                be.ctx.set_option("dev_prices_in_window", 1)          # a broken promise: the fast kernel alone, its blocks refuse these prices
                o2 = torch.zeros(n + 1, dtype=torch.float64, device="cuda")
                be.ctx.sweep_dev(vt.data_ptr(), o2.data_ptr(), False)
                torch.cuda.synchronize()
                assert np.all(np.isnan(o2.cpu().numpy()))
                be.ctx.set_option("dev_prices_in_window", 0)
                psi_h2, _ = be.eval(v)                                 # ... and the stale report does not taint a host-pointer call
                assert rel_to_max(psi_h2, psi_o) <= 1e-12
            v_in = synth.sweep_prices(n, seed=614)                     # and prices inside the window keep working
            vt2 = torch.from_numpy(v_in).cuda()
            be.ctx.sweep_dev(vt2.data_ptr(), ot.data_ptr(), False)
            torch.cuda.synchronize()
            assert rel_to_max(ot.cpu().numpy()[:n], oracle_sweep(batches, n, v_in, nthreads=8)[2]) <= 1e-12
        finally:
            be.close()
    for a, b in zip(res[1], res[0]):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(res[1][1], Do)
    np.testing.assert_array_equal(res[1][2], Lo)
    assert rel_to_max(res[1][0][:n], psi_o) <= 1e-12


@pytest.mark.parametrize("bad", [-1.0, 0.0, float("nan"), float("inf")])
def test_device_pointer_sweep_with_invalid_prices_stays_in_bounds(bad):
    """cfmm_sweep_dev never sees its price vector on the host, so nothing validates it (the host-pointer calls do,
    src/cfmms.jl:129).  Garbage in, garbage out -- but every walk stays inside its list (the threshold scan of a UniV3
    pool is bounded by the list's length, not only by its closing record), nothing faults, and the next sweep at valid
    prices is exact again."""
    import torch
    n = 32
    batches = [synth.product_pools(30_000, n, seed=701), synth.geomean_pools(20_000, n, seed=702),
               synth.univ3_ragged_pools(40_000, n, seed=703)]
    v = synth.sweep_prices(n, seed=704)
    vb = v.copy()
    vb[::5] = bad
    be = cr.DeviceBackend(n, batches)
    try:
        ot = torch.zeros(n + 1, dtype=torch.float64, device="cuda")
        vbt = torch.from_numpy(vb).cuda()
        vt = torch.from_numpy(v).cuda()
        be.ctx.sweep_dev(vbt.data_ptr(), ot.data_ptr(), True)
        torch.cuda.synchronize()
        if bad != bad:       # a NaN price propagates into the pools that touch the token, like the reference's arithmetic
            Db, Lb = be.trades()                           # (round 4 refused the whole sweep); every other pool is exact
            Dq, Lq, _, _ = oracle_sweep(batches, n, vb, nthreads=8)
            Ai = np.concatenate([b_.Ai for b_ in batches]) - 1
            hit = np.isnan(vb)[Ai].any(axis=1)             # pools with a NaN-priced token
            g0q, g1q = 30_000, 50_000
            clean = ~hit
            clean[g0q:g1q] = False                         # (GeometricMean: 1e-12, not bit-exact)
            assert clean.sum() > 10_000 and hit.sum() > 1_000
            np.testing.assert_array_equal(Db[clean], Dq[clean])
            np.testing.assert_array_equal(Lb[clean], Lq[clean])
            assert np.all(np.isnan(Db[hit]).any(axis=1) | np.isnan(Lb[hit]).any(axis=1))
            assert np.all(np.isnan(Dq[:g0q][hit[:g0q]]).any(axis=1) | np.isnan(Lq[:g0q][hit[:g0q]]).any(axis=1))   # ... as on the CPU
        for mat in (True, False):                          # (garbage in, garbage out: nothing faults)
            be.ctx.sweep_dev(vbt.data_ptr(), ot.data_ptr(), mat)
            torch.cuda.synchronize()
        be.ctx.sweep_dev(vt.data_ptr(), ot.data_ptr(), True)
        torch.cuda.synchronize()
        D, L = be.trades()
        Do, Lo, psi_o, acc_o = oracle_sweep(batches, n, v, nthreads=8)
        g0, g1 = 30_000, 50_000
        np.testing.assert_array_equal(D[:g0], Do[:g0])
        np.testing.assert_array_equal(D[g1:], Do[g1:])
        np.testing.assert_array_equal(L[g1:], Lo[g1:])
        out = ot.cpu().numpy()
        assert rel_to_max(out[:n], psi_o) <= 1e-12
    finally:
        be.close()



# ---- single-block launches: the block publishes {Ψ, acc} itself, no fold launch (SweepArgs::direct, round 6) -------------------

def _small_market(kind, m, n, seed):
    if kind == "product":
        return synth.product_pools(m, n, seed=seed)
    if kind == "geomean":
        return synth.geomean_pools(m, n, seed=seed)
    if kind == "bounded":
        return synth.bounded_product_pools(m, n, seed=seed)
    return synth.univ3_pools(m, n, 6, seed=seed)


@pytest.mark.parametrize("kind", ["product", "geomean", "bounded", "univ3"])
@pytest.mark.parametrize("m", [1, 100, 1500, 2048, 2049])
def test_single_block_markets_publish_their_result_without_a_fold(kind, m):
    """Markets of up to 2048 pools of one family (six of the ten sizes of the reference's own benchmark grid, benchmark/scaling.jl:8-38) are swept by ONE 1024-thread block whose row IS the result: it goes straight to the host granules / the device
    output, and no fold launch follows (option direct_small, default on; 2049 pools: the general two-launch geometry).
    Same trades bit for bit as the two-launch form and as the CPU restatement; Ψ to summation-order rounding; host-pointer
    and device-pointer sweeps, fused and materialising, 50 price vectors back to back."""
    import torch
    n = 24
    b = _small_market(kind, m, n, 900 + m)
    one, two = cr.DeviceBackend(n, [b]), cr.DeviceBackend(n, [b])
    two.ctx.set_option("direct_small", 0)
    two.ctx.set_option("alternate", 0)          # (a single-block launch never alternates: its fused and materialising sweeps
                                                #  agree bit for bit whatever ran before; the two-launch form is pinned for the comparison)
    try:
        seg1, seg2 = one.ctx.segments()[0], two.ctx.segments()[0]
        assert (seg1["grid"] == 1 and seg1["block"] == 1024) == (m <= 2048)
        assert seg2["block"] == 512 and seg2["grid"] == (m + 511) // 512
        vt = torch.zeros(n, dtype=torch.float64, device="cuda")
        ot = torch.zeros(n + 1, dtype=torch.float64, device="cuda")
        for it in range(50):
            v = synth.sweep_prices(n, seed=70 + it, spread=0.4)
            Do, Lo, psi_o, acc_o = oracle_sweep([b], n, v)
            scale = max(np.max(np.abs(psi_o)), 1e-300)
            p1, a1 = one.find_arb(v)                                    # host pointer, materialising: granules from the sweep block
            p2, a2 = two.find_arb(v)
            D1, L1 = one.trades()
            D2, L2 = two.trades()
            np.testing.assert_array_equal(D1, D2)
            np.testing.assert_array_equal(L1, L2)
            if kind != "geomean":
                np.testing.assert_array_equal(D1, Do)
                np.testing.assert_array_equal(L1, Lo)
            assert np.max(np.abs(p1 - p2)) <= 1e-13 * scale and np.max(np.abs(p1 - psi_o)) <= 1e-12 * scale
            assert abs(a1 - a2) <= 1e-13 * max(abs(acc_o), 1.0)
            pe, ae = one.eval(v)                                        # fused evaluation: same bits as the materialising sweep
            np.testing.assert_array_equal(pe, p1)
            assert ae == a1
            vt.copy_(torch.from_numpy(v))
            one.ctx.sweep_dev(vt.data_ptr(), ot.data_ptr(), it % 2 == 0)     # device pointer: plain stores to d_out
            torch.cuda.synchronize()
            od = ot.cpu().numpy()
            assert np.max(np.abs(od[:n] - psi_o)) <= 1e-12 * scale
        if m <= 2048:
            one.ctx.set_option("time_kernels", 1)
            one.ctx.kernel_times()
            one.eval(v)
            kt = one.ctx.kernel_times()
            assert kt["sweep_launches"] == 1 and kt["reduce_launches"] == 0      # ONE kernel per evaluation
    finally:
        one.close()
        two.close()


def test_route_on_single_block_markets_armed_and_unarmed():
    """cfmm_route on a 1 000-pool market (one launch per evaluation): pre-armed and launch-when-ready agree bit for bit, a
    cancelled pre-armed launch (prices outside the fast window) publishes nothing and costs nothing, and the result is the CPU
    restatement's within north_star's 1e-6."""
    from cfmmrouter_amd._lib import OBJ_LINEAR_NONNEGATIVE
    from helpers import oracle_objective, oracle_poolset
    from oracle import cfmm_oracle as orc
    n, m = 32, 1000
    b = synth.product_pools(m, n, seed=1234)
    c = synth.linear_prices(n, seed=1234)
    res = []
    for armed in (1, 0):
        be = cr.DeviceBackend(n, [b])
        be.ctx.set_option("armed", armed)
        try:
            assert be.ctx.segments()[0]["grid"] == 1
            for _ in range(3):
                v, psi, info = be.ctx.route(OBJ_LINEAR_NONNEGATIVE, c, 0, v0=np.ones(n))
            res.append((v, psi, info["evaluations"]) + be.trades())
            # prices far outside [2^-150, 2^150]: the waiting fast-kernel launch is cancelled, the evaluation repeats full-range
            c_far = c * 2.0 ** 170
            v2, psi2, info2 = be.ctx.route(OBJ_LINEAR_NONNEGATIVE, c_far, 0, v0=c_far * 1.5)
            assert np.all(np.isfinite(psi2)) and info2["evaluations"] >= 1
            v3, psi3, _ = be.ctx.route(OBJ_LINEAR_NONNEGATIVE, c, 0, v0=np.ones(n))      # clean state afterwards
            np.testing.assert_array_equal(v3, v)
            np.testing.assert_array_equal(psi3, psi)
        finally:
            be.close()
    for x, y in zip(res[0], res[1]):
        np.testing.assert_array_equal(x, y)
    ref = orc.route_oracle(oracle_objective(cr.LinearNonnegative(c)), oracle_poolset([b], n), v0=np.ones(n))
    assert rel_to_max(res[0][1], ref["psi"]) <= 1e-6


def test_single_block_markets_under_the_in_library_exchanges():
    """A market small enough for the single-block geometry on a SHARDED context: the block's row is then only this rank's
    share, so it goes through the fold (cfmm_set_peers: fold + gather, two ranks in one process here) or is written to d_out
    and all-reduced behind the launch (RCCL, world 1) -- never published directly as if it were the market's."""
    import torch
    n = 16
    market = [synth.product_pools(1800, n, seed=321)]
    v = synth.sweep_prices(n, seed=322)
    _, _, psi_o, acc_o = oracle_sweep(market, n, v)
    from cfmmrouter_amd.dist import shard_batches
    ranks = [cr.DeviceBackend(n, shard_batches(market, r, 2)) for r in range(2)]
    try:
        assert all(be.ctx.segments()[0]["grid"] == 1 for be in ranks)
        bufs = [torch.zeros(4 * (n + 1), dtype=torch.float64, device="cuda") for _ in range(2)]
        outs = [torch.zeros(n + 1, dtype=torch.float64, device="cuda") for _ in range(2)]
        vt = torch.from_numpy(v).cuda()
        torch.cuda.synchronize()
        for seq in range(4):
            for r, be in enumerate(ranks):
                be.ctx.set_peers([b.data_ptr() for b in bufs], 2, r, seq)
            for r, be in enumerate(ranks):
                be.ctx.sweep_dev(vt.data_ptr(), outs[r].data_ptr(), seq % 2 == 0)
            torch.cuda.synchronize()
            assert torch.equal(outs[0], outs[1])
            got = outs[0].cpu().numpy()
            assert rel_to_max(got[:n], psi_o) <= 1e-12 and abs(got[n] - acc_o) <= 1e-12 * abs(acc_o)
    finally:
        for be in ranks:
            be.close()
    be = cr.DeviceBackend(n, market)
    try:
        assert be.ctx.segments()[0]["grid"] == 1
        psi0, acc0 = be.eval(v)
        be.ctx.rccl_init_rank(be.ctx.rccl_unique_id(), 1, 0)
        psi1, acc1 = be.eval(v)                                  # direct store to d_out, ncclAllReduce behind it (one rank: identity)
        np.testing.assert_array_equal(psi1, psi0)
        assert acc1 == acc0 and rel_to_max(psi1, psi_o) <= 1e-12
        be.ctx.set_rccl_comm(None)
    finally:
        be.close()


def test_trade_store_policy_changes_no_bit():
    """Option stream_stores (write-through / non-temporal trade-record stores; auto by market size): a cache policy, nothing
    else -- trades (compact records, overflow rows, plain rows), psi and the dual value are identical bit for bit."""
    n = 40
    batches = [synth.product_pools(70_000, n, seed=801), synth.geomean_pools(30_000, n, seed=802),
               synth.univ3_pools(8_000, n, 5, seed=803)]
    batches[1].γ[::7] = 1.01                                   # fee > 1: both directions trade -> overflow rows
    v = synth.sweep_prices(n, seed=804, spread=0.4)
    ref = None
    for compact in (1, 0):
        for policy in (1, 2, 0):
            be = cr.DeviceBackend(n, batches)
            be.ctx.set_option("compact_trades", compact)
            be.ctx.set_option("stream_stores", policy)
            be.ctx.set_option("alternate", 0)
            try:
                psi, acc = be.find_arb(v)
                D, L = be.trades()
            finally:
                be.close()
            if ref is None:
                ref = (psi, acc, D, L)
            np.testing.assert_array_equal(psi, ref[0])
            assert acc == ref[1]
            np.testing.assert_array_equal(D, ref[2])
            np.testing.assert_array_equal(L, ref[3])
    Do, Lo, psi_o, _ = oracle_sweep(batches, n, v)
    np.testing.assert_array_equal(ref[2][:70_000], Do[:70_000])
    np.testing.assert_array_equal(ref[2][100_000:], Do[100_000:])
    assert rel_to_max(ref[0], psi_o) <= 1e-12
