"""CPU-only tests: host mirror logic (constructors, objectives, packing order, the route loop with
a test-injected oracle backend) and the C-ABI library's load/export surface.  No GPU compute."""
import ctypes
import math
import os
import re

import numpy as np
import pytest

import cfmmrouter_amd as cr
from cfmmrouter_amd import synth
from cfmmrouter_amd._lib import KIND_PRODUCT, LIB_PATH
from oracle import cfmm_oracle as orc
from helpers import oracle_sweep
from helpers import OracleBackend, oracle_objective, oracle_poolset, rel_to_max

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "cfmm_amd.h")).read()
    names = set(re.findall(r"\b(cfmm_[a-z0-9_]+)\s*\(", hdr))
    assert len(names) >= 20
    L = ctypes.CDLL(LIB_PATH)
    missing = [n for n in sorted(names) if not hasattr(L, n)]
    assert not missing, missing


def test_shipped_library_carries_no_test_hooks():
    """the test hook "debug_stall_ms" exists only in libcfmm_amd_hooks.so (-DCFMM_TEST_HOOKS), and the
    header documents no debug keys."""
    blob = open(LIB_PATH, "rb").read()
    assert b"debug_stall_ms" not in blob and b"debug_dev_trust" not in blob
    hdr = open(os.path.join(ROOT, "include", "cfmm_amd.h")).read()
    assert not re.findall(r'"debug_[a-z_]+"', hdr.split("EXPERIMENTAL")[0])
    hooks = os.path.join(os.path.dirname(LIB_PATH), "libcfmm_amd_hooks.so")
    if os.path.exists(hooks):
        assert b"debug_stall_ms" in open(hooks, "rb").read()


def test_unresolvable_rccl_is_an_error_code_not_a_crash():
    """ADVICE r5 (abi_rccl.cpp): a host without a loadable RCCL gets CFMM_ERR_UNSUPPORTED and a message -- round 5 built
    the message from TWO dlerror() calls, the second of which returns NULL (std::string + nullptr: SIGSEGV), so the probe
    dist.join_library_rccl makes on every rank would have killed the process instead of falling back.  CFMM_AMD_RCCL_LIB
    names the one image RCCL is resolved from; resolution happens once per process, hence a child process."""
    import subprocess
    import sys
    code = (
        "import ctypes, sys\n"
        f"L = ctypes.CDLL({LIB_PATH!r})\n"
        "L.cfmm_last_error.restype = ctypes.c_char_p\n"
        "L.cfmm_last_error.argtypes = [ctypes.c_void_p]\n"
        "buf = ctypes.create_string_buffer(128)\n"
        "rc = L.cfmm_rccl_unique_id(buf)\n"
        "rc2 = L.cfmm_rccl_unique_id(buf)\n"
        "print(rc, rc2, L.cfmm_last_error(None).decode())\n")
    env = dict(os.environ, CFMM_AMD_RCCL_LIB="/nonexistent/librccl_absent.so.1")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, (out.returncode, out.stderr[-400:])
    rc, rc2, msg = out.stdout.strip().split(" ", 2)
    assert int(rc) == -4 and int(rc2) == -4                      # CFMM_ERR_UNSUPPORTED, again on the second call
    assert "RCCL is not available" in msg and "librccl_absent" in msg, msg


def test_no_cpu_fallback_without_device():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(cr.CFMMDeviceError, match="no CPU fallback"):
        cr.Context(4)
    with pytest.raises(cr.CFMMDeviceError):
        cr.Router(cr.LinearNonnegative(np.ones(2)), [cr.ProductTwoCoin([1, 1], 1, [1, 2])], 2)


def test_package_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "cfmmrouter.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, fn), encoding="utf-8").read()
                assert "oracle" not in src.lower(), f"{fn} mentions the oracle"


class TestConstructors:
    def test_two_coin(self):  # test/cfmms.jl:89-90, src/cfmms.jl:76-90
        p = cr.ProductTwoCoin([1, 1], .9, [1, 2])
        assert len(p) == 2 and p.R.dtype == np.float64 and p.γ == 0.9
        with pytest.raises(cr.ArgumentError):
            cr.ProductTwoCoin([1, 1], .9, [1])
        with pytest.raises(cr.ArgumentError):
            cr.ProductTwoCoin([1, 1, 1], .9, [1, 2])
        g = cr.GeometricMeanTwoCoin([1, 2], [0.3, 0.7], 1, [2, 1])
        assert list(g.Ai) == [2, 1] and g.w[1] == 0.7

    def test_trading_functions_match_oracle(self):  # src/cfmms.jl:113-122, :167-178
        rng = np.random.default_rng(3)
        for _ in range(20):
            R, w1 = rng.random(2) * 100 + 0.1, rng.uniform(0.05, 0.95)
            p = cr.ProductTwoCoin(R, 0.997, [1, 2])
            g = cr.GeometricMeanTwoCoin(R, [w1, 1 - w1], 0.997, [1, 2])
            R2 = rng.random(2) * 50 + 0.1
            out = np.zeros(2)
            assert cr.ϕ(p) == orc.product_phi(R) and cr.ϕ(p, R=R2) == orc.product_phi(R2)
            cr.ϕ_grad_(out, p, R=R2)
            np.testing.assert_array_equal(out, orc.product_grad_phi(R2))
            assert abs(cr.ϕ(g, R=R2) - orc.geomean_phi(R2, g.w)) <= 1e-14 * orc.geomean_phi(R2, g.w)
            cr.ϕ_grad_(out, g, R=R2)
            np.testing.assert_allclose(out, orc.geomean_grad_phi(R2, g.w), rtol=1e-14)
        with pytest.raises(cr.ArgumentError):
            cr.ϕ(cr.UniV3(15.0, [30., 20, 10, 5], [1.0, 2.0, 1.5, 0.0], 1.0, [1, 2]))

    def test_univ3_current_tick(self):  # src/cfmms.jl:235
        for cp, ct in [(15.0, 2), (20.0, 2), (30.0, 1), (31.0, 0), (1.0, 4)]:
            assert cr.UniV3(cp, [30., 20, 10, 5], [1.0, 2.0, 1.5, 0.0], 1.0, [1, 2]).current_tick == ct

    def test_bounded_product_is_two_tick_univ3(self):
        b = cr.BoundedProduct(1.0, 0.5, 2.0, 10.0, 0.997, [1, 2])
        assert list(b.lower_ticks) == [2.0, 0.5] and list(b.liquidity) == [10.0, 0.0] and b.current_tick == 1


class TestObjectives:  # test/objectives.jl:1-46
    def test_linear_nonnegative(self):
        with pytest.raises(cr.ArgumentError):
            cr.LinearNonnegative(-np.ones(2))
        assert cr.LinearNonnegative([1, 1]).c.dtype == np.float64
        obj = cr.LinearNonnegative(np.ones(2))
        assert cr.f(obj, 2 * np.ones(2)) == 0
        assert math.isinf(cr.f(obj, 0.5 * np.ones(2)))
        x = np.ones(2)
        cr.grad_(x, obj, 2 * np.ones(2))
        assert np.all(x == 0)
        cr.grad_(x, obj, 0.5 * np.ones(2))
        assert np.all(np.isinf(x))

    def test_basket_liquidation(self):
        with pytest.raises(cr.ArgumentError):
            cr.BasketLiquidation(0, [0.0, 1.0])
        obj = cr.BasketLiquidation(1, [0, 1])
        assert obj.Δin.dtype == np.float64
        assert cr.f(obj, [2, 3]) == 3
        assert math.isinf(cr.f(obj, 0.5 * np.ones(2)))
        x = np.ones(2)
        cr.grad_(x, obj, 2 * np.ones(2))
        assert np.all(x - [0, 1] == 0)
        cr.grad_(x, obj, 0.5 * np.ones(2))
        assert np.all(np.isinf(x))

    def test_swap(self):
        swap, obj = cr.Swap(1, 2, 5.0, 3), cr.BasketLiquidation(1, [0.0, 5.0, 0.0])
        assert isinstance(swap, cr.BasketLiquidation)
        assert np.all(swap.Δin == obj.Δin) and swap.i == obj.i

    def test_host_objectives_match_oracle(self):
        rng = np.random.default_rng(0)
        c, v = rng.random(9) + 0.1, rng.random(9) + 0.6
        for o in (cr.LinearNonnegative(c), cr.BasketLiquidation(3, rng.random(9) * 100)):
            oo = oracle_objective(o)
            for vv in (v, v + 1.0, np.ones(9)):
                assert cr.f(o, vv) == oo.f(vv)
                g = np.empty(9)
                cr.grad_(g, o, vv)
                np.testing.assert_array_equal(g, oo.grad(vv))
            np.testing.assert_array_equal(cr.lower_limit(o), oo.lower_limit())
            assert np.all(np.isinf(cr.upper_limit(o)))


def check_primal_feasibility(r, arb=True, TOL=1e-3):
    """test/arb.jl:5-23.  The reference uses TOL = 1e-4 on its one Julia-RNG instance; L-BFGS-B
    stops on factr (relative reduction of f) there and here, which leaves constraint-active
    netflows of up to ~1.5e-4 on our own seeded instances, hence 1e-3."""
    all_flows = np.zeros_like(r.v)
    for Δ, Λ, c in zip(r.Δs, r.Λs, r.cfmms):
        assert np.all(Δ >= -TOL) and np.all(Λ >= -TOL)
        Rn = c.R + c.γ * Δ - Λ
        assert Rn[0] * Rn[1] >= c.R[0] * c.R[1] - math.sqrt(np.finfo(float).eps)
        all_flows[c.Ai - 1] += Λ - Δ
    assert np.all(all_flows == cr.netflows(r, exact=True))     # test/arb.jl:16: the reference's loop over the rows, bit for bit
    assert np.max(np.abs(all_flows - cr.netflows(r))) <= 1e-12 * max(1.0, np.max(np.abs(all_flows)))   # the backend's own reduction
    if arb:
        assert np.all(all_flows >= -TOL)
    else:
        assert np.sum(all_flows >= -TOL) == 1


def check_dual_feasibility(r, TOL=1e-4):
    """test/arb.jl:25-28"""
    assert np.all(r.v >= cr.lower_limit(r.objective) - TOL)
    assert np.all(r.v <= cr.upper_limit(r.objective) + TOL)


def oracle_router(objective, pools, n):
    probe = cr.Router.__new__(cr.Router)  # pack first to learn the batches the backend needs
    from cfmmrouter_amd.router import _segments_of
    batches, _, _ = _segments_of(pools if isinstance(pools, cr.PoolBatch) else list(pools))
    return cr.Router(objective, pools, n, _backend=OracleBackend(n, batches))


class TestRouteLoopOnOracleBackend:
    """The reference's integration tests (test/arb.jl, test/swap.jl) against the host route loop,
    with the device swapped for the oracle -- checks the loop, the cache rule and the packing."""

    def test_arb_simple(self):  # test/arb.jl:42-58
        pools = [cr.ProductTwoCoin([100, 100], 1, [1, 2]), cr.ProductTwoCoin([1, 2], 1, [1, 2])]
        r = oracle_router(cr.LinearNonnegative(np.ones(2)), pools, 2)
        cr.route_(r)
        check_primal_feasibility(r)
        check_dual_feasibility(r)

    def test_arb_random(self):  # test/arb.jl:60-85
        n = 10
        b = synth.product_pools(100, n, seed=1234)
        b.γ[:] = 1.0
        r = oracle_router(cr.LinearNonnegative(synth.linear_prices(n)), [b[i] for i in range(100)], n)
        cr.route_(r)
        check_primal_feasibility(r)
        check_dual_feasibility(r)

    def test_swap_simple_and_random(self):  # test/swap.jl:1-46
        pools = [cr.ProductTwoCoin([100, 100], 1, [1, 2]), cr.ProductTwoCoin([1, 2], 1, [1, 2])]
        r = oracle_router(cr.BasketLiquidation(1, [5.0, 0.0]), pools, 2)
        cr.route_(r)
        check_primal_feasibility(r)
        check_dual_feasibility(r)
        n = 10
        b = synth.product_pools(100, n, seed=1234)
        b.γ[:] = 1.0
        r = oracle_router(cr.BasketLiquidation(1, synth.basket(n)), [b[i] for i in range(100)], n)
        cr.route_(r)
        check_primal_feasibility(r, arb=False)
        check_dual_feasibility(r)

    def test_readme_example_value(self):  # README.md:27-38; analytic check from SURVEY §8c
        pools = [cr.ProductTwoCoin([1e6, 1e6], 1, [1, 2]), cr.ProductTwoCoin([1e3, 2e3], 1, [1, 2])]
        r = oracle_router(cr.LinearNonnegative(np.ones(2)), pools, 2)
        cr.route_(r)
        Ψ = cr.netflows(r)
        p = r.v[0] / r.v[1]
        assert abs(Ψ[1] - ((1e6 - 1e6 * math.sqrt(p)) + (2e3 - math.sqrt(2e6 * p)))) < 1e-6
        assert abs(Ψ[1] - 171.4) < 0.1 and abs(Ψ[0]) < 1e-3

    def test_host_loop_equals_oracle_route(self):
        n, m = 32, 3000
        b = synth.product_pools(m, n, seed=3)
        obj = cr.LinearNonnegative(synth.linear_prices(n, seed=3))
        r = oracle_router(obj, b, n)
        cr.route_(r, v=np.ones(n))
        ref = orc.route_oracle(oracle_objective(obj), oracle_poolset([b], n), v0=np.ones(n))
        np.testing.assert_array_equal(r.v, ref["v"])          # same callbacks -> same iterates
        np.testing.assert_array_equal(cr.netflows(r), ref["psi"])
        assert r.n_sweeps == ref["n_sweeps"]

    def test_native_solver_on_injected_backend(self):
        n, m = 16, 1500
        b = synth.product_pools(m, n, seed=8)
        obj = cr.BasketLiquidation(1, synth.basket(n, seed=8))
        r1, r2 = oracle_router(obj, b, n), oracle_router(obj, b, n)
        cr.route_(r1)
        cr.route_(r2, solver="native")
        assert rel_to_max(cr.netflows(r2), cr.netflows(r1)) <= 1e-6
        np.testing.assert_allclose(r2.v, r1.v, rtol=1e-6)
        check_primal_feasibility(r2, arb=False)

    def test_interleaved_families_keep_router_order(self):
        n = 6
        bp, bg = synth.product_pools(5, n, 1), synth.geomean_pools(4, n, 2)
        pools = [bp[0], bg[0], bp[1], bg[1], bp[2], bg[2], bp[3], bg[3], bp[4]]
        r = oracle_router(cr.LinearNonnegative(np.ones(n)), pools, n)
        v = synth.sweep_prices(n, 5)
        cr.find_arb_(r, v)
        for i, c in enumerate(pools):
            if c.kind == KIND_PRODUCT:
                D, L = orc.product_find_arb(c.R, c.γ, v[c.Ai - 1])
            else:
                D, L = orc.geomean_find_arb(c.R, c.w, c.γ, v[c.Ai - 1])
            np.testing.assert_array_equal(r.Δs[i], D)
            np.testing.assert_array_equal(r.Λs[i], L)

    def test_update_reserves_and_no_fee_optimality(self):
        """The reference's disabled check_opt_conditions_no_fee! (test/arb.jl:30-39): after route!
        and update_reserves!, every pool's marginal price ∇φ(R) is parallel to v[Ai] (γ = 1)."""
        n = 10
        b = synth.product_pools(100, n, seed=1234)
        b.γ[:] = 1.0
        pools = [b[i] for i in range(100)]
        r = oracle_router(cr.LinearNonnegative(synth.linear_prices(n)), pools, n)
        cr.route_(r)
        k0 = np.array([c.R[0] * c.R[1] for c in pools])
        cr.update_reserves_(r)
        for c, k in zip(r.cfmms, k0):
            p = np.array([c.R[1], c.R[0]])                       # ∇φ for ProductTwoCoin, src/cfmms.jl:117-122
            vv = r.v[c.Ai - 1]
            assert abs(p @ vv / (np.linalg.norm(p) * np.linalg.norm(vv)) - 1.0) < 1e-9   # cosangle ≈ 1
            assert abs(c.R[0] * c.R[1] - k) <= 1e-9 * k          # zero fee: the invariant is preserved
        cr.route_(r)                                             # nothing is left to arbitrage
        assert np.max(np.abs(cr.netflows(r))) < 1e-3

    def test_update_reserves_unsupported_for_univ3(self):
        r = oracle_router(cr.LinearNonnegative(np.ones(2)), [cr.UniV3(15.0, [30., 20, 10, 5], [1., 2, 1.5, 0], 1.0, [1, 2])], 2)
        cr.find_arb_(r, np.array([16.0, 1.0]))
        with pytest.raises(NotImplementedError):
            cr.update_reserves_(r)


class TestSynth:
    def test_counter_mode_is_order_independent(self):
        a = synth.uniform(7, 3, 1000)
        b = np.concatenate([synth.uniform(7, 3, 400), synth.uniform(7, 3, 600, first=400)])
        np.testing.assert_array_equal(a, b)
        assert 0.0 <= a.min() and a.max() < 1.0 and abs(a.mean() - 0.5) < 0.05

    def test_pairs_distinct_in_range(self):
        p = synth.token_pairs(1, 2, 100_000, 7)
        assert p.min() == 1 and p.max() == 7 and np.all(p[:, 0] != p[:, 1])

    def test_shard_slices_reassemble(self):
        b = synth.univ3_pools(50, 9, 3, seed=2)
        parts = [b.slice(0, 20), b.slice(20, 50)]
        assert np.array_equal(np.concatenate([p.lower_ticks for p in parts]), b.lower_ticks)
        assert parts[1].tick_off[0] == 0 and len(parts[1]) == 30


def test_market_file_round_trip(tmp_path):
    from cfmmrouter_amd import poolfile
    n = 12
    batches = [synth.product_pools(30, n, 1), synth.geomean_pools(20, n, 2), synth.univ3_pools(10, n, 4, 3)]
    for obj, v0 in ((cr.LinearNonnegative(synth.linear_prices(n, 4)), np.ones(n)),
                    (cr.BasketLiquidation(2, synth.basket(n, 5)), None)):
        p = str(tmp_path / "m.bin")
        poolfile.save_market(p, batches, n, obj, v0)
        b2, n2, obj2, v02 = poolfile.load_market(p)
        assert n2 == n and type(obj2) is type(obj) and (v0 is None) == (v02 is None)
        for a, b in zip(batches, b2):
            assert a.kind == b.kind
            np.testing.assert_array_equal(a.γ, b.γ)
            np.testing.assert_array_equal(a.Ai, b.Ai)
            if a.kind != 2:
                np.testing.assert_array_equal(a.R, b.R)
            else:
                np.testing.assert_array_equal(a.lower_ticks, b.lower_ticks)
                np.testing.assert_array_equal(a.tick_off, b.tick_off)


def test_c_abi_error_codes_without_device():
    """Argument validation that does not need a GPU, straight through ctypes."""
    import ctypes as C
    L = cr.lib()
    h = C.c_void_p()
    assert L.cfmm_ctx_create(0, 0, C.byref(h)) == -1 and b"n_tokens" in L.cfmm_last_error(None)
    assert L.cfmm_ctx_create(0, 4, None) == -1
    assert L.cfmm_ctx_create(0, (1 << 26) + 1, C.byref(h)) == -4
    assert L.cfmm_pools_count(None) == 0 and L.cfmm_n_tokens(None) == 0 and L.cfmm_segment_count(None) == 0
    assert L.cfmm_set_stream(None, None) == -1 and L.cfmm_find_arb(None, None) == -1
    assert L.cfmm_peer_buffer_bytes(512) == 4 * 513 * 8 and L.cfmm_set_peers(None, None, 0, 0, 0) == -1
    assert b"gfx950" in L.cfmm_version()


def test_bench_self_spawns_one_rank_per_gpu(tmp_path):
    """`python bench.py --gpus 2` with no torchrun environment re-executes itself under
    torch.distributed.run (127.0.0.1 rendezvous): both ranks start; with no GPU each fails loudly."""
    import subprocess
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    out = r.stdout + r.stderr
    assert "[bench] spawning 2 ranks" in out and "--nproc-per-node=2" in out and "127.0.0.1" in out
    import torch
    if not torch.cuda.is_available():
        assert r.returncode != 0
        # (torchrun tears the other rank down as soon as one has failed: at least one of them got to say why)
        assert "[rank 0/2] bench.py needs an MI355X" in out or "[rank 1/2] bench.py needs an MI355X" in out


def test_univ3_ticks_visited_bookkeeping_matches_the_oracle_walk():
    """bench.py prices a multi-tick UniV3 sweep by the ticks the walk VISITS (32 B + 16 B per tick): the
    bookkeeping function must agree with what find_arb! (src/cfmms.jl:339-395, CPU restatement) does -- a pool counted
    as idle trades nothing, and a pool that trades through k ticks has at least k - 1 of them drained to their bound."""
    n = 24
    b = synth.univ3_ragged_pools(4000, n, seed=5)
    v = synth.sweep_prices(n, seed=6) * synth.token_price_vector(n, seed=5)
    vis = synth.univ3_ticks_visited(b, v)
    D, L, _, _ = oracle_sweep([b], n, v)
    assert np.all(D[vis == 0] == 0) and np.all(L[vis == 0] == 0)
    assert np.mean(vis > 1) > 0.3 and vis.max() <= np.diff(b.tick_off).max()
    shard = synth.univ3_ragged_pools(1000, n, seed=5, first=3000)       # shards regenerate the same pools
    np.testing.assert_array_equal(shard.lower_ticks, b.slice(3000, 4000).lower_ticks)
    np.testing.assert_array_equal(shard.liquidity, b.slice(3000, 4000).liquidity)
