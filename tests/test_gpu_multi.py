"""Single-process multi-device context (cfmm_ctx_create_multi; SURVEY 8b/8e, the sharded axis of
src/router.jl:39) on the 1-GPU box: the device list repeats ordinal 0, so N shards with N pool
stores, N streams and N host worker threads run on one MI355X -- everything but the physical
placement is what an 8-GPU node executes."""
import os
import subprocess
import sys

import numpy as np
import pytest

import cfmmrouter_amd as cr
from cfmmrouter_amd import synth
from helpers import oracle_objective, oracle_poolset, oracle_sweep, rel_to_max
from oracle import cfmm_oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("nd", [1, 2, 8])
def test_plain_c_multi_client(tmp_path, nd):
    exe = str(tmp_path / "abi_multi")
    libdir = os.path.join(ROOT, "cfmmrouter.jl_amd")
    subprocess.run(["gcc", "-O1", "-std=c11", "-Wall", "-I", os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "c", "abi_multi.c"), "-o", exe, "-L", libdir, "-lcfmm_amd",
                    "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-lm"], check=True)
    r = subprocess.run([exe, str(nd)] + ["0"] * nd, capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0
    assert f"ok nd={nd}" in r.stdout and "two token indices must differ" in r.stdout


@pytest.mark.parametrize("nd,threads", [(2, 1), (3, 0), (8, 1)])
def test_mixed_families_sharded_equals_oracle(nd, threads):
    n = 96
    batches = [synth.product_pools(70_001, n, seed=1), synth.geomean_pools(30_000, n, seed=2),
               synth.univ3_pools(9_000, n, 7, seed=3), synth.product_pools(5, n, seed=4)]   # 5 pools over 8 shards: empties
    v = synth.sweep_prices(n, seed=9)
    be = cr.DeviceBackend(n, batches, device=[0] * nd)
    be.ctx.set_option("multi_threads", threads)
    try:
        assert be.ctx.device_count == nd and be.ctx.pool_count == sum(len(b) for b in batches)
        psi, acc = be.find_arb(v)
        D, L = be.trades()
        Do, Lo, psi_o, acc_o = oracle_sweep(batches, n, v)
        assert rel_to_max(psi, psi_o) <= 1e-12 and abs(acc - acc_o) <= 1e-12 * abs(acc_o)
        g0, g1 = 70_001, 100_001
        np.testing.assert_array_equal(D[:g0], Do[:g0])                 # ProductTwoCoin: bit-exact
        np.testing.assert_array_equal(L[:g0], Lo[:g0])
        np.testing.assert_array_equal(D[g1:], Do[g1:])                 # UniV3 (CSR rebased per shard) + the 5-pool batch
        np.testing.assert_array_equal(L[g1:], Lo[g1:])
        scale = np.maximum(batches[1].R.max(axis=1), 1.0)[:, None]
        assert np.max(np.abs(D[g0:g1] - Do[g0:g1]) / scale) <= 1e-12   # GeometricMean: log-space forms
        segs = be.ctx.segments()
        assert [s["m"] for s in segs] == [len(b) for b in batches]
        Dw, Lw = be.ctx.trades_range(2, 1000, 5000)
        np.testing.assert_array_equal(Dw, Do[g1 + 1000:g1 + 6000])
        psi2, acc2 = be.eval(v)                                        # fused evaluation: same bits
        np.testing.assert_array_equal(psi2, psi)
        assert acc2 == acc
        with pytest.raises(RuntimeError):                              # eval does not produce trades
            be.ctx.trades_range(2, 0, 1)
    finally:
        be.close()


def test_config4_shape_eight_shards_route_parity():
    """BASELINE config 4 in shape (ProductTwoCoin, 512 tokens, 8 shards), 1/10 of its size per shard so
    that the oracle's route! stays in the test budget: ONE cfmm_route call drives all shards."""
    n, m = 512, 400_000
    batches = [synth.product_pools(m, n, seed=1234)]
    obj = cr.LinearNonnegative(synth.linear_prices(n, seed=1234))
    r8 = cr.Router(obj, batches, n, device=[0] * 8)
    r1 = cr.Router(obj, batches, n, device=0)
    try:
        cr.route_(r8, v=np.ones(n), solver="native")
        cr.route_(r1, v=np.ones(n), solver="native")
        ref = orc.route_oracle(oracle_objective(obj), oracle_poolset(batches, n), v0=np.ones(n), nthreads=orc.lib().oracle_max_threads())
        scale = np.max(np.abs(ref["psi"]))
        assert np.max(np.abs(cr.netflows(r8) - ref["psi"])) <= 1e-6 * scale
        assert np.max(np.abs(cr.netflows(r8) - cr.netflows(r1))) <= 1e-6 * scale
        assert r8.info["funcalls"] >= 5
        # trades at v*: the sharded router's equal a plain sweep of the whole market at the same v
        D, L, _, _ = oracle_sweep(batches, n, r8.v, nthreads=orc.lib().oracle_max_threads())
        np.testing.assert_array_equal(r8.Δs, D)
        np.testing.assert_array_equal(r8.Λs, L)
    finally:
        r8.close()
        r1.close()


def test_config4_at_full_size_eight_shards():
    """VERDICT r5 item 2 / BASELINE config 4 AT ITS STATED SIZE (4M ProductTwoCoin pools, 512 tokens, 8 shards of 500k --
    /root/reference/src/router.jl:38-42 over 4M pools, :111-119) through ONE multi-device context with the device listed 8
    times: all 4M trade rows bit-equal to the CPU restatement at fixed prices, Ψ and the dual value <= 1e-12, and route!
    (one cfmm_route call driving the 8 shards) within north_star's 1e-6 of the Fortran L-BFGS-B 3.0 run on the restatement
    of this market (tests/golden/route_fortran.npz: full_config4)."""
    from benchlib.workloads import build_market, objective_for, sweep_prices_for
    n = 512
    batches = build_market("config4", 0, 1, "weak")
    assert sum(len(b) for b in batches) == 4_000_000
    obj = objective_for("config4", n)
    v = sweep_prices_for("config4", n)
    threads = orc.lib().oracle_max_threads()
    r8 = cr.Router(obj, batches, n, device=[0] * 8)
    try:
        assert r8._backend.ctx.device_count == 8
        cr.find_arb_(r8, v)
        D, L, psi_o, acc_o = oracle_sweep(batches, n, v, nthreads=threads)
        np.testing.assert_array_equal(r8.Δs, D)                   # 4M rows, bit for bit
        np.testing.assert_array_equal(r8.Λs, L)
        assert rel_to_max(cr.netflows(r8), psi_o) <= 1e-12 and abs(r8._acc - acc_o) <= 1e-12 * abs(acc_o)
        del D, L
        cr.route_(r8, v=np.ones(n), solver="native")
        g = np.load(os.path.join(ROOT, "tests", "golden", "route_fortran.npz"))
        psi_f = g["full_config4_psi"]
        gap = float(np.max(np.abs(cr.netflows(r8) - psi_f)) / np.max(np.abs(psi_f)))
        print(f"config 4 (4M pools, 8 shards): route! vs Fortran L-BFGS-B {gap:.2e} (that run's pool-order slack "
              f"{float(g['full_config4_slack']):.2e}), {r8.info['funcalls']} evaluations (Fortran {int(g['full_config4_evaluations'])})")
        assert gap <= 1e-6 and r8.info["funcalls"] >= 5
        D, L, _, _ = oracle_sweep(batches, n, r8.v, nthreads=threads)   # trades at v*: the reference's find_arb!(r, v*)
        np.testing.assert_array_equal(r8.Δs, D)
        np.testing.assert_array_equal(r8.Λs, L)
    finally:
        r8.close()


def test_polish_on_a_multi_device_parent_equals_the_single_device_polish():
    """cfmm_polish drives a multi-device parent like cfmm_route does (host-pointer sweeps over all shards): from the same
    route! both reach the same converged point as one device."""
    n = 64
    market = [synth.bounded_product_pools(90_000, n, seed=41, consistent=True)]
    obj = cr.BasketLiquidation(1, synth.basket(n, seed=41))
    out = {}
    for dev in (0, [0, 0, 0]):
        r = cr.Router(obj, market, n, device=dev)
        cr.route_(r, solver="native")
        cr.polish_(r)                              # native: one C-ABI call
        out[str(dev)] = (cr.netflows(r).copy(), r.v.copy(), dict(r.info["polish"]))
        r.close()
    (p1, v1, i1), (p3, v3, i3) = out["0"], out["[0, 0, 0]"]
    assert rel_to_max(p3, p1) <= 1e-12 and np.max(np.abs(v3 - v1) / v1) <= 1e-12
    assert i3["residual"] <= 1e-9 * np.max(np.abs(p1)) and "native_seconds" in i3


def test_failed_upload_rolls_back_every_shard():
    """UniV3 batches are validated by the shards themselves: a bad pool in the LAST shard's block must
    undo the blocks the earlier shards already stored."""
    n = 16
    good = synth.univ3_pools(4000, n, 5, seed=5)
    be = cr.DeviceBackend(n, [good], device=[0, 0, 0, 0])
    try:
        bad = synth.univ3_pools(4000, n, 5, seed=6)
        bad.liquidity[-2] = -1.0                      # last pool -> last shard
        from cfmmrouter_amd.cfmms import _upload
        with pytest.raises(cr.ArgumentError, match="shard 3"):
            _upload(be.ctx, bad)
        assert be.ctx.pool_count == 4000 and len(be.ctx.segments()) == 1
        v = synth.sweep_prices(n, seed=7)
        psi, acc = be.find_arb(v)
        D, L, psi_o, acc_o = oracle_sweep([good], n, v)
        assert rel_to_max(psi, psi_o) <= 1e-12
        np.testing.assert_array_equal(be.trades()[0], D)
    finally:
        be.close()


def test_multi_context_rejects_device_pointer_calls():
    be = cr.DeviceBackend(8, [synth.product_pools(100, 8, seed=1)], device=[0, 0])
    try:
        with pytest.raises(NotImplementedError):
            be.ctx.set_stream(0)
        with pytest.raises(NotImplementedError):
            be.ctx.sweep_dev(0x1000, 0x2000, False)
        with pytest.raises(cr.ArgumentError):
            cr.DeviceBackend(8, [], device=[0, 99])
    finally:
        be.close()


@pytest.mark.parametrize("world", [2, 4, 8])
def test_peer_gather_protocol_with_several_ranks_on_one_gpu(world):
    """The sharded fold + all-reduce launch (cfmm_set_peers, reduce_gather) at world > 1 on ONE GPU
    (tests/peer_ranks_worker.py): every "rank" is a context with its own stream and shard; ranks wait for
    each other's granules exactly as over xGMI.  Run in a subprocess with one hardware queue per rank
    (GPU_MAX_HW_QUEUES): ranks that shared a queue would serialise behind each other's waiting launch --
    an artefact of emulating several GPUs on one, not of the protocol."""
    env = dict(os.environ, GPU_MAX_HW_QUEUES=str(2 * world), CFMM_AMD_PEER_TIMEOUT_S="10")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "peer_ranks_worker.py"), str(world)],
                       capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    print(r.stdout[-2000:], r.stderr[-2000:])
    assert r.returncode == 0 and f"PEER_RANKS_OK world={world}" in r.stdout
