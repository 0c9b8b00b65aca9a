"""Soak test of the host <-> device hand-overs that are on by default (pre-armed route! evaluations through the PCIe
BAR, output granules): many thousand evaluations, every result compared bit for bit with a context that uses the plain
paths (armed = 0, host_flag = 0), every call bounded in time.  usage: python scripts/soak.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import cfmmrouter_amd as cr
from cfmmrouter_amd import synth
from cfmmrouter_amd._lib import OBJ_LINEAR_NONNEGATIVE

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
n = 96
batches = [synth.product_pools(40_000, n, seed=1), synth.geomean_pools(20_000, n, seed=2), synth.bounded_product_pools(15_000, n, seed=3),
           synth.univ3_ragged_pools(12_000, n, seed=4)]       # (round 5: multi-tick ladders too -- the threshold heads)
a, b = cr.DeviceBackend(n, batches), cr.DeviceBackend(n, batches)
b.ctx.set_option("armed", 0); b.ctx.set_option("host_flag", 0)
for be in (a, b):
    be.ctx.set_option("alternate", 0)      # identical summation order on every sweep
rng = np.random.default_rng(5)
t_end = time.perf_counter() + budget
evals = routes = 0
worst = 0.0
c = synth.linear_prices(n, seed=9)
while time.perf_counter() < t_end:
    for _ in range(200):
        v = np.exp(rng.uniform(-0.3, 0.3, n))
        t0 = time.perf_counter(); pa = a.eval(v); worst = max(worst, time.perf_counter() - t0)
        pb = b.eval(v)
        assert np.array_equal(pa[0], pb[0]) and pa[1] == pb[1], "eval mismatch"
        evals += 1
    for _ in range(20):
        v0 = np.exp(rng.uniform(-0.05, 0.05, n)) * np.maximum(c, 1.0)
        t0 = time.perf_counter(); ra = a.ctx.route(OBJ_LINEAR_NONNEGATIVE, c, 0, v0=v0); worst = max(worst, time.perf_counter() - t0)
        rb = b.ctx.route(OBJ_LINEAR_NONNEGATIVE, c, 0, v0=v0)
        assert np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1], rb[1]) and ra[2]["evaluations"] == rb[2]["evaluations"], "route mismatch"
        routes += 1
        evals += ra[2]["evaluations"]
# round 6: the single-block geometry (one family, <= 2048 pools: the block publishes {Ψ, acc} itself) under the same regime --
# pre-armed routes and granules straight from the sweep block, against the same geometry on the plain paths (launch when
# ready, stream wait + copy)
small = [synth.product_pools(1500, 24, seed=11)]
sa, sb_ = cr.DeviceBackend(24, small), cr.DeviceBackend(24, small)
sb_.ctx.set_option("armed", 0); sb_.ctx.set_option("host_flag", 0)
assert sa.ctx.segments()[0]["grid"] == 1
cs = synth.linear_prices(24, seed=12)
t_end = time.perf_counter() + budget / 3
small_routes = small_evals = 0
while time.perf_counter() < t_end:
    for _ in range(200):
        v = np.exp(rng.uniform(-0.3, 0.3, 24))
        t0 = time.perf_counter(); pa = sa.eval(v); worst = max(worst, time.perf_counter() - t0)
        pb = sb_.eval(v)
        assert np.array_equal(pa[0], pb[0]) and pa[1] == pb[1], "single-block eval mismatch"
        small_evals += 1
    for _ in range(50):
        v0 = np.exp(rng.uniform(-0.05, 0.05, 24)) * np.maximum(cs, 1.0)
        t0 = time.perf_counter(); ra = sa.ctx.route(OBJ_LINEAR_NONNEGATIVE, cs, 0, v0=v0); worst = max(worst, time.perf_counter() - t0)
        rb = sb_.ctx.route(OBJ_LINEAR_NONNEGATIVE, cs, 0, v0=v0)
        assert np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1], rb[1]), "single-block route mismatch"
        small_routes += 1
        small_evals += ra[2]["evaluations"]
sa.close(); sb_.close()
print(f"single-block soak ok: {small_evals} evaluations, {small_routes} routes, all bit-identical to the plain paths")
print(f"soak ok: {evals} evaluations, {routes} routes in {budget:.0f} s, all bit-identical to the plain paths; slowest call {1e3 * worst:.2f} ms")
a.close(); b.close()
