"""N>1 path on CPU: world_size-2 gloo run of the sharded router.  The rank-local sweep is the
test-only OracleBackend (no GPU here); what is under test is the sharding, the single
all-reduce per evaluation and the lockstep route loop of cfmmrouter.jl_amd/dist.py."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist

    import cfmmrouter_amd as cr
    from cfmmrouter_amd import dist as crd
    from cfmmrouter_amd import synth
    from helpers import OracleBackend

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    n = 24
    market = [synth.product_pools(2001, n, seed=5), synth.geomean_pools(777, n, seed=6),
              synth.univ3_pools(301, n, 3, seed=7)]
    obj = cr.LinearNonnegative(synth.linear_prices(n, seed=5))
    r = crd.ShardedRouter(obj, market, n, _local_backend_factory=lambda nt, b: OracleBackend(nt, b))
    v = synth.sweep_prices(n, seed=8)
    cr.find_arb_(r, v)
    psi_fixed = cr.netflows(r).copy()
    cr.route_(r, v=np.ones(n))
    q.put((rank, len(r.Δs), psi_fixed, cr.netflows(r).copy(), r.v.copy(), r.n_sweeps))
    dist.destroy_process_group()


def test_shard_ranges_cover_exactly_once():
    sys.path.insert(0, ROOT)
    from cfmmrouter_amd.dist import shard_range
    for m in (0, 1, 7, 8, 1000, 4_000_001):
        for world in (1, 2, 3, 8):
            spans = [shard_range(m, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == m
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


@pytest.mark.timeout(300)
def test_two_rank_gloo_matches_single_process():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import cfmmrouter_amd as cr
    from cfmmrouter_amd import synth
    from helpers import OracleBackend, rel_to_max

    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    n = 24
    market = [synth.product_pools(2001, n, seed=5), synth.geomean_pools(777, n, seed=6),
              synth.univ3_pools(301, n, 3, seed=7)]
    obj = cr.LinearNonnegative(synth.linear_prices(n, seed=5))
    single = cr.Router(obj, market, n, _backend=OracleBackend(n, market))
    cr.find_arb_(single, synth.sweep_prices(n, seed=8))
    psi_fixed = cr.netflows(single).copy()
    cr.route_(single, v=np.ones(n))

    assert sum(r[1] for r in res) == 2001 + 777 + 301          # every pool on exactly one rank
    for rank, m_loc, pf, psi, v, sweeps in res:
        assert rel_to_max(pf, psi_fixed) <= 1e-13               # fixed-v netflows: shard + all-reduce == full sweep
        assert rel_to_max(psi, cr.netflows(single)) <= 1e-6     # route!: north_star tolerance
        np.testing.assert_allclose(v, single.v, rtol=1e-7)
    np.testing.assert_array_equal(res[0][3], res[1][3])         # ranks are bit-identical (lockstep)
    np.testing.assert_array_equal(res[0][4], res[1][4])
    assert res[0][5] == res[1][5]


def _guard_worker(rank, world, port, q):
    """PeerGuard's vote / resync and the collective retry of router.py::_route_native at world 2 over gloo, with a stand-in
    for the device context (there is no GPU here): rank 1's first route! fails, BOTH ranks must learn it, re-align the
    exchange's sequence number on the maximum + 2, switch pre-arming off and repeat; a second failure is raised on both."""
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch
    import torch.distributed as dist

    import cfmmrouter_amd as cr
    from cfmmrouter_amd import dist as crd
    from cfmmrouter_amd import synth

    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    n = 6

    class FakeCtx:                      # what _route_native and PeerGuard touch of a Context
        def __init__(self):
            self.fail = 0
            self.calls, self.options, self.peer_seq, self.set_peers_calls = 0, {}, 10 + 5 * rank, []

        def route(self, kind, vec, idx, **kw):
            self.calls += 1
            self.peer_seq += 3          # a route! performs sharded sweeps
            if self.fail > 0:
                self.fail -= 1
                raise RuntimeError("injected failure (test)")
            return np.full(n, 2.0), np.arange(n, dtype=np.float64), {"sweeps": 3, "f": 1.0, "evaluations": 2, "iterations": 1,
                                                                     "status": 0, "sweep_seconds": 0.0, "total_seconds": 0.0}

        def get_option(self, key):
            return self.peer_seq if key == "peer_seq" else self.options.get(key, 1)

        def set_option(self, key, value):
            self.options[key] = value

        def set_peers(self, ptrs, w, r_, seq):
            self.set_peers_calls.append((list(ptrs), w, r_, seq))
            self.peer_seq = seq

        def dual_value(self):
            return 0.5

    class Peers:
        ptrs = [111, 222]

    be = object.__new__(cr.DeviceBackend)       # a DeviceBackend shell around the stand-in context (no device is touched)
    be.ctx, be.n_tokens = FakeCtx(), n
    pools = [synth.product_pools(4, n, seed=3)]
    r = cr.Router(cr.LinearNonnegative(np.ones(n)), pools, n, _backend=be)
    r._guard = crd.PeerGuard(be.ctx, Peers(), world, rank, None, torch.device("cpu"))
    out = {}
    # (1) nobody fails: one call, no retry
    cr.route_(r, solver="native")
    out["plain"] = (be.ctx.calls, getattr(r, "collective_retries", 0))
    # (2) rank 1 fails once: both ranks retry once, sequence numbers re-aligned on max + 2, armed switched off
    be.ctx.calls = 0
    be.ctx.fail = 1 if rank == 1 else 0
    cr.route_(r, solver="native")
    out["retry"] = (be.ctx.calls, r.collective_retries, be.ctx.set_peers_calls[-1], be.ctx.options.get("armed"))
    # (3) rank 1 fails twice: raised on BOTH ranks
    be.ctx.fail = 2 if rank == 1 else 0
    try:
        cr.route_(r, solver="native")
        out["fatal"] = "no error"
    except RuntimeError as e:
        out["fatal"] = str(e)
    q.put((rank, out))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_collective_retry_of_a_failed_sharded_route():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_guard_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank in (0, 1):
        assert res[rank]["plain"] == (1, 0)
        calls, retries, last_set_peers, armed = res[rank]["retry"]
        assert calls == 2 and retries == 1 and armed == 0
        # before the vote: rank 0 at 10 + 3 + 3 = 16, rank 1 at 15 + 3 + 3 = 21 -> both continue from 21 + 2
        assert last_set_peers == ([111, 222], 2, rank, 23)
    assert "injected failure" in res[1]["fatal"] and "another rank" in res[0]["fatal"]
