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
