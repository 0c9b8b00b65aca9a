"""Sharding the pool sweep across GPUs (one process per GPU, torch.distributed; "nccl" == RCCL).

The reference has no distributed path; its only parallel axis is the pool index
(`Threads.@threads for i in 1:length(r.Δs)`, src/router.jl:39).  Pools are independent given v, and
the only coupling is the sum over pools (Ψ and the dual scalar), so:

  * every rank stores a contiguous block of each pool family (no pool is replicated),
  * v (n_tokens doubles) is replicated -- every rank runs the same O(n_tokens) L-BFGS-B step,
  * per evaluation there is exactly ONE collective: all-reduce(sum) of the n_tokens+1 doubles
    {Ψ, acc}.  At n_tokens = 512 that is 4104 B: latency-bound on xGMI, never bandwidth-bound, so
    it is issued as a single small all-reduce on the sweep's stream (no bucketing to tune).

An all-reduce leaves bit-identical values on every rank, so all ranks take identical L-BFGS-B
steps and stay in lockstep without any further synchronisation.
"""
from __future__ import annotations

import numpy as np

from .cfmms import PoolBatch
from .router import DeviceBackend, Router, _segments_of


class IpcPeers:
    """The symmetric buffers of cfmm_set_peers through the LIBRARY's own IPC export
    (cfmm_peer_buffer_alloc / _open: hipIpcGetMemHandle / hipIpcOpenMemHandle) -- no private torch
    API; torch.distributed only carries the 64-byte handles (any launcher's channel would do).
    `ptrs[p]` is rank p's buffer mapped into this process (own rank: the allocation itself).

    Every rank takes part in every collective of the set-up whatever its local outcome (a rank that
    fails locally must not walk into the next collective while the others are still in this one):
    `ok` tells whether THIS rank has all mappings; open_peer_buffers votes on it."""

    def __init__(self, ctx, group):
        import torch.distributed as dist

        self.ctx, self.group = ctx, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.own, self.ptrs, self.ok = None, [], True
        handle = None
        try:
            self.own, handle = ctx.peer_buffer_alloc()
        except Exception:
            self.ok = False
        handles = [None] * self.world
        dist.all_gather_object(handles, handle, group=group)     # a failed rank contributes None
        if self.ok and all(h is not None for h in handles):
            try:
                for p in range(self.world):
                    self.ptrs.append(self.own if p == self.rank else ctx.peer_buffer_open(handles[p]))
            except Exception:
                self.ok = False
        else:
            self.ok = False
        dist.barrier(group=group)       # every rank has mapped every buffer (or given up) before anybody writes

    def close(self, collective=True):
        """Unmap the peers' buffers and free the own one.  collective=True: all ranks call this together."""
        import torch.distributed as dist
        try:
            self.ctx.set_peers([], 0, 0, 0)
        except Exception:
            pass
        for p, ptr_ in enumerate(self.ptrs):
            if p != self.rank:
                try:
                    self.ctx.peer_buffer_close(ptr_)
                except Exception:
                    pass
        self.ptrs = []
        if collective:
            try:
                dist.barrier(group=self.group)   # nobody still maps the buffer that is about to be freed
            except Exception:
                pass
        if self.own is not None:
            try:
                self.ctx.peer_buffer_free(self.own)
            except Exception:
                pass
            self.own = None


def open_peer_buffers(ctx, group, device):
    """Symmetric buffers for `ctx` on every rank of `group` through the library's IPC export, or None on ALL
    ranks (collective vote) -- callers then all-reduce through torch.distributed (RCCL).
    Returns an object with `.ptrs` and `.close()`."""
    import torch
    import torch.distributed as dist

    peers = IpcPeers(ctx, group)          # takes part in all its collectives even when it fails locally
    t = torch.tensor([1.0 if peers.ok else 0.0], dtype=torch.float64,
                     device=device if dist.get_backend(group) == "nccl" else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    if float(t.item()) == 1.0:
        return peers
    peers.close(collective=True)          # every rank is here: the barrier inside is matched
    return None


def join_library_rccl(ctx, group, device):
    """Puts `ctx` into sharded operation through the LIBRARY's RCCL entry points (include/cfmm_amd.h: cfmm_rccl_unique_id /
    cfmm_rccl_init_rank): behind every sweep's fold the library itself enqueues ncclAllReduce(n_tokens + 1 doubles) on the
    context's stream -- north_star's collective, with no torch in the evaluation loop (torch.distributed only carries the
    128-byte id here, as any launcher's channel would).  Collective: True on ALL ranks or False on all ranks.
    Failure semantics are RCCL's (include/cfmm_amd.h): no PeerGuard, no vote, no time limit of the library's own -- a rank
    that dies between two evaluations leaves the others in their stream synchronisation until the launcher's watchdog
    (torchrun's, NCCL_ASYNC_ERROR_HANDLING) tears the job down.  The peer exchange (tried first) bounds that case itself."""
    import torch
    import torch.distributed as dist

    rank, world = dist.get_rank(group), dist.get_world_size(group)
    vote_dev = device if dist.get_backend(group) == "nccl" else "cpu"

    def all_ok(flag):
        t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float64, device=vote_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        return float(t.item()) == 1.0

    uid = None
    try:
        uid = ctx.rccl_unique_id()          # every rank: proves RCCL resolves in this process BEFORE anybody enters the init
    except Exception:
        uid = None
    if not all_ok(uid is not None):
        return False
    box = [uid]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    ok = True
    try:
        ctx.rccl_init_rank(box[0], world, rank)     # collective inside RCCL: every rank got here (the vote above)
    except Exception:
        ok = False
    if all_ok(ok):
        return True
    try:
        ctx.set_rccl_comm(None)
    except Exception:
        pass
    return False


class PeerGuard:
    """Collective safety net of a router whose all-reduce lives inside the library (cfmm_set_peers).

    * `self_check()` -- right after the mappings are opened, before anything depends on them: a few sharded sweeps at a
      synthetic price vector against the same sweeps all-reduced through torch.distributed (RCCL); every rank votes
      (MIN) and ALL ranks keep or drop the in-library exchange together.  What no single-GPU test can show -- that
      fine-grained stores become visible to a peer's system-scope loads over xGMI the way they do within one GPU -- is
      thereby checked on the machine the router runs on, and a disagreement costs the fast path, not the route.
    * `vote(ok)` / `resync()` -- after a native route!: if ANY rank's call failed (a lost pre-armed hand-over, a peer
      that did not publish in time), all ranks re-align their exchange sequence numbers on the maximum, switch
      pre-arming off and repeat the route launch-when-ready; only a second collective failure is raised."""

    def __init__(self, ctx, peers, world, rank, group, device):
        import torch
        import torch.distributed as dist
        self._torch, self._dist = torch, dist
        self.ctx, self.peers, self.world, self.rank, self.group = ctx, peers, world, rank, group
        self.device = device if dist.get_backend(group) == "nccl" else torch.device("cpu")
        self.cuda = device

    def vote(self, ok: bool) -> bool:
        t = self._torch.tensor([1.0 if ok else 0.0], dtype=self._torch.float64, device=self.device)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MIN, group=self.group)
        return float(t.item()) == 1.0

    def resync(self):
        t = self._torch.tensor([float(self.ctx.get_option("peer_seq"))], dtype=self._torch.float64, device=self.device)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX, group=self.group)
        self.ctx.set_peers(self.peers.ptrs, self.world, self.rank, int(t.item()) + 2)   # stale granules carry older tags

    def self_check(self, n_tokens, sweeps=3, rtol=1e-12):
        import os
        torch, dist = self._torch, self._dist
        # Every rank runs the SAME sequence of collectives whatever happens locally: an iteration that raises
        # still takes part in that iteration's all-reduce (on a NaN-filled tensor), so no rank ever sits in the SUM
        # all-reduce of n_tokens + 1 doubles while another one is already in the vote's MIN all-reduce of one.
        ok = True
        out = ref = stream = None
        try:
            with torch.cuda.device(self.cuda):
                j = torch.arange(n_tokens, dtype=torch.float64, device=self.cuda)
                out = torch.zeros(n_tokens + 1, dtype=torch.float64, device=self.cuda)
                ref = torch.zeros(n_tokens + 1, dtype=torch.float64, device=self.cuda)
                stream = torch.cuda.current_stream(self.cuda)
                self.ctx.set_stream(stream.cuda_stream)
        except Exception:
            ok = False
        for k in range(sweeps):
            r = None
            try:
                if not ok:
                    raise RuntimeError("set-up failed")
                with torch.cuda.device(self.cuda):
                    v = torch.exp(0.2 * torch.sin(j * (0.7 + k)))            # positive, off the no-arbitrage manifold
                    self.ctx.set_peers(self.peers.ptrs, self.world, self.rank, k)
                    self.ctx.sweep_dev(v.data_ptr(), out.data_ptr(), False)
                    self.ctx.set_peers([], 0, 0, 0)
                    self.ctx.sweep_dev(v.data_ptr(), ref.data_ptr(), False)
                    stream.synchronize()
                    r = ref.cpu() if self.device.type == "cpu" else ref
            except Exception:
                ok = False
                r = torch.full((n_tokens + 1,), float("nan"), dtype=torch.float64, device=self.device)
            try:
                dist.all_reduce(r, group=self.group)        # the iteration's collective: always, on every rank
                if ok:
                    r = r.to(self.cuda)
                    ok = bool(torch.isfinite(out).all()) and bool(torch.isfinite(r).all()) and \
                        float((out - r).abs().max()) <= rtol * max(float(r.abs().max()), 1e-300)
            except Exception:
                ok = False
        try:
            self.ctx.reset_stream()
        except Exception:
            pass
        if os.environ.get("CFMM_AMD_PEER_SELFTEST_FAIL") == str(self.rank):   # test hook: this rank reports a disagreement
            ok = False
        return self.vote(ok), sweeps


def shard_range(m: int, rank: int, world: int):
    """Contiguous block [lo, hi) of m items owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(int(m), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batches(batches, rank: int, world: int):
    """This rank's contiguous slice of every homogeneous batch."""
    out = []
    for b in batches:
        lo, hi = shard_range(len(b), rank, world)
        out.append(b.slice(lo, hi))
    return out


class ShardedBackend:
    """A rank-local backend plus the per-evaluation all-reduce of {Ψ, acc} through torch.distributed
    (RCCL on GPUs, gloo in the CPU tests) -- the fall-back of ShardedRouter; its fast path keeps the
    collective inside the library (cfmm_set_peers: the fold launch gathers over xGMI)."""

    def __init__(self, local, group=None):
        import torch
        import torch.distributed as dist

        self._torch, self._dist = torch, dist
        self.local = local
        self.group = group
        self.n_tokens = local.n_tokens
        self._on_device = isinstance(local, DeviceBackend) and torch.cuda.is_available()
        if self._on_device:
            dev = torch.device("cuda", local.ctx.device)
            self._v = torch.empty(self.n_tokens, dtype=torch.float64, device=dev)
            self._out = torch.empty(self.n_tokens + 1, dtype=torch.float64, device=dev)
            self._v_pin = torch.empty(self.n_tokens, dtype=torch.float64).pin_memory()
            self._out_pin = torch.empty(self.n_tokens + 1, dtype=torch.float64).pin_memory()
            self._dev = dev
            self._stream = torch.cuda.Stream(device=dev)   # sweep, all-reduce and copies share it

    def _reduce_host(self, psi, acc):
        t = self._torch.from_numpy(np.concatenate([psi, [acc]]))
        self._dist.all_reduce(t, group=self.group)
        out = t.numpy()
        return out[:-1].copy(), float(out[-1])

    def _sweep(self, v, materialize):
        if not self._on_device:
            psi, acc = (self.local.find_arb if materialize else self.local.eval)(v)
            return self._reduce_host(psi, acc)
        torch = self._torch
        with torch.cuda.device(self._dev), torch.cuda.stream(self._stream):
            stream = self._stream
            self.local.ctx.set_stream(stream.cuda_stream)
            self._v_pin.copy_(torch.from_numpy(np.ascontiguousarray(v, dtype=np.float64)))
            self._v.copy_(self._v_pin, non_blocking=True)
            self.local.ctx.sweep_dev(self._v.data_ptr(), self._out.data_ptr(), materialize)
            self._dist.all_reduce(self._out, group=self.group)   # RCCL, 8·(n_tokens+1) bytes
            self._out_pin.copy_(self._out, non_blocking=True)
            stream.synchronize()
        out = self._out_pin.numpy()
        if not np.all(np.isfinite(out)):
            raise RuntimeError("sharded sweep returned a non-finite {Ψ, acc}: a rank did not publish within the peer "
                               "all-reduce's time limit (CFMM_AMD_PEER_TIMEOUT_S), or a shard overflowed")
        return out[:-1].copy(), float(out[-1])

    def eval(self, v):
        return self._sweep(v, False)

    def find_arb(self, v):
        return self._sweep(v, True)

    def trades(self):
        """The LOCAL shard's trades (Δ, Λ), segment order."""
        if self._on_device:
            self._stream.synchronize()
        return self.local.trades()

    def close(self):
        if hasattr(self.local, "close"):
            self.local.close()


def ShardedRouter(objective, cfmms, n_tokens, rank=None, world=None, device=None, group=None,
                  already_sharded=False, _local_backend_factory=None, self_check=True):
    """Router(objective, cfmms, n_tokens) whose pools are block-partitioned over the ranks of a
    torch.distributed process group.  `cfmms` is the FULL market on every rank (or this rank's
    shard with already_sharded=True).  r.Δs / r.Λs / r.cfmms describe the local shard; r.v and
    netflows(r) are global and identical on all ranks.  r.collective says which all-reduce the router ended up with:
    the in-library peer exchange (after its start-up check against torch.distributed, `PeerGuard.self_check`) or the
    torch.distributed fall-back."""
    import torch.distributed as dist

    rank = dist.get_rank(group) if rank is None else rank
    world = dist.get_world_size(group) if world is None else world
    if not isinstance(cfmms, PoolBatch):
        cfmms = list(cfmms)
    batches, _, host = _segments_of(cfmms)
    if host:
        from ._lib import ArgumentError
        raise ArgumentError("ShardedRouter: pools without a device kernel (host-evaluated CFMM subclasses) are not sharded; "
                            "use Router on one GPU")
    local = batches if already_sharded else shard_batches(batches, rank, world)
    if _local_backend_factory is not None:
        backend = _local_backend_factory(n_tokens, local)
    else:
        import torch
        dev = rank if device is None else device
        backend = DeviceBackend(n_tokens, local, device=dev)
        import os
        if torch.cuda.is_available() and n_tokens <= 8192 and os.environ.get("CFMM_AMD_NO_PEER", "0") != "1":
            # fast path: the all-reduce happens INSIDE the library, in the launch that folds the partial rows
            # (cfmm_set_peers), so find_arb_/route_ -- including the one-call native route! -- work on the
            # global market unchanged.  Falls through to ShardedBackend (torch.distributed all-reduce) when no
            # rank-to-rank mapping can be set up.
            with torch.cuda.device(dev):
                peers = open_peer_buffers(backend.ctx, group, torch.device("cuda", dev))
            if peers is not None:
                # the exchange is checked against RCCL on THIS machine before anything depends on it (collective vote:
                # all ranks keep it or all drop it), and the router keeps the guard for collective retries of route!
                guard = PeerGuard(backend.ctx, peers, world, rank, group, torch.device("cuda", dev))
                good, done = guard.self_check(n_tokens) if self_check else (True, 0)
                if good:
                    backend.ctx.set_peers(peers.ptrs, world, rank, done)
                    backend.peer = peers   # keeps the mappings alive
                    r = Router(objective, local, n_tokens, _backend=backend)
                    r._guard = guard
                    r.collective = "peer all-reduce inside the library (cfmm_set_peers), checked against torch.distributed"
                    return r
                backend.ctx.set_peers([], 0, 0, 0)
                peers.close(collective=True)      # every rank is here (the vote was collective)
        if torch.cuda.is_available() and n_tokens <= 8192 and os.environ.get("CFMM_AMD_NO_LIB_RCCL", "0") != "1":
            # second choice: RCCL INSIDE the library (cfmm_rccl_init_rank): the fold launch is followed, in-stream, by
            # ncclAllReduce of the n_tokens + 1 doubles -- route!'s one-call native solver works unchanged, launch-when-ready
            with torch.cuda.device(dev):
                joined = join_library_rccl(backend.ctx, group, torch.device("cuda", dev))
            if joined:
                r = Router(objective, local, n_tokens, _backend=backend)
                r.collective = "RCCL all-reduce inside the library (cfmm_rccl_init_rank: ncclAllReduce behind every fold)"
                return r
    r = Router(objective, local, n_tokens, _backend=ShardedBackend(backend, group))
    r.collective = "all-reduce through torch.distributed (RCCL)"
    return r
