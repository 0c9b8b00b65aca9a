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
    """A rank-local backend plus the per-evaluation all-reduce of {Ψ, acc}."""

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
                  already_sharded=False, _local_backend_factory=None):
    """Router(objective, cfmms, n_tokens) whose pools are block-partitioned over the ranks of a
    torch.distributed process group.  `cfmms` is the FULL market on every rank (or this rank's
    shard with already_sharded=True).  r.Δs / r.Λs / r.cfmms describe the local shard; r.v and
    netflows(r) are global and identical on all ranks."""
    import torch.distributed as dist

    rank = dist.get_rank(group) if rank is None else rank
    world = dist.get_world_size(group) if world is None else world
    if not isinstance(cfmms, PoolBatch):
        cfmms = list(cfmms)
    batches, _ = _segments_of(cfmms)
    local = batches if already_sharded else shard_batches(batches, rank, world)
    if _local_backend_factory is not None:
        backend = _local_backend_factory(n_tokens, local)
    else:
        backend = DeviceBackend(n_tokens, local, device=rank if device is None else device)
    return Router(objective, local, n_tokens, _backend=ShardedBackend(backend, group))
