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


class PeerAllReduce:
    """One-shot all-reduce(sum) of {Ψ, acc} over xGMI peer mappings (csrc/peer_allreduce.hip).

    The buffers are a torch.distributed._symmetric_memory allocation (torch does the IPC handle
    exchange); the kernel is ours.  `slot()` is where this step's local {Ψ, acc} must be written
    (cfmm_sweep_dev targets it directly), `reduce(out)` then leaves the rank-ordered sum in `out` on
    every rank.  `PeerAllReduce.create` returns None whenever anything about the fast path is not
    available or does not reproduce RCCL's result on a self-test -- callers then use dist.all_reduce."""

    def __init__(self, count, group, device):
        import ctypes as C

        import torch
        import torch.distributed as dist
        import torch.distributed._symmetric_memory as symm_mem

        from ._lib import lib

        self._torch, self._C, self._lib = torch, C, lib()
        self.count = int(count)
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        # [2][count] doubles + 2 uint64 flags (this class's kernel) + [2][count][2] granules (cfmm_set_peers)
        words = 6 * self.count + 2
        self.buf = symm_mem.empty(words, dtype=torch.float64, device=device)
        self.buf.zero_()
        gname = (group or dist.group.WORLD).group_name
        self.hdl = symm_mem.rendezvous(self.buf, gname)
        ptrs = [int(p) for p in self.hdl.buffer_ptrs]
        if len(ptrs) != self.world:
            raise RuntimeError("symmetric memory returned an unexpected number of peers")
        self._ptrs = (C.c_uint64 * self.world)(*ptrs)
        self.seq = 0
        torch.cuda.synchronize(device)
        dist.barrier(group=group)                        # everybody's flags are zero before step 1 (plain RCCL
        torch.cuda.synchronize(device)                   # barrier: nothing here may spin on a peer mapping)

    def slot(self):
        """Device view [count] that the NEXT reduce() will read as this rank's contribution."""
        parity = (self.seq + 1) & 1
        return self.buf[parity * self.count:(parity + 1) * self.count]

    def reduce(self, out):
        self.seq += 1
        stream = self._torch.cuda.current_stream().cuda_stream
        rc = self._lib.cfmm_peer_allreduce(self._C.c_void_p(stream), self._ptrs, self.world, self.rank, self.count,
                                           self._C.c_uint64(self.seq), self._C.c_void_p(out.data_ptr()))
        if rc != 0:
            raise RuntimeError("cfmm_peer_allreduce launch failed")

    @staticmethod
    def create(count, group, device, checks=4):
        import torch
        import torch.distributed as dist
        try:
            par = PeerAllReduce(count, group, device)
            good = True
            out = torch.empty(count, dtype=torch.float64, device=device)
            for k in range(checks):   # self-test against RCCL before trusting the fast path
                g = torch.Generator(device="cpu").manual_seed(1000 * k + par.rank)
                x = torch.rand(count, dtype=torch.float64, generator=g).to(device) * (10.0 ** k)
                par.slot().copy_(x)
                par.reduce(out)
                ref = x.clone()
                dist.all_reduce(ref, group=group)
                torch.cuda.synchronize(device)
                scale = float(ref.abs().max())
                good = good and bool(torch.isfinite(out).all()) and float((out - ref).abs().max()) <= 1e-12 * scale
        except Exception:
            par, good = None, False
        flag = torch.tensor([1.0 if good else 0.0], dtype=torch.float64, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)   # all ranks or none
        return par if float(flag.item()) == 1.0 else None


class IpcPeers:
    """The symmetric buffers of cfmm_set_peers through the LIBRARY's own IPC export
    (cfmm_peer_buffer_alloc / _open: hipIpcGetMemHandle / hipIpcOpenMemHandle) -- no private torch
    API; torch.distributed only carries the 64-byte handles (any launcher's channel would do).
    `ptrs[p]` is rank p's buffer mapped into this process (own rank: the allocation itself)."""

    def __init__(self, ctx, group):
        import torch.distributed as dist

        self.ctx, self.group = ctx, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)
        self.own, handle = ctx.peer_buffer_alloc()
        handles = [None] * self.world
        dist.all_gather_object(handles, handle, group=group)
        self.ptrs = [self.own if p == self.rank else ctx.peer_buffer_open(handles[p]) for p in range(self.world)]
        dist.barrier(group=group)       # every rank has mapped every buffer before anybody writes

    def close(self):
        import torch.distributed as dist
        try:
            self.ctx.set_peers([], 0, 0, 0)
            for p, ptr_ in enumerate(self.ptrs):
                if p != self.rank:
                    self.ctx.peer_buffer_close(ptr_)
            dist.barrier(group=self.group)   # nobody still maps the buffer that is about to be freed
            self.ctx.peer_buffer_free(self.own)
        except Exception:
            pass
        self.ptrs = []


def open_peer_buffers(ctx, group, device):
    """Symmetric buffers for `ctx` on every rank of `group`, or None on ALL ranks (collective vote):
    first the library's own IPC export, then torch's symmetric memory (nccl groups only).
    Returns an object with `.ptrs` (and `.close()` for the IPC flavour)."""
    import torch
    import torch.distributed as dist

    def vote(ok):
        t = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64,
                         device=device if dist.get_backend(group) == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
        return float(t.item()) == 1.0

    peers = None
    try:
        peers = IpcPeers(ctx, group)
    except Exception:
        peers = None
    if vote(peers is not None):
        return peers
    if peers is not None:
        peers.close()
    if dist.get_backend(group) != "nccl":
        return None
    sym = PeerAllReduce.create(ctx.n_tokens + 1, group, device)   # votes internally
    if sym is None:
        return None
    sym.ptrs = [int(p) for p in sym.hdl.buffer_ptrs]
    return sym


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
                backend.ctx.set_peers(peers.ptrs, world, rank, 0)   # the granules are fresh
                backend.peer = peers   # keeps the mappings alive
                return Router(objective, local, n_tokens, _backend=backend)
    return Router(objective, local, n_tokens, _backend=ShardedBackend(backend, group))
