"""One rank's timed machinery (ShardBench) and the sharded route! leg of bench.py."""
import os
import time

import numpy as np
import torch
import torch.distributed as dist

import cfmmrouter_amd as cr
from cfmmrouter_amd._lib import KIND_GEOMEAN

from .workloads import (HBM_PEAK_GBS, L3_BYTES, WORKLOADS, alg_bytes, build_market, objective_for, ring_copies, sweep_prices_for,
                        touched_bytes)


class ShardBench:
    """One rank's timed machinery for one workload: backend, stream, peer buffers (N > 1), the step."""

    def __init__(self, args, name, scaling, rank, world, local_rank, use_dist, collective=None):
        """collective (sharded runs): which all-reduce of {Ψ, acc} the step uses --
             "peer"          the library's fold + one-shot xGMI peer gather in ONE launch (cfmm_set_peers),
             "rccl_library"  ncclAllReduce enqueued by the library behind every fold (cfmm_rccl_init_rank): north_star's collective,
             "rccl_torch"    torch.distributed.all_reduce on the sweep's stream (RCCL; gloo in the shared-GPU rehearsal),
             "auto"          the first of those three that works on THIS machine (every choice is a collective vote);
           None = args.collective.  `self.collective` says which one the steps use, `self.why_not` why an explicitly requested
           one is not available (the steps then fall back to rccl_torch so that no rank is left alone)."""
        self.args, self.name, self.rank, self.world, self.use_dist = args, name, rank, world, use_dist
        self.desc, self.n, _ = WORKLOADS[name]
        n = self.n
        self.batches = build_market(name, rank, world, scaling)
        self.m_rank = sum(len(b) for b in self.batches)
        self.v = sweep_prices_for(name, n)
        self.local_rank = local_rank
        self.be = cr.DeviceBackend(n, self.batches, device=local_rank)
        self.apply_options(self.be)
        self.stream = torch.cuda.Stream()          # the sweep, the RCCL all-reduce and the events share it
        torch.cuda.set_stream(self.stream)
        self.be.ctx.set_stream(self.stream.cuda_stream)
        self.v_t = torch.from_numpy(self.v).to("cuda")
        self.out_t = torch.zeros(n + 1, dtype=torch.float64, device="cuda")
        self.materialize = not args.fused
        self.peer, self.fused_peer, self.peer_ptrs, self.n_fused = None, False, None, 0
        self.steps_run = 0
        self.ring, self.ring_pos = None, 0
        self.lib_rccl = False
        want = collective or getattr(args, "collective", "auto")
        self.collective, self.why_not = ("none" if not use_dist else "rccl_torch"), None
        if use_dist and want in ("auto", "peer"):
            if os.environ.get("CFMM_AMD_NO_PEER", "0") == "1":
                self.why_not = "CFMM_AMD_NO_PEER=1"
            else:
                self.setup_peers()
                if self.fused_peer:
                    self.collective = "peer"
                elif self.why_not is None:
                    self.why_not = "no rank-to-rank mapping of the peer buffers (hipIpc) on this machine"
        if use_dist and self.collective != "peer" and want in ("auto", "rccl_library"):
            # the collective through the library's own RCCL entry points (cfmm_rccl_init_rank: ncclAllReduce behind every fold,
            # on the sweep's stream) -- what a Julia / C host gets; torch.distributed only carries the 128-byte id
            if dist.get_backend() != "nccl":
                self.why_not = "ranks share a GPU (rehearsal over gloo): RCCL needs one device per rank"
            else:
                from cfmmrouter_amd.dist import join_library_rccl
                self.lib_rccl = join_library_rccl(self.be.ctx, None, torch.device("cuda", local_rank))
                if self.lib_rccl:
                    self.collective = "rccl_library"
                else:
                    self.why_not = "cfmm_rccl_unique_id / cfmm_rccl_init_rank failed on a rank (collective vote)"
        self.fell_back = self.why_not if (want == "auto" and self.collective == "rccl_torch") else None   # auto: why the faster two are out
        if want == "auto" or self.collective == want:
            self.why_not = None

    def apply_options(self, be):
        for kv in self.args.opt:
            k, val = kv.split("=")
            be.ctx.set_option(k, int(val))

    def setup_peers(self):
        # N > 1 (or N = 1 under torchrun): the launch that folds the partial rows also all-reduces {Ψ, acc}
        # over xGMI peer mappings (cfmm_set_peers: one launch, rank-ordered sum, bit-identical on every
        # rank).  At start-up that path is checked against sweep + RCCL all-reduce on every rank; if it is
        # unavailable or disagrees anywhere, ALL ranks use the RCCL all-reduce instead.
        from cfmmrouter_amd.dist import open_peer_buffers
        be, world, rank = self.be, self.world, self.rank
        self.peer = open_peer_buffers(be.ctx, None, torch.device("cuda", self.local_rank))  # None (on every rank) -> RCCL
        if self.peer is None:
            return
        self.why_not = "the peer exchange disagreed with torch.distributed's all-reduce in the start-up check (collective vote)"
        self.peer_ptrs = list(self.peer.ptrs)
        good = True
        for _ in range(3):
            be.ctx.set_peers(self.peer_ptrs, world, rank, self.n_fused)
            be.ctx.sweep_dev(self.v_t.data_ptr(), self.out_t.data_ptr(), self.materialize)
            self.n_fused += 1
            got = self.out_t.clone()
            be.ctx.set_peers([], 0, 0, 0)
            be.ctx.sweep_dev(self.v_t.data_ptr(), self.out_t.data_ptr(), self.materialize)
            ref = self.out_t.clone()
            dist.all_reduce(ref)
            torch.cuda.synchronize()
            good = good and bool(torch.isfinite(got).all()) and \
                float((got - ref).abs().max()) <= 1e-12 * float(ref.abs().max())
        flag = torch.tensor([1.0 if good else 0.0], dtype=torch.float64, device="cuda")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        self.fused_peer = float(flag.item()) == 1.0
        if self.fused_peer:
            self.why_not = None
            be.ctx.set_peers(self.peer_ptrs, world, rank, self.n_fused)

    def step(self):
        if self.ring is not None:     # rotate over enough copies of the market to exceed the 256 MB Infinity Cache
            b_ = self.ring[self.ring_pos % len(self.ring)]
            self.ring_pos += 1
            b_.ctx.sweep_dev(self.v_t.data_ptr(), self.out_t.data_ptr(), self.materialize)
            return
        self.be.ctx.sweep_dev(self.v_t.data_ptr(), self.out_t.data_ptr(), self.materialize)   # sharded context: already the global {Ψ, acc}
        if self.use_dist and not self.fused_peer and not self.lib_rccl:
            dist.all_reduce(self.out_t)  # Ψ and the dual scalar: one small RCCL collective per evaluation
        self.steps_run += 1

    def touched_per_copy(self):
        """Bytes ONE sweep over this rank's shard moves across the L2 <-> fabric boundary, i.e. what has to fall out of the
        256 MiB Infinity Cache between two visits of the same market copy -- counted CONSERVATIVELY (a smaller figure means
        more copies): the packed layout's own bytes (workloads.touched_bytes: a lower bound that leaves out the scattered walk
        records of multi-tick UniV3 ladders), or 0.7 x the committed PMC figure of this workload (profiles/traffic.json:
        FETCH_SIZE doubled + WRITE_SIZE of the sweep launch) where that is larger (multi-tick ladders: 274 MB measured against
        88 MB by construction; building 8 copies of 17M ticks for nothing).  NOT the reference-layout bytes of SURVEY 8d (64 /
        80 B per pool), which round 4 used and which left three of five rings inside the cache."""
        tot = touched_bytes(self.batches, self.materialize)
        pmc = None
        try:
            import json
            tf = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "traffic.json")
            pmc = json.load(open(tf)).get(self.name + ("" if self.materialize else "_fused"))
            if pmc and self.world > 1:
                pmc = None            # the committed figure is the single-GPU market's
        except Exception:
            pmc = None
        return int(max(tot, 0.7 * pmc)) if pmc else int(tot)

    def market_copies(self):
        """(touched bytes per copy, copies): enough distinct copies of the market that the ring's TOUCHED bytes are at least
        twice the Infinity Cache -- whatever the replacement policy, a copy is gone from the cache when its turn comes again."""
        per_copy = self.touched_per_copy()
        return per_copy, ring_copies(per_copy)

    def use_ring(self):
        """--cold-only: the TIMED steps rotate over the ring of market_copies() (no collective: local sweeps)."""
        _, copies = self.market_copies()
        self.ring = [self.be] + [cr.DeviceBackend(self.n, self.batches, device=self.local_rank) for _ in range(copies - 1)]
        for b_ in self.ring[1:]:
            b_.ctx.set_stream(self.stream.cuda_stream)
            self.apply_options(b_)

    def timed_pass(self, steps, device_events=False):
        if self.use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        ev0 = ev1 = None
        if device_events:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        if device_events:
            ev0.record(self.stream)
        for _ in range(steps):
            self.step()
        if device_events:
            ev1.record(self.stream)
        while not self.stream.query():   # busy-wait for the last step (a blocking wait adds its wake-up latency to the K
            pass                         # steps: ~1 us per step at the driver's K = 20), then the synchronize of the contract
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0    # this rank's K steps are complete (with the collective inside every step no
        if self.use_dist:                # rank finishes step k before all ranks contributed to it); the closing
            dist.barrier()               # barrier follows the clock read, and the MAX over ranks is reported
        return dt, (ev0.elapsed_time(ev1) if device_events else None)

    def max_over_ranks(self, x):
        if not self.use_dist:
            return float(x)
        t = torch.tensor([x], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def kernel_pass(self, steps):
        """The same K steps again with a hipEvent pair attached to every kernel launch (start / stop written by the
        command processor, hipExtLaunchKernel) for the roofline; kept out of the timed region."""
        ctxs = self.ring if self.ring else [self.be]
        for b_ in ctxs:
            b_.ctx.set_option("time_kernels", 1)
            b_.ctx.kernel_times()  # reset
        elapsed2, _ = self.timed_pass(steps, device_events=True)
        kt = {"sweep_ms": 0.0, "reduce_ms": 0.0}
        for b_ in ctxs:
            kt_b = b_.ctx.kernel_times()
            kt["sweep_ms"] += kt_b["sweep_ms"]
            kt["reduce_ms"] += kt_b["reduce_ms"]
            b_.ctx.set_option("time_kernels", 0)
        return kt, elapsed2

    def cold_pass(self, steps):
        """HBM-resident figure: every working set here (<= 100 MB) fits the 256 MiB Infinity Cache, so the
        timed passes are "warm" (what a running route! sees).  Rotating LOCAL sweeps over enough distinct copies of this
        rank's shard that the ring's TOUCHED bytes are >= 2 x the cache (market_copies) makes every sweep read its pool
        state from HBM.  Every rank runs it (N > 1: the slowest rank's kernel time is reported)."""
        per_copy, copies = self.market_copies()
        extra = [cr.DeviceBackend(self.n, self.batches, device=self.local_rank) for _ in range(copies - (0 if self.lib_rccl else 1))]
        sharded = self.fused_peer
        if sharded:
            self.be.ctx.set_peers([], 0, 0, 0)
        ring = ([] if self.lib_rccl else [self.be]) + extra    # (a context with a library RCCL communicator does not sweep locally)
        outs = [torch.zeros(self.n + 1, dtype=torch.float64, device="cuda") for _ in ring]
        for b_ in extra:
            b_.ctx.set_stream(self.stream.cuda_stream)
            self.apply_options(b_)
        for k in range(2 * copies):
            ring[k % copies].ctx.sweep_dev(self.v_t.data_ptr(), outs[k % copies].data_ptr(), self.materialize)
        torch.cuda.synchronize()
        cold_steps = max(steps, 60)     # a stable average: at the driver's K = 20 the figure moves by +-0.02
        t0 = time.perf_counter()        # first WITHOUT kernel events: the HBM-resident step as the timed region would see it
        for k in range(cold_steps):
            ring[k % copies].ctx.sweep_dev(self.v_t.data_ptr(), outs[k % copies].data_ptr(), self.materialize)
        torch.cuda.synchronize()
        cold_plain = time.perf_counter() - t0
        for b_ in ring:
            b_.ctx.set_option("time_kernels", 1)
            b_.ctx.kernel_times()
        t0 = time.perf_counter()
        for k in range(cold_steps):
            ring[k % copies].ctx.sweep_dev(self.v_t.data_ptr(), outs[k % copies].data_ptr(), self.materialize)
        torch.cuda.synchronize()
        cold_elapsed = time.perf_counter() - t0
        sw = sum(b_.ctx.kernel_times()["sweep_ms"] for b_ in ring) / cold_steps
        for b_ in ring:
            b_.ctx.set_option("time_kernels", 0)
        for b_ in extra:
            b_.close()
        if sharded:
            self.be.ctx.set_peers(self.peer_ptrs, self.world, self.rank, self.n_fused + self.steps_run)
        sw = self.max_over_ranks(sw)
        ab = alg_bytes(self.batches, self.materialize, self.v)
        cold = {"copies": copies, "touched_per_copy": per_copy, "bytes_touched": copies * per_copy,
                "hbm_resident": bool(copies * per_copy >= 2 * L3_BYTES),
                "bytes_touched_is": "ring copies x bytes one sweep moves over the L2 <-> fabric boundary (PMC figure of this "
                                    "workload, or the packed layout's own bytes: a lower bound); >= 2 x 256 MiB Infinity Cache",
                "kernel_ms": sw,
                "achieved": ab / (sw * 1e-3) / 1e9 if sw > 0 else 0.0,
                "ms_per_step": 1e3 * cold_plain / cold_steps,
                "ms_per_step_with_kernel_events": 1e3 * cold_elapsed / cold_steps, "sweeps": cold_steps}
        cold["frac"] = cold["achieved"] / HBM_PEAK_GBS
        return cold

    def collective_check(self):
        """sharded runs: the timed path's global {Ψ, acc} against a plain RCCL all-reduce of the local ones"""
        self.step()
        got = self.out_t.clone()
        if self.fused_peer:
            self.be.ctx.set_peers([], 0, 0, 0)           # a LOCAL sweep for the reference
        local = self.be
        if self.lib_rccl:                                # the library's communicator is part of self.be: a plain context instead
            local = cr.DeviceBackend(self.n, self.batches, device=self.local_rank)
            local.ctx.set_stream(self.stream.cuda_stream)
            self.apply_options(local)
        local.ctx.sweep_dev(self.v_t.data_ptr(), self.out_t.data_ptr(), self.materialize)
        ref = self.out_t.clone()
        if local is not self.be:
            torch.cuda.synchronize()
            local.close()
        dist.all_reduce(ref)
        torch.cuda.synchronize()
        if self.fused_peer:
            self.be.ctx.set_peers(self.peer_ptrs, self.world, self.rank, self.n_fused + self.steps_run)
        self.out_t.copy_(got)
        return float((got - ref).abs().max() / ref.abs().max())

    def sharding_text(self):
        if not self.use_dist:
            return "single GPU, no collective"
        if self.fused_peer:
            return (f"pools x{self.world}, fold + one-shot xGMI peer all-reduce of n_tokens+1 f64 in one launch per step "
                    f"(buffers: library IPC export)")
        if self.lib_rccl:
            return (f"pools x{self.world}, RCCL all-reduce of n_tokens+1 f64 per step INSIDE the library (cfmm_rccl_init_rank: "
                    f"ncclAllReduce enqueued behind the fold on the sweep's stream)")
        return f"pools x{self.world}, all-reduce of n_tokens+1 f64 per step through torch.distributed ({dist.get_backend()})"

    def close(self):
        if self.ring:
            for b_ in self.ring[1:]:
                b_.close()
        if self.peer is not None and hasattr(self.peer, "close"):
            self.peer.close()
        self.be.close()


def sharded_route(sb, local_rank):
    """sharded route!: every rank drives the same L-BFGS-B on the all-reduced {Ψ, acc} of its own shard"""
    def all_ok(flag):   # collective vote, so that no rank walks into a collective alone
        t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return float(t.item()) == 1.0

    sr, err, out, psi, v_star, hidden = None, None, None, None, None, {}
    try:
        from cfmmrouter_amd import dist as crd
        obj = objective_for(sb.name, sb.n)
        v0 = np.ones(sb.n) if isinstance(obj, cr.LinearNonnegative) else None
        sr = crd.ShardedRouter(obj, sb.batches, sb.n, device=local_rank, already_sharded=True)
        ctx = getattr(sr._backend, "ctx", None) or getattr(getattr(sr._backend, "local", None), "ctx", None)
        if ctx is not None:
            for kv in sb.args.opt:
                k, val = kv.split("=")
                ctx.set_option(k, int(val))
            if os.environ.get("CFMM_BENCH_SHARE_GPU") == "1":
                ctx.set_option("armed", 0)   # ranks that share a GPU: a waiting launch of one rank holds the CUs another rank's sweep of the SAME evaluation needs
        cr.route_(sr, v=v0, solver="native")   # warm
    except Exception as e:
        err = repr(e)[:300]
    if all_ok(err is None):
        ts = []
        try:
            for _ in range(3):
                t0 = time.perf_counter()
                cr.route_(sr, v=v0, solver="native")
                ts.append(time.perf_counter() - t0)
        except Exception as e:
            err = repr(e)[:300]
        if all_ok(err is None):
            tmax = torch.tensor([min(ts)], dtype=torch.float64, device="cuda")
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            vchk = torch.from_numpy(sr.v.copy()).to("cuda")
            vmax, vmin = vchk.clone(), vchk.clone()
            dist.all_reduce(vmax, op=dist.ReduceOp.MAX)
            dist.all_reduce(vmin, op=dist.ReduceOp.MIN)
            psi, v_star = cr.netflows(sr).copy(), sr.v.copy()
            in_lib = isinstance(sr._backend, cr.DeviceBackend)
            extra = {}
            try:     # every rank polishes (bit-identical Ψ on all ranks: lockstep), rank 0's CPU leg compares the converged points
                J = cr.dual_jacobian(sr)
                cr.polish_(sr, jacobian=J)
                extra = {"_J": J, "_psi_polished": cr.netflows(sr).copy(), "polish": dict(sr.info["polish"])}
            except Exception as e:
                extra = {"polish_error": repr(e)[:200]}
            out = {"ms": 1e3 * float(tmax.item()), "evaluations": sr.info.get("funcalls"),
                   "pools_total": sb.world * sb.m_rank, "ranks_agree_on_v": bool(torch.equal(vmax, vmin)),
                   "max_netflow": float(np.max(np.abs(psi))),
                   "pre_armed": bool(in_lib and sr._backend.ctx.get_option("armed")),
                   "collective": getattr(sr, "collective", "?"), "collective_retries": getattr(sr, "collective_retries", 0),
                   **{k: val for k, val in extra.items() if not k.startswith("_")}}
            hidden = {k: val for k, val in extra.items() if k.startswith("_")}
    if out is None:
        out = {"error": err or "another rank failed"}
        hidden = {}
    return out, psi, v_star, sr, hidden




def collectives_leg(args, name, scaling, rank, world, local_rank, headline):
    """The SAME step timed under each of the three all-reduces, side by side in one run (the first
    multi-GPU lease is one shot): `peer` (cfmm_set_peers: fold + xGMI gather in one launch), `rccl_library` (north_star's
    collective: ncclAllReduce enqueued by the library behind the fold, cfmm_rccl_init_rank) and `rccl_torch`
    (torch.distributed.all_reduce on the sweep's stream).  Per collective: W warm-up steps, K steps between barrier +
    synchronize, MAX over ranks; the result against a plain all-reduce of the local {Ψ, acc}; the sweep kernel's mean span on
    the fastest and the slowest rank.  `headline` (the ShardBench of the timed region) is reused for its own collective.
    Every decision is a collective vote, so a failure on one rank costs that entry (`ok: false, why_not`), never a hang."""
    out = {}

    def all_ok(flag):
        t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return float(t.item()) == 1.0

    for mode in ("peer", "rccl_library", "rccl_torch"):
        sb, own, err = None, False, None
        try:
            if headline is not None and headline.collective == mode and headline.name == name:
                sb = headline
            else:
                sb, own = ShardBench(args, name, scaling, rank, world, local_rank, True, collective=mode), True
        except Exception as e:
            err = repr(e)[:200]
        if not all_ok(err is None):
            out[mode] = {"ok": False, "why_not": err or "construction failed on another rank"}
            if own and sb is not None:
                sb.close()
            continue
        if sb.collective != mode:
            out[mode] = {"ok": False, "why_not": sb.why_not or "not available"}
            if own:
                sb.close()
            continue
        rec = {"ok": True}
        try:
            for _ in range(args.warmup):
                sb.step()
            elapsed, _ = sb.timed_pass(args.steps)
            elapsed = sb.max_over_ranks(elapsed)
            kt, _ = sb.kernel_pass(args.steps)
            k_ms = kt["sweep_ms"] / max(args.steps, 1)
            ks = torch.tensor([k_ms, -k_ms], dtype=torch.float64, device="cuda")
            dist.all_reduce(ks, op=dist.ReduceOp.MAX)
            rec.update(ms_per_step=1e3 * elapsed / args.steps, value=world * sb.m_rank * args.steps / elapsed,
                       kernel_ms_max=float(ks[0].item()), kernel_ms_min=float(-ks[1].item()),
                       fold_or_gather_kernel_ms=kt["reduce_ms"] / max(args.steps, 1),
                       check_rel_err=sb.collective_check(), sharding=sb.sharding_text(),
                       **({"rccl_ranks": world} if mode != "peer" else {"peer_ranks": world}))
        except Exception as e:
            err = repr(e)[:200]
        if not all_ok(err is None):
            rec = {"ok": False, "why_not": err or "failed on another rank"}
        out[mode] = rec
        if own:
            sb.close()
        if headline is not None:
            torch.cuda.set_stream(headline.stream)
    return out
