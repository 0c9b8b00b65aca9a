#!/usr/bin/env python
"""bench.py -- find_arb! sweep throughput on MI355X (BASELINE.json metric), one JSON line.

A "step" is ONE materialising find_arb! sweep over the workload's pools at a fixed price vector:
every pool's closed-form arbitrage is solved, Δ/Λ are written to HBM, and Ψ (netflows) plus the
dual scalar are reduced -- the work of `find_arb!(r, v)` + the scatter loops of `fn`/`g!`
(src/router.jl:38-42, :79-83, :98-100).  Inputs (pools, v) are resident in HBM before the timed
region; at N>1 every rank sweeps its own shard of the same size (weak scaling) and the ranks
all-reduce the n_tokens+1 doubles {Ψ, acc} over RCCL once per step.

    python bench.py [--gpus N --steps K --warmup W --workload config3]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Extra keys on the line: roofline (dominant kernel, hipEvents attached to every sweep launch), route
(route! wall-clock on the same market through the GPU path), cpu_baseline (the C restatement of
the reference path on the host cores: sweep throughput on a bounded sample and route! wall-clock)
and parity (the GPU legs' netflows against that restatement).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# kernel arguments in device memory: measured 22.2 vs 24.9 us per step (config3) and 7.0 vs 9.1 (config2)
# against HIP_FORCE_DEV_KERNARG=0; it is the ROCm 7 default on this box, pinned here in case it is not
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import numpy as np
import torch
import torch.distributed as dist

import cfmmrouter_amd as cr
from cfmmrouter_amd import synth
from cfmmrouter_amd._lib import KIND_GEOMEAN, KIND_PRODUCT, KIND_UNIV3

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

# algorithmic bytes per pool-evaluation, materialising sweep (SURVEY.md §8d / DESIGN.md):
#   read pool state + write Δ(16 B) + Λ(16 B)
ALG_BYTES = {KIND_PRODUCT: 32 + 32, KIND_GEOMEAN: 48 + 32}
ALG_BYTES_FUSED = {KIND_PRODUCT: 32, KIND_GEOMEAN: 48}


def alg_bytes(batches, materialize=True):
    tot = 0
    for b in batches:
        if b.kind == KIND_UNIV3:  # 32 B header + 16 B per tick (+ 32 B of trades)
            tot += len(b) * 32 + 16 * b.lower_ticks.size + (32 * len(b) if materialize else 0)
        else:
            tot += len(b) * (ALG_BYTES if materialize else ALG_BYTES_FUSED)[b.kind]
    return tot


WORKLOADS = {
    # name: (description, n_tokens, builder(rank) -> [PoolBatch], objective builder)
    "config2": ("100k ProductTwoCoin pools, 64 tokens, LinearNonnegative arbitrage", 64,
                lambda rank: [synth.product_pools(100_000, 64, seed=1234, first=rank * 100_000)]),
    "config3": ("1M mixed ProductTwoCoin + GeometricMeanTwoCoin pools (500k each), 256 tokens, "
                "LinearNonnegative arbitrage", 256,
                lambda rank: [synth.product_pools(500_000, 256, seed=1234, first=rank * 500_000),
                              synth.geomean_pools(500_000, 256, seed=1234, first=rank * 500_000)]),
    "config4shard": ("500k ProductTwoCoin pools per GPU (4M over 8 GPUs), 512 tokens", 512,
                     lambda rank: [synth.product_pools(500_000, 512, seed=1234, first=rank * 500_000)]),
    "config5": ("1M BoundedProduct (2-tick UniV3) pools quoted around one token price vector (1 % noise), 256 tokens, "
                "BasketLiquidation (interior dual optimum)", 256,
                lambda rank: [synth.bounded_product_pools(1_000_000, 256, seed=1234, first=rank * 1_000_000,
                                                          consistent=True)]),
    "config5corner": ("1M BoundedProduct pools with independent random prices (arbitrage-rich: route! ends at the box "
                      "corner after 2 evaluations), 256 tokens, BasketLiquidation", 256,
                      lambda rank: [synth.bounded_product_pools(1_000_000, 256, seed=1234, first=rank * 1_000_000)]),
    "large_n": ("1M ProductTwoCoin pools, 65536 tokens (global-bin path), LinearNonnegative arbitrage", 65536,
                lambda rank: [synth.product_pools(1_000_000, 65536, seed=1234, first=rank * 1_000_000)]),
    "product1m": ("1M ProductTwoCoin pools, 256 tokens, LinearNonnegative arbitrage", 256,
                  lambda rank: [synth.product_pools(1_000_000, 256, seed=1234, first=rank * 1_000_000)]),
}


def objective_for(name, n):
    if name.startswith("config5"):
        return cr.BasketLiquidation(1, synth.basket(n, seed=1234))
    return cr.LinearNonnegative(synth.linear_prices(n, seed=1234))


def cpu_baseline_leg(name, batches, n, v, psi_dev, route_gpu, budget_s=12.0):
    """THE one place in bench.py that touches oracle/ (test infrastructure): the CPU restatement of
    the reference path is (a) timed on the host cores as the reported baseline -- OpenMP sweep like
    Threads.@threads, then the reference's two SERIAL reductions (src/router.jl:81-83, :98-100) --
    on a bounded sample, (b) timed once on route!, and (c) used as the checker of the numbers the
    GPU legs produced.  It is never the thing measured as `value`."""
    from oracle import cfmm_oracle as orc
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import oracle_objective, oracle_poolset

    ps = oracle_poolset(batches, n)
    threads = orc.lib().oracle_max_threads()
    m = ps.m
    reps, t_tot = 0, 0.0
    ps.sweep(v, threads)  # warm caches / thread pool
    while t_tot < budget_s and reps < 200:
        t0 = time.perf_counter()
        D, L = ps.sweep(v, threads)
        acco = orc.dual_acc(D, L, ps.Ai, v)
        G = np.zeros(n)
        orc.grad_scatter(G, D, L, ps.Ai)
        t_tot += time.perf_counter() - t0
        reps += 1
    base = {"value": m * reps / t_tot, "unit": "pools/s", "cores": int(threads), "kind": "port",
            "sample": f"{reps} full sweeps of the same {m}-pool workload (oracle/cfmm_oracle.c: OpenMP sweep + "
                      f"serial dual/gradient reductions), {t_tot:.1f} s of host time; the Julia reference itself "
                      f"cannot run here (no Julia toolchain)"}
    parity = {"netflow_rel_err_at_fixed_v": float(np.max(np.abs(psi_dev[:n] - G)) / np.max(np.abs(G))),
              "dual_rel_err": float(abs(psi_dev[n] - acco) / max(abs(acco), 1.0))}
    if route_gpu and "error" not in route_gpu:
        obj = objective_for(name, n)
        v0 = np.ones(n) if isinstance(obj, cr.LinearNonnegative) else None
        t0 = time.perf_counter()
        ref = orc.route_oracle(oracle_objective(obj), ps, v0=v0, nthreads=threads)
        base["route_ms"] = 1e3 * (time.perf_counter() - t0)
        base["route_evaluations"] = ref["info"]["funcalls"]
        scale = np.max(np.abs(ref["psi"]))
        parity["route_netflow_rel_err"] = float(np.max(np.abs(route_gpu.pop("_psi") - ref["psi"])) / scale)
        parity["route_native_netflow_rel_err"] = float(np.max(np.abs(route_gpu.pop("_psi_native") - ref["psi"])) / scale)
    return base, parity


def route_leg(name, batches, n):
    """route! wall-clock on the workload, GPU path only: SciPy driving one C-ABI call per evaluation,
    and the library's own L-BFGS-B (cfmm_route, one call)."""
    obj = objective_for(name, n)
    v0 = np.ones(n) if isinstance(obj, cr.LinearNonnegative) else None
    r = cr.Router(obj, batches, n, device=torch.cuda.current_device())
    out = {}
    for key, solver in (("gpu_ms", "scipy"), ("gpu_native_solver_ms", "native")):
        cr.route_(r, v=v0, solver=solver)  # warm
        times = []
        for _ in range(3):
            t0 = time.perf_counter()
            cr.route_(r, v=v0, solver=solver)
            times.append(time.perf_counter() - t0)
        out[key] = 1e3 * min(times)
        out["evaluations" if solver == "scipy" else "native_evaluations"] = r.info.get("funcalls")
        if solver == "native":   # where the one-call route! spends its time: device sweeps vs the host L-BFGS-B
            out["native_sweep_ms"] = 1e3 * r.info["sweep_seconds"]
            out["native_host_solver_ms"] = 1e3 * (r.info["total_seconds"] - r.info["sweep_seconds"])
            # the same call with launch-when-ready evaluations instead of pre-armed ones (option "armed" = 0)
            if isinstance(r._backend, cr.DeviceBackend) and r._backend.ctx.get_option("armed"):
                r._backend.ctx.set_option("armed", 0)
                cr.route_(r, v=v0, solver=solver)
                t_un = []
                for _ in range(3):
                    t0 = time.perf_counter()
                    cr.route_(r, v=v0, solver=solver)
                    t_un.append(time.perf_counter() - t0)
                out["native_unarmed_ms"] = 1e3 * min(t_un)
                r._backend.ctx.set_option("armed", 1)
                cr.route_(r, v=v0, solver=solver)
        out["_psi" if solver == "scipy" else "_psi_native"] = cr.netflows(r).copy()
    r.close()
    return out


def single_process_main(args):
    """N shards driven by ONE host thread / process through cfmm_ctx_create_multi (what a Julia or C caller
    uses): every step is a host-pointer cfmm_find_arb -- v staged on every device, N sweeps launched by N
    worker threads, the shards' {Ψ, acc} summed on the host.  PCIe-inclusive by construction."""
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback for the product path")
    devs = [int(d) for d in args.devices.split(",")] if args.devices else list(range(args.gpus))
    if len(devs) != args.gpus:
        raise SystemExit("--devices must list --gpus ordinals")
    desc, n, build = WORKLOADS[args.workload]
    shards = [build(r) for r in range(args.gpus)]                  # weak scaling: one config-sized shard per device
    batches = [cr.PoolBatch.concat([sh[k] for sh in shards]) for k in range(len(shards[0]))]
    m_total = sum(len(b) for b in batches)
    v = synth.sweep_prices(n, seed=1234)
    be = cr.DeviceBackend(n, batches, device=devs)
    for kv in args.opt:
        k, val = kv.split("=")
        be.ctx.set_option(k, int(val))
    for _ in range(args.warmup):
        be.ctx.find_arb(v)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        be.ctx.find_arb(v)
    elapsed = time.perf_counter() - t0
    psi = be.ctx.netflows()
    obj = objective_for(args.workload, n)
    v0 = np.ones(n) if isinstance(obj, cr.LinearNonnegative) else None
    r = cr.Router(obj, batches, n, _backend=be)
    cr.route_(r, v=v0, solver="native")
    ts = []
    for _ in range(3):
        t1 = time.perf_counter()
        cr.route_(r, v=v0, solver="native")
        ts.append(time.perf_counter() - t1)
    line = {"metric": "find_arb! pools/sec + route! wall-clock, 1M-pool arbitrage, 1/2/4/8 GPU",
            "value": m_total * args.steps / elapsed, "unit": "pools/s", "n_gpus": args.gpus, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "value_is": "host-pointer cfmm_find_arb calls per second x pools (PCIe-inclusive: v in, Ψ out every step)",
            "config": {"workload": f"{args.workload}: {desc}", "pools_total": m_total, "n_tokens": n, "devices": devs,
                       "sharding": f"single process, cfmm_ctx_create_multi over {args.gpus} shards, host-side rank-ordered sum"},
            "route": {"native_ms": 1e3 * min(ts), "evaluations": r.info["funcalls"],
                      "sweep_ms": 1e3 * r.info["sweep_seconds"], "max_netflow": float(np.max(np.abs(psi)))}}
    print(json.dumps(line))
    r.close()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="config3", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline and route legs")
    ap.add_argument("--fused", action="store_true", help="time the fused evaluation (no Δ/Λ write-back)")
    ap.add_argument("--opt", action="append", default=[], help="library option key=value")
    ap.add_argument("--no-cold", action="store_true", help="skip the cache-cold pass")
    ap.add_argument("--rccl", action="store_true", help="force the RCCL all-reduce instead of the one-shot peer gather")
    ap.add_argument("--cold-only", action="store_true",
                    help="the timed region rotates over > 300 MB of market copies (pool state from HBM, not the Infinity "
                         "Cache): used for the rocprofv3 summary of the HBM-resident figure")
    ap.add_argument("--single-process", action="store_true",
                    help="--gpus N from ONE process: the multi-device context of the C ABI (cfmm_ctx_create_multi), a step is "
                         "one host-pointer cfmm_find_arb over all N shards (PCIe-inclusive)")
    ap.add_argument("--devices", default="", help="--single-process: comma-separated HIP ordinals (default 0..N-1; an "
                                                  "ordinal may repeat to put several shards on one GPU)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.single_process:
        return single_process_main(args)
    if args.gpus > 1 and world == 1 and "TORCHELASTIC_RUN_ID" not in os.environ:
        # bare `python bench.py --gpus N`: become the launcher -- one rank per GPU over RCCL, rendezvous on
        # 127.0.0.1 (the container hostname may not resolve).  Same command line, re-executed under torchrun.
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        print(f"[bench] spawning {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
        os.execv(sys.executable, cmd)
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} does not match WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit(f"[rank {rank}/{world}] bench.py needs an MI355X; there is no CPU fallback for the product path")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit(f"[rank {rank}/{world}] local rank {local_rank} has no GPU: {torch.cuda.device_count()} visible "
                         f"(one rank per GPU; RCCL does not share a device between ranks)")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or "TORCHELASTIC_RUN_ID" in os.environ  # under torchrun even N=1 goes through RCCL
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # a healthy peer exchange takes microseconds and the bench's ranks run in lockstep: bound a broken one (the
        # start-up check then falls back to RCCL on all ranks) by seconds, not by the library's 30 s default
        os.environ.setdefault("CFMM_AMD_PEER_TIMEOUT_S", "5")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    desc, n, build = WORKLOADS[args.workload]
    batches = build(rank)
    m_rank = sum(len(b) for b in batches)
    v = synth.sweep_prices(n, seed=1234)
    be = cr.DeviceBackend(n, batches, device=local_rank)
    for kv in args.opt:
        k, val = kv.split("=")
        be.ctx.set_option(k, int(val))
    stream = torch.cuda.Stream()          # the sweep, the RCCL all-reduce and the events share it
    torch.cuda.set_stream(stream)
    be.ctx.set_stream(stream.cuda_stream)
    v_t = torch.from_numpy(v).to("cuda")
    out_t = torch.zeros(n + 1, dtype=torch.float64, device="cuda")
    materialize = not args.fused
    # N > 1 (or N = 1 under torchrun): the launch that folds the partial rows also all-reduces {Ψ, acc}
    # over xGMI peer mappings (cfmm_set_peers: one launch, rank-ordered sum, bit-identical on every
    # rank).  At start-up that path is checked against sweep + RCCL all-reduce on every rank; if it is
    # unavailable or disagrees anywhere, ALL ranks use the RCCL all-reduce instead.
    peer, fused_peer, peer_ptrs, n_fused = None, False, None, 0
    if use_dist and not args.rccl and os.environ.get("CFMM_AMD_NO_PEER", "0") != "1":
        from cfmmrouter_amd.dist import open_peer_buffers
        peer = open_peer_buffers(be.ctx, None, torch.device("cuda", local_rank))  # None (on every rank) -> RCCL fallback
        if peer is not None:
            peer_ptrs = list(peer.ptrs)
            good = True
            for _ in range(3):
                be.ctx.set_peers(peer_ptrs, world, rank, n_fused)
                be.ctx.sweep_dev(v_t.data_ptr(), out_t.data_ptr(), materialize)
                n_fused += 1
                got = out_t.clone()
                be.ctx.set_peers([], 0, 0, 0)
                be.ctx.sweep_dev(v_t.data_ptr(), out_t.data_ptr(), materialize)
                ref = out_t.clone()
                dist.all_reduce(ref)
                torch.cuda.synchronize()
                good = good and bool(torch.isfinite(got).all()) and \
                    float((got - ref).abs().max()) <= 1e-12 * float(ref.abs().max())
            flag = torch.tensor([1.0 if good else 0.0], dtype=torch.float64, device="cuda")
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            fused_peer = float(flag.item()) == 1.0
            if fused_peer:
                be.ctx.set_peers(peer_ptrs, world, rank, n_fused)

    def step():
        be.ctx.sweep_dev(v_t.data_ptr(), out_t.data_ptr(), materialize)   # sharded context: already the global {Ψ, acc}
        if use_dist and not fused_peer:
            dist.all_reduce(out_t)  # Ψ and the dual scalar: one small RCCL collective per evaluation

    n_steps_run = [0]
    _plain_step = step

    def step():   # noqa: F811 -- counts the fused sweeps so that the sequence can be resumed after a local sweep
        _plain_step()
        n_steps_run[0] += 1

    ring_ctx = None
    if args.cold_only and not use_dist:
        # rotate the timed steps over enough copies of the market to exceed the 256 MB Infinity Cache
        per_copy = alg_bytes(batches, True) + 16 * sum(len(b) for b in batches if b.kind == KIND_GEOMEAN)
        copies = int(np.ceil(320e6 / per_copy)) + 1
        ring_ctx = [be] + [cr.DeviceBackend(n, batches, device=local_rank) for _ in range(copies - 1)]
        for b_ in ring_ctx:
            b_.ctx.set_stream(stream.cuda_stream)
            for kv in args.opt:
                k, val = kv.split("=")
                b_.ctx.set_option(k, int(val))
        ring_pos = [0]

        def step():   # noqa: F811
            b_ = ring_ctx[ring_pos[0] % len(ring_ctx)]
            ring_pos[0] += 1
            b_.ctx.sweep_dev(v_t.data_ptr(), out_t.data_ptr(), materialize)

    for _ in range(args.warmup):
        step()

    def timed_pass(device_events=False):
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        ev0 = ev1 = None
        if device_events:
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0 = time.perf_counter()
        if device_events:
            ev0.record(stream)
        for _ in range(args.steps):
            step()
        if device_events:
            ev1.record(stream)
        while not stream.query():   # busy-wait for the last step (a blocking wait adds its wake-up latency to the K
            pass                    # steps: ~1 us per step at the driver's K = 20), then the synchronize of the contract
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0   # this rank's K steps are complete (with the collective inside every step no
        if use_dist:                    # rank finishes step k before all ranks contributed to it); the closing
            dist.barrier()              # barrier follows the clock read, and the MAX over ranks is reported
        return dt, (ev0.elapsed_time(ev1) if device_events else None)

    # pass 1 -- THE timed region: K steps between barrier+synchronize, nothing else on the stream or the host
    elapsed, _ = timed_pass()
    # pass 2 -- the same K steps again with a hipEvent pair attached to every kernel launch (start / stop written
    # by the command processor, hipExtLaunchKernel) for the roofline, and one event pair around the K steps
    # (device-side step time); kept out of pass 1 so that the timed region carries nothing but the work
    timed_ctxs = ring_ctx if ring_ctx else [be]
    for b_ in timed_ctxs:
        b_.ctx.set_option("time_kernels", 1)
        b_.ctx.kernel_times()  # reset
    elapsed2, dev_ms = timed_pass(device_events=True)
    kt = {"sweep_ms": 0.0, "reduce_ms": 0.0}
    for b_ in timed_ctxs:
        kt_b = b_.ctx.kernel_times()
        kt["sweep_ms"] += kt_b["sweep_ms"]
        kt["reduce_ms"] += kt_b["reduce_ms"]
        b_.ctx.set_option("time_kernels", 0)
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # cold pass: every working set here (<= 100 MB) fits the 256 MB Infinity Cache, so
    # the passes above are "warm" (what a running route! sees).  Rotating over enough distinct copies
    # of the market to exceed 300 MB makes every sweep read its pool state from HBM.
    cold = None
    if world == 1 and not use_dist and not args.no_cold and not args.cold_only:
        per_copy = alg_bytes(batches, True) + 16 * sum(len(b) for b in batches if b.kind == KIND_GEOMEAN)
        copies = int(np.ceil(320e6 / per_copy)) + 1
        extra = [cr.DeviceBackend(n, batches, device=local_rank) for _ in range(copies - 1)]
        ring = [be] + extra
        outs = [torch.zeros(n + 1, dtype=torch.float64, device="cuda") for _ in ring]
        for b_ in ring:
            b_.ctx.set_stream(stream.cuda_stream)
        for k in range(2 * copies):
            ring[k % copies].ctx.sweep_dev(v_t.data_ptr(), outs[k % copies].data_ptr(), materialize)
        torch.cuda.synchronize()
        for b_ in ring:
            b_.ctx.set_option("time_kernels", 1)
            b_.ctx.kernel_times()
        cold_steps = max(args.steps, 60)     # a stable average: at the driver's K = 20 the figure moves by +-0.02
        t0 = time.perf_counter()
        for k in range(cold_steps):
            ring[k % copies].ctx.sweep_dev(v_t.data_ptr(), outs[k % copies].data_ptr(), materialize)
        torch.cuda.synchronize()
        cold_elapsed = time.perf_counter() - t0
        sw = sum(b_.ctx.kernel_times()["sweep_ms"] for b_ in ring) / cold_steps
        for b_ in ring:
            b_.ctx.set_option("time_kernels", 0)
        cold = {"copies": copies, "bytes_rotated": copies * per_copy, "kernel_ms": sw,
                "achieved": alg_bytes(batches, materialize) / (sw * 1e-3) / 1e9 if sw > 0 else 0.0,
                "ms_per_step_with_kernel_events": 1e3 * cold_elapsed / cold_steps, "sweeps": cold_steps}
        cold["frac"] = cold["achieved"] / HBM_PEAK_GBS
        for b_ in extra:
            b_.close()
        be.ctx.set_stream(stream.cuda_stream)

    # sharded runs: the timed path's global {Ψ, acc} against a plain RCCL all-reduce of the local ones
    collective_check = None
    if use_dist:
        step()
        got = out_t.clone()
        if fused_peer:
            be.ctx.set_peers([], 0, 0, 0)           # a LOCAL sweep for the reference
        be.ctx.sweep_dev(v_t.data_ptr(), out_t.data_ptr(), materialize)
        ref = out_t.clone()
        dist.all_reduce(ref)
        torch.cuda.synchronize()
        if fused_peer:
            be.ctx.set_peers(peer_ptrs, world, rank, n_fused + n_steps_run[0])
        collective_check = float((got - ref).abs().max() / ref.abs().max())

    # sharded route!: every rank drives the same L-BFGS-B on the all-reduced {Ψ, acc} of its own shard
    route_sharded = None
    if use_dist:
        def all_ok(flag):   # collective vote, so that no rank walks into a collective alone
            t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return float(t.item()) == 1.0

        sr, err = None, None
        try:
            from cfmmrouter_amd import dist as crd
            obj = objective_for(args.workload, n)
            v0 = np.ones(n) if isinstance(obj, cr.LinearNonnegative) else None
            sr = crd.ShardedRouter(obj, batches, n, device=local_rank, already_sharded=True)
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
                route_sharded = {"ms": 1e3 * float(tmax.item()), "evaluations": sr.info.get("funcalls"),
                                 "pools_total": world * m_rank, "ranks_agree_on_v": bool(torch.equal(vmax, vmin)),
                                 "max_netflow": float(np.max(np.abs(cr.netflows(sr)))),
                                 "collective": ("peer all-reduce inside the library (cfmm_set_peers), route! = one call "
                                                "per rank" if isinstance(sr._backend, cr.DeviceBackend)
                                                else "rccl via torch.distributed")}
        if route_sharded is None:
            route_sharded = {"error": err or "another rank failed"}
        if sr is not None:
            sr.close()

    # the synchronous host-pointer boundary (what a ccall from Julia pays per evaluation):
    # pageable v in, Ψ/acc out over PCIe, one stream sync -- never the headline value
    be.ctx.reset_stream()
    host = {}
    if world == 1:
        def eval_copy():
            be.ctx.set_option("zero_copy", 0)
            r = be.eval(v)
            be.ctx.set_option("zero_copy", 1)
            return r

        def eval_stream_wait():
            be.ctx.set_option("host_flag", 0)
            r = be.eval(v)
            be.ctx.set_option("host_flag", 1)
            return r

        for name, fn in (("eval", lambda: be.eval(v)), ("eval_with_copy_commands", eval_copy),
                         ("eval_stream_wait", eval_stream_wait),
                         ("find_arb", lambda: be.find_arb(v))):
            for _ in range(5):
                fn()
            t0 = time.perf_counter()
            for _ in range(50):
                fn()
            host[name + "_us"] = 1e6 * (time.perf_counter() - t0) / 50
        t0 = time.perf_counter()
        be.trades()
        host["get_trades_ms"] = 1e3 * (time.perf_counter() - t0)
    be.ctx.set_stream(stream.cuda_stream)

    # sanity: the timed path produced the oracle's netflows (spot check on rank 0, small sample)
    psi_dev = out_t.cpu().numpy()

    ms_per_step = 1e3 * elapsed / args.steps
    value = world * m_rank * args.steps / elapsed
    bytes_per_launch = alg_bytes(batches, materialize)
    sweep_ms = kt["sweep_ms"] / max(args.steps, 1)  # all sweep launches of one step
    achieved = bytes_per_launch / (sweep_ms * 1e-3) / 1e9 if sweep_ms > 0 else 0.0
    traffic = None
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tf):
        try:
            traffic = json.load(open(tf)).get(args.workload + ("_fused" if args.fused else ""))
        except Exception:
            traffic = None

    # Roofline of the dominant kernel (the sweep launch).  Headline = pool state resident in HBM (the cold
    # pass, or the whole timed region with --cold-only): every working set here fits the 256 MB Infinity
    # Cache, so the warm figure (same market every step, as inside route!) is a cache number and is reported
    # beside it, never as `frac`.  step_frac prices the WHOLE step (sweep + fold + boundaries) the same way.
    warm = {"achieved": achieved, "frac": achieved / HBM_PEAK_GBS, "kernel_ms": sweep_ms}
    if args.cold_only:
        hbm, resid = dict(warm), "hbm-resident: the timed steps rotate over > 300 MB of market copies (--cold-only)"
        warm = None
    elif cold is not None:
        hbm = {"achieved": cold["achieved"], "frac": cold["frac"], "kernel_ms": cold["kernel_ms"]}
        resid = ("hbm-resident: cold pass over %d market copies (%.0f MB rotated) after the timed region; the timed "
                 "region itself sweeps one cache-resident market (see `warm`)" % (cold["copies"], cold["bytes_rotated"] / 1e6))
    else:
        hbm, resid = dict(warm), "cache-warm only (no cold pass in this run: N > 1 or --no-cold)"
    # `achieved` prices the REFERENCE's per-pool bytes (pool state + 32 B of Δ/Λ rows).  With the compact
    # trade records the launch moves 16 B less per pool than that, so the rate over the bytes this layout really has
    # to move is reported beside it (it is the smaller number and the one to read as "fraction of the bus").
    compact = bool(materialize and be.ctx.get_option("compact_trades"))
    moved = bytes_per_launch - (16 * sum(len(b) for b in batches) if compact else 0)
    layout = {"compact_trades": compact, "bytes_per_launch": moved,
              "achieved": moved / (hbm["kernel_ms"] * 1e-3) / 1e9 if hbm["kernel_ms"] > 0 else 0.0}
    layout["frac"] = layout["achieved"] / HBM_PEAK_GBS
    roofline = {"bound": "hbm", "achieved": hbm["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm["frac"],
                "traffic": traffic, "kernel": "cfmm::sweep_multi / cfmm::sweep_kernel (the sweep launch of one step)",
                "alg_bytes_per_launch": bytes_per_launch, "kernel_ms": hbm["kernel_ms"], "residency": resid,
                "layout": layout,
                "warm": warm, "cold": cold,
                "step_frac": bytes_per_launch / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "reduce_kernel_ms": kt["reduce_ms"] / max(args.steps, 1),
                "ms_per_step_with_kernel_events": 1e3 * elapsed2 / args.steps,
                "how": "kernel_ms = mean duration of the sweep launches, from hipEvent pairs written by the command "
                       "processor at each kernel's start and stop (hipExtLaunchKernel) on the launch stream; compare "
                       "profiles/r02_*_kernel_stats.csv (warm: --no-cold runs; hbm-resident: --cold-only runs)"}

    line = {
        "metric": "find_arb! pools/sec + route! wall-clock, 1M-pool arbitrage, 1/2/4/8 GPU",
        "value": value, "unit": "pools/s",
        "value_is": "find_arb! pool-evaluations per second (materialising sweep + Ψ/dual reduction); "
                    "route! wall-clock is reported under `route`", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {desc}", "pools_per_gpu": m_rank, "n_tokens": n,
                   "variant": "materialising" if materialize else "fused", "segments": be.ctx.segments(),
                   "sharding": ((f"pools x{world}, fold + one-shot xGMI peer all-reduce of n_tokens+1 f64 in one launch per step"
                                 + " (buffers: library IPC export)"
                                 if fused_peer else f"pools x{world}, RCCL all-reduce of n_tokens+1 f64 per step")
                                if use_dist else "single GPU, no collective")},
        "roofline": roofline,
        "library_options": {k: be.ctx.get_option(k) for k in ("pack", "compact_trades", "alternate", "fast_math", "armed",
                                                               "stop_in_noise", "host_flag", "zero_copy")},
    }
    if collective_check is not None:
        line["collective_check_rel_err"] = collective_check
    if route_sharded is not None:
        line["route_sharded"] = route_sharded
    if host:
        host["pools_per_s_host_call_find_arb"] = m_rank / (host["find_arb_us"] * 1e-6)
        line["host_boundary"] = host
    if rank == 0 and world == 1 and not use_dist:
        try:
            route_gpu = route_leg(args.workload, batches, n)
        except Exception as e:  # the route leg is informational; never lose the bench line over it
            route_gpu = {"error": repr(e)[:300]}
        if not args.no_cpu:
            line["cpu_baseline"], line["parity"] = cpu_baseline_leg(args.workload, batches, n, v, psi_dev, route_gpu)
        line["route"] = {k: val for k, val in route_gpu.items() if not k.startswith("_")}
    if ring_ctx:
        for b_ in ring_ctx[1:]:
            b_.close()
    if peer is not None and hasattr(peer, "close"):
        peer.close()
    be.close()
    if rank == 0:
        print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
