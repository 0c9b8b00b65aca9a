"""The GPU-side legs of bench.py beside the timed region: route! wall-clock, the host-pointer boundary, the single-process
multi-device line."""
import json
import time

import numpy as np
import torch

import cfmmrouter_amd as cr

from .workloads import METRIC, WORKLOADS, build_global, objective_for, sweep_prices_for


def route_leg(name, batches, n, device):
    """route! wall-clock on the workload, GPU path only: SciPy driving one C-ABI call per evaluation,
    and the library's own L-BFGS-B (cfmm_route, one call) -- with the reference's stopping rules (default), with
    launch-when-ready evaluations instead of pre-armed ones, and with the noise-floor stop."""
    obj = objective_for(name, n)
    v0 = np.ones(n) if isinstance(obj, cr.LinearNonnegative) else None
    r = cr.Router(obj, batches, n, device=device)
    out = {"_router": r}

    inside = {}

    def best_of(k=3, **kw):
        cr.route_(r, v=v0, **kw)  # warm
        ts, ti = [], []
        for _ in range(k):
            t0 = time.perf_counter()
            cr.route_(r, v=v0, **kw)
            ts.append(time.perf_counter() - t0)
            ti.append(r.info.get("total_seconds", float("nan")))
        inside["ms"] = 1e3 * min(ti)     # the C-ABI call's own clock (cfmm_route_info.total_seconds), without the Python wrapper
        return 1e3 * min(ts)

    out["gpu_ms"] = best_of(solver="scipy")
    out["evaluations"] = r.info.get("funcalls")
    out["_psi"], out["_v"] = cr.netflows(r).copy(), r.v.copy()
    out["gpu_native_solver_ms"] = best_of(k=5, solver="native")
    out["native_inside_call_ms"] = inside["ms"]
    out["native_evaluations"] = r.info.get("funcalls")
    out["native_sweep_ms"] = 1e3 * r.info["sweep_seconds"]          # where the one-call route! spends its time:
    out["native_host_solver_ms"] = 1e3 * (r.info["total_seconds"] - r.info["sweep_seconds"])   # device sweeps vs host L-BFGS-B
    out["_psi_native"], out["_v_native"] = cr.netflows(r).copy(), r.v.copy()
    ctx = r._backend.ctx
    if ctx.get_option("armed"):
        ctx.set_option("armed", 0)
        out["native_unarmed_ms"] = best_of(solver="native")
        ctx.set_option("armed", 1)
    ctx.set_option("stop_in_noise", 1)
    out["native_stop_in_noise_ms"] = best_of(solver="native")
    out["native_stop_in_noise_evaluations"] = r.info.get("funcalls")
    out["_psi_native_stop_in_noise"] = cr.netflows(r).copy()
    ctx.set_option("stop_in_noise", 0)

    # route-level parity BY CONVERGENCE: from the library's v*, the gradient-only polish (router.py::polish_) with the
    # device's own finite-difference Jacobian; the CPU leg polishes the restatement's result with the same matrix
    cr.route_(r, v=v0, solver="native")
    t0 = time.perf_counter()
    out["_J"] = cr.dual_jacobian(r)
    cr.polish_(r, jacobian=out["_J"])
    out["polish_ms"] = 1e3 * (time.perf_counter() - t0)
    out["polish"] = dict(r.info["polish"])
    out["_psi_polished"], out["_v_polished"] = cr.netflows(r).copy(), r.v.copy()

    def sweep_at(v):
        cr.find_arb_(r, v)
        return cr.netflows(r).copy()

    out["_sweep_at"] = sweep_at
    return out


def finish_route(route_gpu):
    r = (route_gpu or {}).pop("_router", None)
    if r is not None:
        r.close()
    return {k: val for k, val in (route_gpu or {}).items() if not k.startswith("_")}


def single_process_main(args, cpu_baseline_leg):
    """N shards driven by ONE host thread / process through cfmm_ctx_create_multi (what a Julia or C caller
    uses): every step is a host-pointer cfmm_find_arb -- v staged on every device, N sweeps launched by N
    worker threads, the shards' {Ψ, acc} summed on the host.  PCIe-inclusive by construction."""
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback for the product path")
    devs = [int(d) for d in args.devices.split(",")] if args.devices else list(range(args.gpus))
    if len(devs) != args.gpus:
        raise SystemExit("--devices must list --gpus ordinals")
    desc, n, _ = WORKLOADS[args.workload]
    batches = build_global(args.workload, args.gpus, args.scaling)     # the library splits it into contiguous blocks
    m_total = sum(len(b) for b in batches)
    v = sweep_prices_for(args.workload, n)
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
    psi = np.concatenate([be.ctx.netflows(), [be.ctx.dual_value()]])
    obj = objective_for(args.workload, n)
    v0 = np.ones(n) if isinstance(obj, cr.LinearNonnegative) else None
    r = cr.Router(obj, batches, n, _backend=be)
    route, psi_route, v_route = {}, None, None
    for armed in (1, 0):
        be.ctx.set_option("armed", armed)
        cr.route_(r, v=v0, solver="native")
        ts = []
        for _ in range(3):
            t1 = time.perf_counter()
            cr.route_(r, v=v0, solver="native")
            ts.append(time.perf_counter() - t1)
        route["native_ms" if armed else "native_unarmed_ms"] = 1e3 * min(ts)
        if armed:
            route.update(evaluations=r.info["funcalls"], sweep_ms=1e3 * r.info["sweep_seconds"],
                         pre_armed=len(set(devs)) == len(devs))
            psi_route, v_route = cr.netflows(r).copy(), r.v.copy()
    be.ctx.set_option("armed", 1)
    line = {"metric": METRIC, "value": m_total * args.steps / elapsed, "unit": "pools/s", "n_gpus": args.gpus,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "value_is": "host-pointer cfmm_find_arb calls per second x pools (PCIe-inclusive: v in, Ψ out every step)",
            "config": {"workload": f"{args.workload}: {desc}", "pools_total": m_total, "n_tokens": n, "devices": devs,
                       "sharding": f"single process, cfmm_ctx_create_multi over {args.gpus} shards, host-side rank-ordered sum"},
            "route": route}
    if not args.no_cpu:
        def sweep_at(vv):
            return be.ctx.eval(vv)[0]
        line["cpu_baseline"], line["parity"] = cpu_baseline_leg(
            args.workload, batches, n, v, psi, {"_psi_native": psi_route, "_v_native": v_route, "_sweep_at": sweep_at})
    print(json.dumps(line))
    r.close()




def host_boundary_leg(sb):
    """The synchronous host-pointer boundary (what a ccall from Julia pays per evaluation): pageable v in, Ψ / acc out over
    PCIe -- never the headline value."""
    be, v, host = sb.be, sb.v, {}
    be.ctx.reset_stream()

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
                     ("eval_stream_wait", eval_stream_wait), ("find_arb", lambda: be.find_arb(v))):
        for _ in range(5):
            fn()
        t0 = time.perf_counter()
        for _ in range(50):
            fn()
        host[name + "_us"] = 1e6 * (time.perf_counter() - t0) / 50
    # r.Δs / r.Λs on the host (src/router.jl:7-8): into arrays the caller owns (what the Julia binding fills) and into
    # freshly allocated ones (numpy.empty: the copy then also pays one page fault per 4 KiB of destination)
    be.find_arb(v)
    own = be.trades()                            # (the first call allocates the pinned staging)
    ts, tf = [], []
    for _ in range(3):
        be.find_arb(v)
        t0 = time.perf_counter()
        be.trades(out=own)
        ts.append(time.perf_counter() - t0)
        be.find_arb(v)
        t0 = time.perf_counter()
        be.trades()
        tf.append(time.perf_counter() - t0)
    host["get_trades_ms"] = 1e3 * min(ts)
    host["get_trades_fresh_arrays_ms"] = 1e3 * min(tf)
    host["pools_per_s_host_call_find_arb"] = sb.m_rank / (host["find_arb_us"] * 1e-6)
    be.ctx.set_stream(sb.stream.cuda_stream)
    return host


def expanded_leg(sb, args, local_rank):
    """What the API-faithful materialisation costs.  The timed step writes ONE 16-byte record per pool
    (lossless; `compact_trades`); the reference's r.Δs / r.Λs are 32 bytes per pool (src/router.jl:7-8,40).  Two ways to have
    those rows device-resident, both timed here on the same market, outside the timed region:
      expanded    the timed step followed by cfmm_trades_dev (the `expand_trades` kernel: 16 B read + 32 B written per pool);
      plain_rows  the sweep itself writing {Δ₁, Δ₂} / {Λ₁, Λ₂} (option compact_trades = 0): the sweep of rounds 1-2.
    Each as the cache-warm step and (plain_rows) the HBM-resident kernel, with `frac` in the reference's bytes -- which these
    variants really move."""
    import argparse
    from .shard import ShardBench
    from .workloads import HBM_PEAK_GBS, alg_bytes
    out = {}
    K = max(args.steps, 100)
    be, st = sb.be, sb.stream
    m = sb.m_rank

    def timed(fn):
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K):
            fn()
        while not st.query():
            pass
        torch.cuda.synchronize()
        return 1e3 * (time.perf_counter() - t0) / K

    def sweep_only():
        be.ctx.sweep_dev(sb.v_t.data_ptr(), sb.out_t.data_ptr(), True)

    def sweep_and_expand():
        be.ctx.sweep_dev(sb.v_t.data_ptr(), sb.out_t.data_ptr(), True)
        be.ctx.trades_dev()

    t_plain, t_exp = timed(sweep_only), timed(sweep_and_expand)
    ab = alg_bytes(sb.batches, True, sb.v)
    out["expanded"] = {"ms_per_step": t_exp, "ms_per_step_compact_only": t_plain, "expand_ms": t_exp - t_plain,
                       "expand_bytes": 48 * m, "value": m / (t_exp * 1e-3),
                       "step_frac": ab / (t_exp * 1e-3) / 1e9 / HBM_PEAK_GBS,
                       "is": "cache-warm step = sweep + fold + expand_trades (cfmm_trades_dev): the reference-layout rows device-resident "
                             "after every step; step_frac = reference bytes / that step / 8 TB/s"}
    sub = argparse.Namespace(**vars(args))
    sub.opt = list(args.opt) + ["compact_trades=0"]
    sub.fused = False
    s2 = ShardBench(sub, sb.name, "weak", 0, 1, local_rank, False)
    try:
        for _ in range(10):
            s2.step()
        e2, _ = s2.timed_pass(K)
        kt, _ = s2.kernel_pass(K)
        cold = s2.cold_pass(max(args.steps, 60))
        out["plain_rows"] = {"ms_per_step": 1e3 * e2 / K, "value": m * K / e2, "kernel_ms_warm": kt["sweep_ms"] / K,
                             "kernel_ms_hbm_resident": cold["kernel_ms"], "ms_per_step_hbm_resident": cold["ms_per_step"],
                             "frac": ab / (cold["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "moved_bytes_per_launch": ab - 8 * m,
                             "is": "option compact_trades = 0: the sweep writes the reference's 32-byte Δ/Λ rows itself; frac = reference "
                                   "bytes (what this variant moves, minus the 8 B per pool the packed fee/token record saves) / "
                                   "HBM-resident kernel time / 8 TB/s"}
    finally:
        s2.close()
        torch.cuda.set_stream(sb.stream)
    return out


def roofline_record(sb, args, ms_per_step, sweep_ms, reduce_ms, elapsed2, cold, traffic, traffic_src, traffic_detail):
    """Roofline of the dominant kernel (the sweep launch).  Headline = pool state resident in HBM (the cold pass, or the whole
    timed region with --cold-only): every working set here fits the 256 MB Infinity Cache, so the warm figure (same market
    every step, as inside route!) is a cache number and is reported beside it, never as `frac`.  `frac` / `achieved` price
    the REFERENCE's per-pool bytes (pool state + 32 B of Δ/Λ rows) -- an accounting unit, not bandwidth;
    `bus_frac` prices the bytes the PMC counters saw the launch move (the fraction of the bus), `layout` the bytes this
    layout has to move by construction; `step_frac` the WHOLE step (sweep + fold + boundaries) in the reference's bytes."""
    from .shard import L3_BYTES
    from .workloads import HBM_PEAK_GBS, alg_bytes
    be, n, materialize = sb.be, sb.n, sb.materialize
    bytes_per_launch = alg_bytes(sb.batches, materialize, sb.v)
    achieved = bytes_per_launch / (sweep_ms * 1e-3) / 1e9 if sweep_ms > 0 else 0.0
    warm = {"achieved": achieved, "frac": achieved / HBM_PEAK_GBS, "kernel_ms": sweep_ms}
    if args.cold_only and sb.ring is not None:
        per_copy, copies = sb.market_copies()
        ok = copies * per_copy >= 2 * L3_BYTES and len(sb.ring) == copies
        hbm = dict(warm)
        resid = ("%s: the timed steps rotate over %d market copies = %.0f MB touched (--cold-only)"
                 % ("hbm-resident" if ok else "NOT proven hbm-resident", len(sb.ring), len(sb.ring) * per_copy / 1e6))
        cold = {"copies": len(sb.ring), "touched_per_copy": per_copy, "bytes_touched": len(sb.ring) * per_copy, "hbm_resident": bool(ok),
                "kernel_ms": sweep_ms, "ms_per_step": ms_per_step}
        warm = None
    elif cold is not None:
        hbm = {"achieved": cold["achieved"], "frac": cold["frac"], "kernel_ms": cold["kernel_ms"]}
        resid = ("%s: cold pass over %d market copies (%.0f MB touched per rotation = %.1f x the 256 MiB Infinity Cache; sized by the "
                 "bytes a sweep really moves, not by the reference-layout bytes) after the timed region; the timed region "
                 "itself sweeps one cache-resident market (see `warm`)"
                 % ("hbm-resident" if cold.get("hbm_resident") else "NOT proven hbm-resident (ring smaller than 2 x the cache)",
                    cold["copies"], cold["bytes_touched"] / 1e6, cold["bytes_touched"] / L3_BYTES))
    else:
        hbm, resid = dict(warm), "cache-warm only (no cold pass in this run: --no-cold)"
    compact = bool(materialize and be.ctx.get_option("compact_trades"))
    packed = bool(be.ctx.get_option("pack")) and n <= 8192
    m_all = sum(len(b) for b in sb.batches)
    moved = bytes_per_launch - (16 * m_all if compact else 0) - (8 * m_all if packed else 0)
    layout = {"compact_trades": compact, "packed_records": packed, "bytes_per_launch": moved,
              "achieved": moved / (hbm["kernel_ms"] * 1e-3) / 1e9 if hbm["kernel_ms"] > 0 else 0.0}
    layout["frac"] = layout["achieved"] / HBM_PEAK_GBS
    bus = None
    if traffic and hbm["kernel_ms"] > 0:
        bus = traffic / (hbm["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS
    return {"bound": "hbm", "achieved": hbm["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm["frac"],
            "frac_bus": bus,
            "frac_is": "frac = algorithmic (reference-layout) bytes of SURVEY 8d per launch / HBM-resident kernel time / 8 TB/s: the "
                       "contract's accounting unit -- the timed sweep writes a 16-byte record per pool where the reference's rows are 32 "
                       "(`expanded` prices what producing those rows costs).  frac_bus (= bus_frac) = the bytes the PMC counters saw "
                       "the launch move / the same kernel time / 8 TB/s: the fraction of the bus, the figure to judge the kernel by",
            "traffic": traffic, "bus_frac": bus, "traffic_source": traffic_src, "traffic_detail": traffic_detail,
            "kernel": "cfmm::sweep_multi / cfmm::sweep_kernel (the sweep launch of one step)",
            "alg_bytes_per_launch": bytes_per_launch, "kernel_ms": hbm["kernel_ms"], "residency": resid,
            "layout": layout, "warm": warm, "cold": cold,
            "step_frac": bytes_per_launch / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "reduce_kernel_ms": reduce_ms, "ms_per_step_with_kernel_events": 1e3 * elapsed2 / args.steps,
            "how": "kernel_ms = mean span of the sweep launches (slowest rank) from hipEvent pairs written by the command processor at "
                   "each kernel's start and stop (hipExtLaunchKernel) on the launch stream; the default N = 1 run then REPLACES it by "
                   "the rocprofv3 --kernel-trace --stats average of a --cold-only child of the same command (`kernel_ms_source`; the "
                   "hipEvent figure stays as `kernel_ms_hip_events`) -- the clock of profiles/r06_*_cold_kernel_stats.csv"}
