#!/usr/bin/env python
"""bench.py -- find_arb! sweep throughput on MI355X (BASELINE.json metric), one JSON line.

A "step" is ONE materialising find_arb! sweep over the workload's pools at a fixed price vector:
every pool's closed-form arbitrage is solved, Δ/Λ are written to HBM, and Ψ (netflows) plus the
dual scalar are reduced -- the work of `find_arb!(r, v)` + the scatter loops of `fn`/`g!`
(src/router.jl:38-42, :79-83, :98-100).  Inputs (pools, v) are resident in HBM before the timed
region.  At N > 1 every rank sweeps its own shard and the launch that folds a rank's partial rows also
all-reduces the n_tokens+1 doubles {Ψ, acc} over xGMI (fall-back: one small RCCL all-reduce per step).

    python bench.py [--gpus N --steps K --warmup W --workload config3 --scaling weak|strong]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...
    python bench.py --workload scaling          # the reference's own benchmark grid (benchmark/scaling.jl), route! per point

    --scaling weak   (default)  every GPU sweeps one config-sized shard (config3: 1M pools per GPU)
    --scaling strong            the workload's pool count is divided among the GPUs (config4: 4M / N per GPU)

Where what lives (benchlib/): workloads.py (markets, byte accounting), shard.py (ShardBench: one rank's backend, stream,
peer buffers, the step and the timed passes; the sharded route! leg), legs.py (route! wall-clock, host-pointer boundary,
roofline record, the single-process multi-device line), traffic.py (live PMC traffic), grid.py (the reference's grid).
This file keeps main() -- the timed region is `pass 1` below -- and cpu_baseline_leg, the ONE place that touches oracle/.

Extra keys on the line: roofline (dominant kernel, hipEvents attached to every sweep launch; `frac` is the HBM-resident
figure in the reference's bytes, `bus_frac` the fraction of the bus from the PMC bytes), route (route! wall-clock on the same
market through the GPU path), cpu_baseline (the C restatement of the reference path on the host cores: sweep throughput on
a bounded sample and route! wall-clock) and parity (the GPU legs' netflows against that restatement, at the reference's
tolerances and BY CONVERGENCE) -- at every N: at N > 1 rank 0 regenerates the GLOBAL market on the host for the oracle.  At
N > 1 with the default workload the line also carries `strong_scaling` (config 4: 4M pools / N per GPU).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
HOST_CPUS = len(os.sched_getaffinity(0))   # read NOW: once the OpenMP runtime binds the main thread (OMP_PROC_BIND below) its affinity is one core
# kernel arguments in device memory: measured 22.2 vs 24.9 us per step (config3) and 7.0 vs 9.1 (config2)
# against HIP_FORCE_DEV_KERNARG=0; it is the ROCm 7 default on this box, pinned here in case it is not
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
# the CPU restatement's OpenMP threads (cpu_baseline leg only): one per core, spread over the sockets, pinned -- round to
# round the unpinned figure moved by 1.4x (1.35e7 .. 2.39e7 pools/s over rounds 1-4)
def _single_process_run():
    """True when this process is the only rank on the box (no torchrun world, no --gpus N > 1 about to re-exec under it)."""
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        return False
    for k, a in enumerate(sys.argv):
        if a == "--gpus" and k + 1 < len(sys.argv) and sys.argv[k + 1] != "1":
            return False
        if a.startswith("--gpus=") and a != "--gpus=1":
            return False
    return True


if _single_process_run():
    # (NOT at N > 1: the OpenMP runtime binds every process's MAIN thread to the first place when it starts, so eight ranks would
    #  drive their GPUs from one core; and the environment would be inherited by the ranks this script re-executes itself as)
    os.environ.setdefault("OMP_PLACES", "cores")
    os.environ.setdefault("OMP_PROC_BIND", "spread")
# ... and sleeping between parallel regions: with libgomp's default (spinning) wait 30 % of the 128-thread evaluations on the
# GPU box took ~90 ms instead of 3 ms (the spinning threads burn the container's CPU quota and the whole cgroup is throttled
# for the rest of the scheduler period) -- the reason the figure moved between 1.35e7 and 3.4e7 pools/s from round to round
# (profiles/r05_cpu_baseline_probe.txt: 128 threads, mean 3.4e7 active vs 2.1e8 passive; median 3.1e8 either way)
os.environ.setdefault("OMP_WAIT_POLICY", "passive")

import numpy as np
import torch
import torch.distributed as dist

import cfmmrouter_amd as cr
from benchlib.legs import expanded_leg, finish_route, host_boundary_leg, roofline_record, route_leg, single_process_main
from benchlib.shard import ShardBench, collectives_leg, sharded_route
from benchlib.traffic import live_kernel_stats, live_traffic
from benchlib.workloads import METRIC, WORKLOADS, build_global, objective_for   # noqa: F401  (scripts import WORKLOADS from here)


def cpu_baseline_leg(name, batches, n, v, psi_dev, route_gpu, budget_s=12.0, route_budget_pools=8_500_000):
    """THE one place in bench.py that touches oracle/ (test infrastructure): the CPU restatement of
    the reference path is (a) timed on the host cores as the reported baseline -- OpenMP sweep like
    Threads.@threads, then the reference's two SERIAL reductions (src/router.jl:81-83, :98-100) --
    on a bounded sample, (b) timed once on route!, and (c) used as the checker of the numbers the
    GPU legs produced.  It is never the thing measured as `value`.  `batches` is the GLOBAL market (all ranks'
    shards), `psi_dev` the {Ψ, acc} the timed path left behind, `route_gpu` the route legs' results."""
    from oracle import cfmm_oracle as orc
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import oracle_objective, oracle_poolset

    ps = oracle_poolset(batches, n)
    m = ps.m
    # thread count: the best of {16, 64, one per physical core (<= 128)} on a short calibration -- more threads shorten the sweep
    # (5.6 -> 1.0 ms from 16 to 128 threads on the GPU box's 2 x EPYC) but not the serial reductions (2 ms), and SMT siblings only
    # add contention to this memory-bound loop (256 threads: 200 ms per sweep)
    ncpu = HOST_CPUS
    Dc, Lc = np.empty((m, 2)), np.empty((m, 2))
    best = None
    for cand in sorted({c for c in (16, 64, min(128, max(1, ncpu // 2))) if c <= ncpu} or {1}):
        ps.sweep_into(v, Dc, Lc, cand)
        t0 = time.perf_counter()
        for _ in range(6):
            ps.sweep_into(v, Dc, Lc, cand)
        dt = (time.perf_counter() - t0) / 6
        if best is None or dt < best[0]:
            best = (dt, cand)
    threads = best[1]
    del Dc, Lc
    # find_arb!(r, v) overwrites r.Δs / r.Λs in place (src/router.jl:40): the outputs are allocated ONCE, like the
    # reference's; the threaded sweep (Threads.@threads, :39) and the two SERIAL reductions of fn / g! (:81-83, :98-100) are
    # timed separately -- at 128 threads the serial part is most of a CPU evaluation (Amdahl), which is what the GPU path's
    # in-kernel reductions remove.  Thread placement: OMP_PLACES=cores / OMP_PROC_BIND=spread (set at the top of this file).
    D, L = np.empty((m, 2)), np.empty((m, 2))
    G = np.zeros(n)
    ps.sweep_into(v, D, L, threads)  # warm caches / thread pool
    reps, t_sweep, t_red, per_rep = 0, 0.0, 0.0, []
    while t_sweep + t_red < budget_s and reps < 400:
        t0 = time.perf_counter()
        ps.sweep_into(v, D, L, threads)
        t1 = time.perf_counter()
        acco = orc.dual_acc(D, L, ps.Ai, v)
        G[:] = 0.0
        orc.grad_scatter(G, D, L, ps.Ai)
        t2 = time.perf_counter()
        t_sweep += t1 - t0
        t_red += t2 - t1
        per_rep.append(t2 - t0)
        reps += 1
    t_tot = t_sweep + t_red
    one = []
    for _ in range(2):                   # the same sweep on ONE thread (bounded: two sweeps)
        t0 = time.perf_counter()
        ps.sweep_into(v, D, L, 1)
        one.append(time.perf_counter() - t0)
    ps.sweep_into(v, D, L, threads)
    per_rep = np.sort(np.array(per_rep))
    base = {"value": m * reps / t_tot, "unit": "pools/s", "cores": int(threads), "threads": int(threads), "host_cpus": int(ncpu), "kind": "port",
            "sweep_ms": 1e3 * t_sweep / reps, "reductions_ms": 1e3 * t_red / reps,
            "value_sweep_only": m * reps / t_sweep, "value_1thread": m / (min(one) + t_red / reps),
            "sweep_1thread_ms": 1e3 * min(one),
            "evaluation_ms_quartiles": [float(1e3 * per_rep[int(q * (len(per_rep) - 1))]) for q in (0.25, 0.5, 0.75)],
            "is": "value = pools / (threaded sweep + the reference's two serial reductions); sweep_ms is the Threads.@threads part "
                  "(src/router.jl:39), reductions_ms the single-threaded loops of fn / g! (:81-83, :98-100); value_1thread = the "
                  "same evaluation on one thread; outputs pre-allocated, OMP_PLACES=cores OMP_PROC_BIND=spread OMP_WAIT_POLICY=passive; "
                  "thread count = the best of {16, 64, one per physical core} on a short calibration",
            "sample": f"{reps} full evaluations of the same {m}-pool workload (oracle/cfmm_oracle.c: OpenMP sweep + "
                      f"serial dual/gradient reductions), {t_tot:.1f} s of host time; the Julia reference itself "
                      f"cannot run here (no Julia toolchain)"}
    parity = {"netflow_rel_err_at_fixed_v": float(np.max(np.abs(psi_dev[:n] - G)) / np.max(np.abs(G))),
              "dual_rel_err": float(abs(psi_dev[n] - acco) / max(abs(acco), 1.0))}
    legs = {k: val for k, val in (route_gpu or {}).items() if k.startswith("_psi") and val is not None}
    if legs and m <= route_budget_pools:
        obj = objective_for(name, n)
        v0 = np.ones(n) if isinstance(obj, cr.LinearNonnegative) else None
        t0 = time.perf_counter()
        ref = orc.route_oracle(oracle_objective(obj), ps, v0=v0, nthreads=threads)
        base["route_ms"] = 1e3 * (time.perf_counter() - t0)
        base["route_evaluations"] = ref["info"]["funcalls"]
        scale = np.max(np.abs(ref["psi"]))
        for key, psi in legs.items():
            parity["route" + key[len("_psi"):] + "_netflow_rel_err"] = float(np.max(np.abs(psi - ref["psi"])) / scale)
        parity.update(fortran_parity(name, legs, m))
        # the KERNEL isolated from the solver: the HIP sweep at the oracle's v* against the oracle's Ψ* (callback given by
        # the caller: it owns the device contexts), and how far the two solvers' v* are apart
        at = route_gpu.get("_sweep_at")
        if at is not None:
            parity["sweep_at_oracle_vstar_rel_err"] = float(np.max(np.abs(at(ref["v"]) - ref["psi"])) / scale)
        for key in ("_v", "_v_native", "_v_sharded"):
            if route_gpu.get(key) is not None:
                parity["vstar_rel_diff" + key[2:]] = float(np.max(np.abs(route_gpu[key] - ref["v"]) / ref["v"]))
        # Route-level parity BY CONVERGENCE.  At the reference's tolerances both sides stop on factr inside
        # the rounding noise of the dual VALUE (the figures above: up to ~2e-6 on interior optima); the same gradient-only
        # polish (router.py::polish_) on both sides -- here on the restatement, from ITS v*, with the device's Jacobian as
        # the chord matrix (only the rate depends on it) -- brings the two netflow vectors to the same point.
        if route_gpu.get("_psi_polished") is not None and route_gpu.get("_J") is not None:
            from helpers import OracleBackend
            ro = cr.Router(obj, batches, n, _backend=OracleBackend(n, batches, threads))
            ro.v[:] = ref["v"]
            t0 = time.perf_counter()
            cr.polish_(ro, jacobian=route_gpu["_J"])
            psi_ref = cr.netflows(ro)
            parity["route_converged_netflow_rel_err"] = float(np.max(np.abs(route_gpu["_psi_polished"] - psi_ref)) / np.max(np.abs(psi_ref)))
            parity["route_converged"] = {
                "is": "max|dPsi|/max|Psi| between the device route! and the CPU restatement's after the same gradient-only polish "
                      "on both sides; `*_from_converged` = how far each side's route! ended from its own converged point",
                "device_from_converged": float(np.max(np.abs(route_gpu.get("_psi_native", route_gpu.get("_psi_sharded")) - route_gpu["_psi_polished"])) / scale)
                if (route_gpu.get("_psi_native") is not None or route_gpu.get("_psi_sharded") is not None) else None,
                "oracle_from_converged": float(np.max(np.abs(ref["psi"] - psi_ref)) / scale),
                "oracle_polish": dict(ro.info["polish"]), "oracle_polish_s": time.perf_counter() - t0}
    return base, parity




def fortran_parity(name, legs, m):
    """Route-level parity against the solver the reference CALLS: tests/golden/route_fortran.npz holds runs of the Fortran
    L-BFGS-B 3.0 `setulb` (the code behind LBFGSB.jl, src/router.jl:60,105) with the reference's call shape on the CPU
    restatement of this very market (tests/golden/make_route_golden.py).  `fortran_reorder_slack` = how far that Fortran run
    ends from ITSELF when the market's pools are listed in another order (the reference sums in pool order): what the
    reference's netflows are defined to at its own tolerances."""
    path = os.path.join(ROOT, "tests", "golden", "route_fortran.npz")
    key = "full_" + name
    try:
        g = np.load(path)
        if key + "_psi" not in g.files or int(np.sum(g[key + "_v"].size)) == 0:
            return {}
        psi_f = g[key + "_psi"]
        out = {"fortran_reorder_slack": float(g[key + "_slack"]), "fortran_evaluations": int(g[key + "_evaluations"]),
               "fortran_reordered_evaluations": [int(x) for x in g[key + "_perm_evaluations"]],
               "fortran_is": "Fortran L-BFGS-B 3.0 setulb (SciPy 1.7.1 F2PY wrap, reference call shape nbd=2 / u=Inf / m=5 / "
                             "factr=1e1 / pgtol=1e-5) on the CPU restatement of this market: tests/golden/route_fortran.npz"}
        for k, psi in legs.items():
            if psi is not None and psi.shape == psi_f.shape:
                out["route" + k[len("_psi"):] + "_vs_fortran_netflow_rel_err"] = float(np.max(np.abs(psi - psi_f)) / np.max(np.abs(psi_f)))
        return out
    except Exception as e:
        return {"fortran_parity_error": repr(e)[:200]}


def other_configs(args, local_rank, budget_s):
    """Every other BASELINE configuration (+ product1m and the multi-tick UniV3 workload) in the SAME default run, so the
    driver's line shows all of them: per workload a short timed pass (K = 20, W = 5: ms/step, pools/s),
    the kernel-event pass (warm kernel time), the HBM-resident pass over a ring sized by touched bytes (kernel time, frac,
    bus_frac from the committed PMC bytes of that workload), fixed-v parity against the CPU restatement (ONE oracle sweep: this
    function is part of the CPU leg) and the library's route! against the Fortran L-BFGS-B fixture.  A few seconds each;
    bounded by `budget_s` (workloads that no longer fit are listed under `skipped`)."""
    from oracle import cfmm_oracle as orc
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import oracle_poolset
    from benchlib.workloads import HBM_PEAK_GBS, alg_bytes
    out, t_start = {}, time.perf_counter()
    try:
        committed = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    except Exception:
        committed = {}
    sub = argparse.Namespace(**vars(args))
    sub.steps, sub.warmup = 20, 5
    for name in ("config2", "config4shard", "config5", "product1m", "univ3_ticks"):
        if time.perf_counter() - t_start > budget_s:
            out.setdefault("skipped", []).append(name)
            continue
        t0 = time.perf_counter()
        try:
            sb = ShardBench(sub, name, "weak", 0, 1, local_rank, False)
            for _ in range(sub.warmup):
                sb.step()
            elapsed, _ = sb.timed_pass(sub.steps)
            kt, _ = sb.kernel_pass(sub.steps)
            psi = sb.out_t.cpu().numpy().copy()
            cold = sb.cold_pass(sub.steps)
            ab = alg_bytes(sb.batches, sb.materialize, sb.v)
            warm_ms = kt["sweep_ms"] / sub.steps
            tr = committed.get(name)
            rec = {"workload": WORKLOADS[name][0], "pools": sb.m_rank, "n_tokens": sb.n,
                   "ms_per_step": 1e3 * elapsed / sub.steps, "value": sb.m_rank * sub.steps / elapsed,
                   "ms_per_step_hbm_resident": cold["ms_per_step"], "value_hbm_resident": sb.m_rank / (cold["ms_per_step"] * 1e-3),
                   "kernel_ms_warm": warm_ms, "kernel_ms_hbm_resident": cold["kernel_ms"], "reduce_kernel_ms": kt["reduce_ms"] / sub.steps,
                   "alg_bytes_per_launch": ab, "frac": cold["frac"], "frac_warm": ab / (warm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                   "traffic": tr, "bus_frac": (tr / (cold["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS) if tr else None,
                   "traffic_is": "committed" if tr else None,
                   "traffic_source": "profiles/traffic.json (committed rocprofv3 PMC passes of this workload, NOT measured in this run)" if tr else None,
                   "step_frac": ab / (1e3 * elapsed / sub.steps * 1e-3) / 1e9 / HBM_PEAK_GBS,
                   "ring": {k: cold[k] for k in ("copies", "touched_per_copy", "bytes_touched", "hbm_resident")}}
            # parity at fixed v: the timed path's {Ψ, acc} against ONE sweep of the CPU restatement
            ps = oracle_poolset(sb.batches, sb.n)
            D, L = np.empty((ps.m, 2)), np.empty((ps.m, 2))
            ps.sweep_into(sb.v, D, L, min(64, HOST_CPUS))
            G = np.zeros(sb.n)
            orc.grad_scatter(G, D, L, ps.Ai)
            acco = orc.dual_acc(D, L, ps.Ai, sb.v)
            rec["parity"] = {"netflow_rel_err_at_fixed_v": float(np.max(np.abs(psi[:sb.n] - G)) / np.max(np.abs(G))),
                             "dual_rel_err": float(abs(psi[sb.n] - acco) / max(abs(acco), 1.0))}
            # route! (cfmm_route, one call) against the Fortran fixture of this market
            obj = objective_for(name, sb.n)
            v0 = np.ones(sb.n) if isinstance(obj, cr.LinearNonnegative) else None
            r = cr.Router(obj, sb.batches, sb.n, device=local_rank)
            try:
                cr.route_(r, v=v0, solver="native")
                ts = []
                for _ in range(3):
                    t1 = time.perf_counter()
                    cr.route_(r, v=v0, solver="native")
                    ts.append(time.perf_counter() - t1)
                rec["route_ms"] = 1e3 * min(ts)
                rec["route_evaluations"] = r.info.get("funcalls")
                rec["parity"].update(fortran_parity(name, {"_psi_native": cr.netflows(r).copy()}, sb.m_rank))
            finally:
                r.close()
            sb.close()
            rec["seconds"] = time.perf_counter() - t0
            out[name] = rec
        except Exception as e:      # informational block: never lose the bench line over it
            out[name] = {"error": repr(e)[:300]}
    # BASELINE config 4 AT ITS STATED SIZE on this one GPU: 4M ProductTwoCoin pools, 512 tokens, 8 shards of
    # 500k pools through ONE multi-device context with the device listed 8 times (cfmm_ctx_create_multi: 8 pool stores, 8 streams,
    # 8 host worker threads; the shards' {Ψ, acc} summed on the host) -- what an 8-GPU node executes, minus the placement.
    # A step here is a host-pointer cfmm_find_arb (PCIe-inclusive: the multi-device context has no device-pointer sweep).
    if time.perf_counter() - t_start > budget_s:
        out.setdefault("skipped", []).append("config4_full")
        return out
    t0 = time.perf_counter()
    try:
        from benchlib.workloads import build_market, sweep_prices_for
        n = WORKLOADS["config4"][1]
        batches = build_market("config4", 0, 1, "weak")
        m = sum(len(b) for b in batches)
        v = sweep_prices_for("config4", n)
        be = cr.DeviceBackend(n, batches, device=[local_rank] * 8)
        try:
            for _ in range(3):
                be.ctx.find_arb(v)
            ts = []
            for _ in range(10):
                t1 = time.perf_counter()
                be.ctx.find_arb(v)
                ts.append(time.perf_counter() - t1)
            psi = np.concatenate([be.ctx.netflows(), [be.ctx.dual_value()]])
            rec = {"workload": WORKLOADS["config4"][0] + "; 8 shards of 500k pools on ONE GPU through one multi-device context",
                   "pools": m, "n_tokens": n, "shards": 8, "ms_per_step": 1e3 * float(np.median(ts)), "ms_per_step_min": 1e3 * min(ts),
                   "value": m / float(np.median(ts)),
                   "value_is": "host-pointer cfmm_find_arb over all 8 shards (v in, Ψ out over PCIe every step; 8 concurrent sweeps "
                               "share the one GPU): pools / median call time"}
            ps = oracle_poolset(batches, n)
            D, L = np.empty((ps.m, 2)), np.empty((ps.m, 2))
            ps.sweep_into(v, D, L, min(64, HOST_CPUS))
            G = np.zeros(n)
            orc.grad_scatter(G, D, L, ps.Ai)
            acco = orc.dual_acc(D, L, ps.Ai, v)
            Dd, Ld = be.trades()
            rec["parity"] = {"netflow_rel_err_at_fixed_v": float(np.max(np.abs(psi[:n] - G)) / np.max(np.abs(G))),
                             "dual_rel_err": float(abs(psi[n] - acco) / max(abs(acco), 1.0)),
                             "all_trade_rows_bit_equal": bool(np.array_equal(Dd, D) and np.array_equal(Ld, L))}
            del D, L, Dd, Ld
            obj = objective_for("config4", n)
            r = cr.Router(obj, batches, n, _backend=be)
            cr.route_(r, v=np.ones(n), solver="native")
            tr = []
            for _ in range(3):
                t1 = time.perf_counter()
                cr.route_(r, v=np.ones(n), solver="native")
                tr.append(time.perf_counter() - t1)
            rec["route_ms"] = 1e3 * min(tr)
            rec["route_evaluations"] = r.info.get("funcalls")
            rec["parity"].update(fortran_parity("config4", {"_psi_native": cr.netflows(r).copy()}, m))
        finally:
            be.close()
        rec["seconds"] = time.perf_counter() - t0
        out["config4_full"] = rec
    except Exception as e:
        out["config4_full"] = {"error": repr(e)[:300]}
    return out


def at_scale_leg(args, local_rank, m=8_000_000, n=256):
    """The same ProductTwoCoin sweep where the launch floor no longer matters: 8M pools (320 MB touched per sweep: HBM-resident
    by size, rotated over 3 copies all the same), kernel span by CP events -- the roofline figure of the KERNEL rather than of
    the 1M-pool configuration (scripts/size_scaling.py prints the whole curve: profiles/r05_size_scaling.txt)."""
    from benchlib.workloads import HBM_PEAK_GBS, ring_copies, touched_bytes
    t0 = time.perf_counter()
    batch = [synth_mod().product_pools(m, n, seed=1234)]
    copies = ring_copies(touched_bytes(batch, True))
    ring = [cr.DeviceBackend(n, batch, device=local_rank) for _ in range(copies)]
    try:
        stream = torch.cuda.current_stream()
        v_t = torch.from_numpy(synth_mod().sweep_prices(n, seed=1234)).to("cuda")
        out_t = torch.zeros(n + 1, dtype=torch.float64, device="cuda")
        for b_ in ring:
            b_.ctx.set_stream(stream.cuda_stream)
        K = 8 * copies
        for k in range(2 * copies):
            ring[k % copies].ctx.sweep_dev(v_t.data_ptr(), out_t.data_ptr(), True)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for k in range(K):
            ring[k % copies].ctx.sweep_dev(v_t.data_ptr(), out_t.data_ptr(), True)
        torch.cuda.synchronize()
        step_ms = 1e3 * (time.perf_counter() - t1) / K
        for b_ in ring:
            b_.ctx.set_option("time_kernels", 1)
            b_.ctx.kernel_times()
        for k in range(K):
            ring[k % copies].ctx.sweep_dev(v_t.data_ptr(), out_t.data_ptr(), True)
        torch.cuda.synchronize()
        sw = sum(b_.ctx.kernel_times()["sweep_ms"] for b_ in ring) / K
    finally:
        for b_ in ring:
            b_.close()
    return {"workload": f"{m} ProductTwoCoin pools, {n} tokens, materialising, HBM-resident ({copies} copies x {touched_bytes(batch, True) / 1e6:.0f} MB touched)",
            "pools": m, "kernel_ms": sw, "ms_per_step": step_ms, "value": m / (step_ms * 1e-3),
            "frac": 64.0 * m / (sw * 1e-3) / 1e9 / HBM_PEAK_GBS, "bus_frac": 40.0 * m / (sw * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "is": "frac = reference-layout bytes (64 B per pool, SURVEY 8d) / kernel time / 8 TB/s; bus_frac = the 40 B per pool this "
                  "layout moves (24 read + 16 written) / kernel time / 8 TB/s (a streaming copy reaches ~0.79 of 8 TB/s)",
            "seconds": time.perf_counter() - t0}


def synth_mod():
    from cfmmrouter_amd import synth
    return synth


def summary_string(rec):
    """<= 120 characters: what a parser that keeps only scalars of `config` still shows of a workload"""
    if "error" in rec:
        return ("error: " + rec["error"])[:120]
    p = rec.get("parity", {})
    return ("step %.1fus hbm %.1fus | kern %.2f/%.2fus | frac %.2f bus %s | fixv %.0e fortran %s slack %s"
            % (1e3 * rec["ms_per_step"], 1e3 * rec["ms_per_step_hbm_resident"], 1e3 * rec["kernel_ms_warm"],
               1e3 * rec["kernel_ms_hbm_resident"], rec["frac"], ("%.2f" % rec["bus_frac"]) if rec.get("bus_frac") else "-",
               p.get("netflow_rel_err_at_fixed_v", float("nan")),
               ("%.0e" % p["route_native_vs_fortran_netflow_rel_err"]) if "route_native_vs_fortran_netflow_rel_err" in p else "-",
               ("%.0e" % p["fortran_reorder_slack"]) if "fortran_reorder_slack" in p else "-"))[:120]


def scaling_grid_main(args):
    """--workload scaling: the reference's published benchmark (benchmark/scaling.jl:8-38) on the GPU path, the CPU
    restatement timed per point beside it (1 and 8 threads, the better one reported: the reference's thread count is not
    stated, its machine had 8 cores), and BASELINE.md's read-off plot values where the plot has them."""
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU fallback for the product path")
    from benchlib.grid import run_grid

    def cpu_route(obj, market, n, v0, psi_dev):
        from oracle import cfmm_oracle as orc
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from helpers import oracle_objective, oracle_poolset
        ps, best = oracle_poolset(market, n), None
        for threads in (1, 8):
            ts = []
            for _ in range(3):
                t0 = time.perf_counter()
                ref = orc.route_oracle(oracle_objective(obj), ps, v0=v0, nthreads=threads)
                ts.append(time.perf_counter() - t0)
            if best is None or min(ts) < best[0]:
                best = (min(ts), threads, ref)
        ref = best[2]
        return {"cpu_restatement_ms": 1e3 * best[0], "cpu_threads": best[1], "cpu_evaluations": int(ref["info"]["funcalls"]),
                "netflow_rel_err": float(np.max(np.abs(psi_dev - ref["psi"])) / np.max(np.abs(ref["psi"])))}

    rows = run_grid(0, None if args.no_cpu else cpu_route)
    big = rows[-3]     # m = 10 000, sqrt(m) tokens: the point BASELINE.md reads 0.19 s for
    line = {"metric": METRIC, "value": big["m"] * big["native_evaluations"] / (big["native_ms"] * 1e-3), "unit": "pools/s",
            "value_is": "pool-evaluations per second INSIDE route! at the largest grid point (m = 10 000, 100 tokens); the "
                        "grid's route! wall-clocks are under `grid`", "n_gpus": 1, "steps": len(rows), "warmup": 1,
            "ms_per_step": big["native_ms"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic", "config": {"workload": "scaling: the reference's benchmark grid (benchmark/scaling.jl:8-38): "
                                                        "route! on 100 .. 10 000 ProductTwoCoin pools, {1, 2, 4}·sqrt(m) tokens"},
            "grid": rows,
            "grid_is": "route! wall-clock per point, median: native = cfmm_route (one C-ABI call, pre-armed evaluations), "
                       "scipy_driven = SciPy's L-BFGS-B with one C-ABI call per evaluation, cpu_restatement = the C restatement "
                       "of the reference path driven by SciPy on this box's host cores (NOT the Julia reference), "
                       "reference_plot_ms = BASELINE.md's read-off of the reference's plot (MacBook Pro 2.3 GHz i9, +-15 %)"}
    print(json.dumps(line))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="config3", choices=sorted(WORKLOADS) + ["scaling"])
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = one config-sized shard per GPU (default); strong = the config's pools divided among the GPUs")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline / parity / route / host-boundary legs")
    ap.add_argument("--no-strong", action="store_true", help="N > 1: skip the additional strong-scaling leg (config 4)")
    ap.add_argument("--fused", action="store_true", help="time the fused evaluation (no Δ/Λ write-back)")
    ap.add_argument("--opt", action="append", default=[], help="library option key=value")
    ap.add_argument("--no-cold", action="store_true", help="skip the cache-cold pass")
    ap.add_argument("--no-configs", action="store_true",
                    help="default workload at N = 1: skip the `configs` block (the other BASELINE configurations in the same run)")
    ap.add_argument("--configs-budget-s", type=float, default=150.0, help="time budget of the `configs` block")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure roofline.traffic with rocprofv3 PMC passes of this command (two bounded child runs); "
                         "use the committed profiles/traffic.json")
    ap.add_argument("--collective", default="auto", choices=["auto", "peer", "rccl_library", "rccl_torch"],
                    help="N > 1: the all-reduce of {Psi, acc} the HEADLINE step uses (auto = the first that works of: the library's "
                         "one-launch peer gather, ncclAllReduce inside the library, torch.distributed); all three are timed side by "
                         "side under `collectives` whatever is chosen here")
    ap.add_argument("--rccl", action="store_true", help="alias of --collective rccl_library (north_star's collective as the headline)")
    ap.add_argument("--no-collectives", action="store_true", help="N > 1: skip the side-by-side `collectives` block")
    ap.add_argument("--cold-only", action="store_true",
                    help="the timed region rotates over > 300 MB of market copies (pool state from HBM, not the Infinity "
                         "Cache): used for the rocprofv3 summary of the HBM-resident figure")
    ap.add_argument("--single-process", action="store_true",
                    help="--gpus N from ONE process: the multi-device context of the C ABI (cfmm_ctx_create_multi), a step is "
                         "one host-pointer cfmm_find_arb over all N shards (PCIe-inclusive)")
    ap.add_argument("--devices", default="", help="--single-process: comma-separated HIP ordinals (default 0..N-1; an "
                                                  "ordinal may repeat to put several shards on one GPU)")
    args = ap.parse_args()
    if args.rccl:
        args.collective = "rccl_library"

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.workload == "scaling":
        return scaling_grid_main(args)
    if args.single_process:
        return single_process_main(args, cpu_baseline_leg)
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
    # CFMM_BENCH_SHARE_GPU=1 (rehearsal of the N > 1 orchestration on a 1-GPU box: shard indexing, global-market parity,
    # max over ranks, the strong-scaling leg): the ranks share the visible GPUs round-robin and rendezvous over gloo --
    # RCCL does not share a device between ranks.  The line it prints is NOT a measurement (`config.rehearsal`).
    share = os.environ.get("CFMM_BENCH_SHARE_GPU") == "1"
    if share and torch.cuda.device_count() > 0:
        local_rank %= torch.cuda.device_count()
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
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    sb = ShardBench(args, args.workload, args.scaling, rank, world, local_rank, use_dist)
    n, be, materialize = sb.n, sb.be, sb.materialize
    if args.cold_only and not use_dist:
        sb.use_ring()
    for _ in range(args.warmup):
        sb.step()

    # pass 1 -- THE timed region: K steps between barrier + synchronize (ShardBench.timed_pass), nothing else on the stream
    # or the host; the MAX over ranks is reported
    elapsed, _ = sb.timed_pass(args.steps)
    elapsed = sb.max_over_ranks(elapsed)
    ms_per_step = 1e3 * elapsed / args.steps
    value = world * sb.m_rank * args.steps / elapsed

    # pass 2 -- the same K steps with kernel events (roofline), out of the timed region
    kt, elapsed2 = sb.kernel_pass(args.steps)
    sweep_ms = sb.max_over_ranks(kt["sweep_ms"] / max(args.steps, 1))      # all sweep launches of one step, slowest rank
    reduce_ms = kt["reduce_ms"] / max(args.steps, 1)
    psi_timed = None if sb.ring is not None else sb.out_t.cpu().numpy().copy()   # what the timed path left behind (global at N > 1)
    # pass 3 -- HBM-resident sweeps (rotation over > 300 MB of market copies), every rank
    cold = sb.cold_pass(args.steps) if not args.no_cold and not args.cold_only else None
    collective_check = sb.collective_check() if use_dist else None
    if psi_timed is None:
        be.ctx.sweep_dev(sb.v_t.data_ptr(), sb.out_t.data_ptr(), materialize)
        torch.cuda.synchronize()
        psi_timed = sb.out_t.cpu().numpy().copy()

    route_sharded, psi_sharded, v_sharded, sr, hidden = None, None, None, None, {}
    if use_dist:
        route_sharded, psi_sharded, v_sharded, sr, hidden = sharded_route(sb, local_rank)
    host = host_boundary_leg(sb) if world == 1 and not use_dist and not args.no_cpu else {}

    traffic, traffic_src, traffic_detail = None, None, None
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    if world == 1 and not use_dist and not args.no_cpu and not args.no_live_traffic:
        traffic, traffic_detail = live_traffic(__file__, args.workload, args.fused, args.opt)
        if traffic is not None:
            traffic_src = ("measured in this run: HBM bytes per sweep launch from two rocprofv3 PMC passes of this command "
                           "(--pmc FETCH_SIZE / --pmc WRITE_SIZE in separate child runs with --kernel-trace only; KiB, "
                           "FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md)")
        else:
            traffic_src = f"live PMC pass unavailable ({traffic_detail}); "
            traffic_detail = None
    if traffic is None and os.path.exists(tf):
        try:
            traffic = json.load(open(tf)).get(args.workload + ("_fused" if args.fused else ""))
            traffic_src = (traffic_src or "") + (
                "HBM bytes per launch from the committed rocprofv3 PMC passes of this command (FETCH_SIZE, WRITE_SIZE "
                "in separate runs, gfx950 corrections of MI355X_MICROARCH.md; profiles/traffic.json)")
        except Exception:
            traffic = None

    line = {
        "metric": METRIC, "value": value, "unit": "pools/s",
        "value_is": "find_arb! pool-evaluations per second (materialising sweep + Ψ/dual reduction); "
                    "route! wall-clock is reported under `route`", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {sb.desc}", "pools_per_gpu": sb.m_rank, "pools_total": world * sb.m_rank,
                   "n_tokens": n, "variant": "materialising" if materialize else "fused", "segments": be.ctx.segments(),
                   "sharding": sb.sharding_text(), **({"rehearsal": "ranks share a GPU over gloo: not a measurement"} if share else {})},
        "roofline": roofline_record(sb, args, ms_per_step, sweep_ms, reduce_ms, elapsed2, cold, traffic, traffic_src, traffic_detail),
        "library_options": {k: be.ctx.get_option(k) for k in ("pack", "compact_trades", "alternate", "fast_math", "armed",
                                                               "stop_in_noise", "host_flag", "zero_copy")},
    }
    if world == 1 and not use_dist and not args.no_cpu and not args.no_live_traffic and not args.cold_only and cold is not None:
        # the HBM-resident kernel span as rocprofv3 reports it (a --cold-only child of this command under --kernel-trace --stats):
        # `kernel_ms`, `frac` and `frac_bus` are quoted on THAT clock -- the one profiles/ holds -- with the hipEvent figure beside it
        ks, why = live_kernel_stats(__file__, args.workload, args.fused, args.opt)
        roof = line["roofline"]
        roof["kernel_ms_hip_events"] = roof["kernel_ms"]
        if ks is not None and ks["avg_ms"] > 0:
            ab = roof["alg_bytes_per_launch"]
            roof.update(kernel_ms=ks["avg_ms"], kernel_ms_source="rocprofv3 --kernel-trace --stats of a --cold-only child run of this "
                        "command: average of %d launches of %s (min %.2f us, max %.2f us)" % (ks["calls"], ks["kernel"], 1e3 * ks["min_ms"], 1e3 * ks["max_ms"]),
                        achieved=ab / (ks["avg_ms"] * 1e-3) / 1e9, frac=ab / (ks["avg_ms"] * 1e-3) / 1e9 / roof["peak"])
            if roof.get("traffic"):
                roof["bus_frac"] = roof["frac_bus"] = roof["traffic"] / (ks["avg_ms"] * 1e-3) / 1e9 / roof["peak"]
            roof["frac_hip_events"] = ab / (roof["kernel_ms_hip_events"] * 1e-3) / 1e9 / roof["peak"]
        else:
            roof["kernel_ms_source"] = "hipEvent pairs written by the command processor around every sweep launch (rocprofv3 child unavailable: %s)" % why
    if world == 1 and not use_dist and not args.no_cpu and not args.fused and not args.cold_only:
        try:
            line["roofline"]["expanded"] = expanded_leg(sb, args, local_rank)
        except Exception as e:
            line["roofline"]["expanded"] = {"error": repr(e)[:200]}
    if cold is not None:
        # the HBM-resident step beside `value` (which is the cache-warm step: the same market every step, as inside route!)
        line["value_hbm_resident"] = world * sb.m_rank / (cold["ms_per_step"] * 1e-3)
        line["ms_per_step_hbm_resident"] = cold["ms_per_step"]
        line["value_hbm_resident_is"] = ("pool-evaluations per second of the SAME step when every sweep reads its pool state from "
                                         "HBM (steps rotate over roofline.cold.copies market copies, roofline.cold.bytes_touched bytes "
                                         ">= 2 x the 256 MiB Infinity Cache); `value` is the cache-warm step")
        line["config"]["hbm_resident"] = "step %.2fus kernel %.2fus frac %.3f ring %d copies %.0fMB touched" % (
            1e3 * cold["ms_per_step"], 1e3 * cold["kernel_ms"], cold["frac"], cold["copies"], cold["bytes_touched"] / 1e6)
    if collective_check is not None:
        line["collective_check_rel_err"] = collective_check
        line["collective"] = {"headline": sb.collective, "requested": args.collective,
                              **({"requested_unavailable": sb.why_not} if sb.why_not else {}),
                              **({"auto_fell_back_because": sb.fell_back} if getattr(sb, "fell_back", None) else {})}
    if use_dist and not args.no_collectives:
        # the same step under all three all-reduces, side by side (one run answers which collective the curve should use)
        try:
            line["collectives"] = collectives_leg(args, args.workload, args.scaling, rank, world, local_rank, sb)
        except Exception as e:
            line["collectives"] = {"error": repr(e)[:300]}
        line["collectives_is"] = ("ms_per_step of the SAME step (K steps between barrier + synchronize, max over ranks) under each all-reduce "
                                  "of the n_tokens+1 doubles: peer = cfmm_set_peers (fold + xGMI gather in one launch), rccl_library = "
                                  "ncclAllReduce enqueued by the library behind the fold (cfmm_rccl_init_rank; north_star's collective), "
                                  "rccl_torch = torch.distributed.all_reduce on the sweep's stream; kernel_ms_min/max = the sweep kernel's "
                                  "mean span on the fastest / slowest rank; `value` of the line is the `collective.headline` one")
        if world == 1:
            # N = 1 under torchrun against the plain N = 1 step in the SAME process (same box, same clocks): the sharded
            # machinery at world 1 (fold + gather launch with one rank) must cost nothing
            try:
                plain = ShardBench(args, args.workload, args.scaling, 0, 1, local_rank, False)
                k1 = max(args.steps, 300)          # (K = 20 steps of 11 us are a 0.2 ms region: +-3 % of jitter by themselves)
                for _ in range(max(args.warmup, 20)):
                    plain.step()
                    sb.step()
                pairs = []
                for _ in range(3):                 # interleaved: both see the same clocks
                    e_sh, _ = sb.timed_pass(k1)
                    e_pl, _ = plain.timed_pass(k1)
                    pairs.append((e_sh, e_pl))
                plain.close()
                torch.cuda.set_stream(sb.stream)
                e_sh, e_pl = min(p_[0] for p_ in pairs), min(p_[1] for p_ in pairs)
                line["plain_n1"] = {"ms_per_step": 1e3 * e_pl / k1, "ms_per_step_under_torchrun": 1e3 * e_sh / k1,
                                    "ratio_torchrun_over_plain": e_sh / e_pl, "steps": k1,
                                    "is": "the same steps on a plain single-GPU context (no process group) and on this run's sharded "
                                          "context (world 1), interleaved in this process, best of 3 passes each"}
            except Exception as e:
                line["plain_n1"] = {"error": repr(e)[:200]}
    if route_sharded is not None:
        line["route_sharded"] = route_sharded
    if host:
        line["host_boundary"] = host

    # N > 1 with the default workload: BASELINE's sharded configuration (config 4: 4M ProductTwoCoin pools, 512 tokens)
    # divided among the GPUs -- the strong-scaling figure beside the weak-scaling headline
    if use_dist and world > 1 and args.workload == "config3" and args.scaling == "weak" and not args.no_strong:
        try:
            s4 = ShardBench(args, "config4", "strong", rank, world, local_rank, use_dist)
            for _ in range(args.warmup):
                s4.step()
            e4, _ = s4.timed_pass(args.steps)
            e4 = s4.max_over_ranks(e4)
            chk = s4.collective_check()
            line["strong_scaling"] = {"workload": "config4: " + s4.desc, "pools_total": 4_000_000, "pools_per_gpu": s4.m_rank,
                                      "ms_per_step": 1e3 * e4 / args.steps, "value": 4_000_000 * args.steps / e4, "unit": "pools/s",
                                      "scaling": "strong", "sharding": s4.sharding_text(), "collective": s4.collective,
                                      "collective_check_rel_err": chk}
            if not args.no_collectives:
                line["strong_scaling"]["collectives"] = collectives_leg(args, "config4", "strong", rank, world, local_rank, s4)
            s4.close()
        except Exception as e:
            line["strong_scaling"] = {"error": repr(e)[:300]}
        torch.cuda.set_stream(sb.stream)

    # route! on one GPU, the CPU restatement beside everything, and the parity of every GPU leg against it (rank 0;
    # at N > 1 on the GLOBAL market, regenerated on the host)
    if rank == 0 and not args.no_cpu:
        local = None
        if not use_dist:
            try:
                route_gpu = route_leg(args.workload, sb.batches, n, local_rank)
            except Exception as e:  # the route leg is informational; never lose the bench line over it
                route_gpu = {"error": repr(e)[:300]}
        else:
            route_gpu = {"_psi_sharded": psi_sharded, "_v_sharded": v_sharded, **hidden}
            if world == 1:        # world 1 under torchrun: the rank's shard IS the market
                local = cr.DeviceBackend(n, sb.batches, device=local_rank)
                route_gpu["_sweep_at"] = lambda vv: local.eval(vv)[0]
        try:
            global_batches = sb.batches if world == 1 else build_global(args.workload, world, args.scaling)
            line["cpu_baseline"], line["parity"] = cpu_baseline_leg(args.workload, global_batches, n, sb.v, psi_timed, route_gpu)
            line["parity"]["market"] = f"{sum(len(b) for b in global_batches)} pools (all ranks' shards), checked on rank 0"
        except Exception as e:
            line["cpu_baseline"], line["parity"] = {"error": repr(e)[:300]}, {"error": repr(e)[:300]}
        if local is not None:
            local.close()
        if not use_dist:
            line["route"] = finish_route(route_gpu)
    if sr is not None:
        sr.close()
    sb.close()
    if (rank == 0 and world == 1 and not use_dist and args.workload == "config3" and not args.no_cpu and not args.no_configs
            and not args.cold_only and not args.no_cold and not os.environ.get("CFMM_BENCH_CHILD")):
        line["configs"] = other_configs(args, local_rank, args.configs_budget_s)
        line["configs_is"] = ("the other BASELINE configurations measured in this same run (K = 20, W = 5 each): ms/step, HBM-resident "
                              "kernel time + frac, bus_frac from the committed PMC bytes, parity at fixed v vs the CPU restatement "
                              "and route! vs the Fortran L-BFGS-B fixture; `config.cfg_*` repeat them as one-line strings")
        for k, rec in line["configs"].items():
            if isinstance(rec, dict) and k != "config4_full":
                line["config"]["cfg_" + k] = summary_string(rec)
        c4 = line["configs"].get("config4_full")
        if isinstance(c4, dict) and "error" not in c4:
            line["config"]["cfg_config4_full"] = ("4M pools 8 shards 1 GPU: find_arb %.2fms | rows bit-equal %s fixv %.0e | route! %.2fms %s evals fortran %s" % (
                c4["ms_per_step"], c4["parity"]["all_trade_rows_bit_equal"], c4["parity"]["netflow_rel_err_at_fixed_v"], c4["route_ms"],
                c4["route_evaluations"], ("%.0e" % c4["parity"]["route_native_vs_fortran_netflow_rel_err"])
                if "route_native_vs_fortran_netflow_rel_err" in c4["parity"] else "-"))[:160]
        try:     # the kernel against the roofline where the launch floor no longer matters
            sc = at_scale_leg(args, local_rank)
            line["roofline"]["at_scale"] = sc
            line["config"]["at_scale"] = "8M Product pools HBM-resident: kernel %.1fus = frac %.2f (ref. bytes) bus %.2f of 8 TB/s; %.2e pools/s" % (
                1e3 * sc["kernel_ms"], sc["frac"], sc["bus_frac"], sc["value"])
        except Exception as e:
            line["roofline"]["at_scale"] = {"error": repr(e)[:200]}
    if rank == 0:
        print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
