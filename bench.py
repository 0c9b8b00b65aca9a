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

    --scaling weak   (default)  every GPU sweeps one config-sized shard (config3: 1M pools per GPU)
    --scaling strong            the workload's pool count is divided among the GPUs (config4: 4M / N per GPU)

Extra keys on the line: roofline (dominant kernel, hipEvents attached to every sweep launch; `frac` is the
HBM-resident figure), route (route! wall-clock on the same market through the GPU path), cpu_baseline (the C
restatement of the reference path on the host cores: sweep throughput on a bounded sample and route! wall-clock)
and parity (the GPU legs' netflows against that restatement) -- at every N: at N > 1 rank 0 regenerates the GLOBAL
market on the host for the oracle.  At N > 1 with the default workload the line also carries `strong_scaling`
(config 4: 4M pools / N per GPU) measured after the main timed region.
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
METRIC = "find_arb! pools/sec + route! wall-clock, 1M-pool arbitrage, 1/2/4/8 GPU"

# algorithmic bytes per pool-evaluation, materialising sweep (SURVEY.md §8d / DESIGN.md):
#   read pool state + write Δ(16 B) + Λ(16 B)
ALG_BYTES = {KIND_PRODUCT: 32 + 32, KIND_GEOMEAN: 48 + 32}
ALG_BYTES_FUSED = {KIND_PRODUCT: 32, KIND_GEOMEAN: 48}


def live_traffic(workload, fused, opts):
    """roofline.traffic measured NOW: HBM bytes per sweep launch from two bounded rocprofv3 PMC passes of this very command
    (`--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` in separate runs with --kernel-trace only, MI355X_MICROARCH.md §HBM: both
    in KiB, FETCH_SIZE doubled on gfx950).  Returns (bytes, detail) or (None, reason).  The child runs are this script with
    --no-cpu --no-cold --no-live-traffic; each is bounded, and on a time-out exactly the process group started here is
    killed."""
    import csv
    import glob
    import shutil
    import signal
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    if os.environ.get("CFMM_BENCH_CHILD") or any(k.startswith("ROCPROF") for k in os.environ):
        return None, "already inside a profiled run"
    tmp = tempfile.mkdtemp(prefix="cfmm_pmc_", dir="/tmp")
    env = dict(os.environ, CFMM_BENCH_CHILD="1", TMPDIR="/tmp")
    counters = {}
    try:
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = [exe, "--pmc", c, "--kernel-trace", "--output-format", "csv", "-d", os.path.join(tmp, c), "-o", "w", "--",
                   sys.executable, os.path.abspath(__file__), "--steps", "20", "--warmup", "3", "--no-cpu", "--no-cold",
                   "--no-live-traffic", "--workload", workload] + (["--fused"] if fused else [])
            for o in opts:
                cmd += ["--opt", o]
            p = subprocess.Popen(cmd, cwd="/tmp", env=env, stdin=subprocess.DEVNULL, stdout=subprocess.DEVNULL,
                                 stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                p.wait(timeout=90)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, signal.SIGKILL)     # the session started above, nothing else
                p.wait()
                return None, f"rocprofv3 --pmc {c} timed out"
            acc = {}
            for f in glob.glob(os.path.join(tmp, c, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if row.get("Counter_Name") == c and "cfmm::sweep" in row["Kernel_Name"]:
                        acc.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
            if not acc:
                return None, f"rocprofv3 --pmc {c}: no counter rows (exit code {p.returncode})"
            counters[c] = {k: sum(v) / len(v) for k, v in acc.items()}
        tag = ("<false,", " false,") if fused else ("<true,", " true,")
        total, detail = 0.0, {}
        for k in sorted(set(counters["FETCH_SIZE"]) | set(counters["WRITE_SIZE"])):
            if not any(t in k for t in tag):
                continue
            f, w = counters["FETCH_SIZE"].get(k, 0.0), counters["WRITE_SIZE"].get(k, 0.0)
            detail[k] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0}
            total += detail[k]["hbm_bytes_per_launch"]
        return (total, detail) if detail else (None, "no sweep kernel of this variant in the counter rows")
    except Exception as e:      # a profiler problem must not cost the bench line
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def alg_bytes(batches, materialize=True, v=None):
    """SURVEY §8d bytes of one launch.  UniV3: 32 B header + 16 B per tick (+ 32 B of trades); multi-tick ladders
    (`v` given): per tick VISITED by the walk at these prices, not per tick stored."""
    tot = 0
    for b in batches:
        if b.kind == KIND_UNIV3:
            ticks = b.lower_ticks.size
            if v is not None and ticks > 2 * len(b):
                ticks = int(np.sum(np.maximum(synth.univ3_ticks_visited(b, v), 1)))   # an idle pool still reads its current tick
            tot += len(b) * 32 + 16 * ticks + (32 * len(b) if materialize else 0)
        else:
            tot += len(b) * (ALG_BYTES if materialize else ALG_BYTES_FUSED)[b.kind]
    return tot


# name: (description, n_tokens, [(generator, pools per GPU (weak) = pools in total (strong), kwargs)])
WORKLOADS = {
    "config2": ("100k ProductTwoCoin pools, 64 tokens, LinearNonnegative arbitrage", 64,
                [(synth.product_pools, 100_000, {})]),
    "config3": ("1M mixed ProductTwoCoin + GeometricMeanTwoCoin pools (500k each), 256 tokens, "
                "LinearNonnegative arbitrage", 256,
                [(synth.product_pools, 500_000, {}), (synth.geomean_pools, 500_000, {})]),
    "config4shard": ("500k ProductTwoCoin pools per GPU (4M over 8 GPUs), 512 tokens", 512,
                     [(synth.product_pools, 500_000, {})]),
    "config4": ("4M ProductTwoCoin pools in total, 512 tokens (BASELINE config 4; --scaling strong: 4M / N per GPU)", 512,
                [(synth.product_pools, 4_000_000, {})]),
    "config5": ("1M BoundedProduct (2-tick UniV3) pools quoted around one token price vector (1 % noise), 256 tokens, "
                "BasketLiquidation (interior dual optimum)", 256,
                [(synth.bounded_product_pools, 1_000_000, {"consistent": True})]),
    "config5corner": ("1M BoundedProduct pools with independent random prices (arbitrage-rich: route! ends at the box "
                      "corner after 2 evaluations), 256 tokens, BasketLiquidation", 256,
                      [(synth.bounded_product_pools, 1_000_000, {})]),
    "univ3_ticks": ("1M UniV3 pools with ragged ladders of 2..64 initialised ticks (17 on average) quoted around one token "
                    "price vector, 256 tokens; at the sweep's prices 3/4 of the pools walk through more than one tick", 256,
                    [(synth.univ3_ragged_pools, 1_000_000, {})]),
    "large_n": ("1M ProductTwoCoin pools, 65536 tokens (global-bin path), LinearNonnegative arbitrage", 65536,
                [(synth.product_pools, 1_000_000, {})]),
    "product1m": ("1M ProductTwoCoin pools, 256 tokens, LinearNonnegative arbitrage", 256,
                  [(synth.product_pools, 1_000_000, {})]),
}


def shard_range(m, rank, world):
    base, rem = divmod(int(m), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def build_market(name, rank, world, scaling):
    """This rank's shard of the workload: weak = one config-sized shard per rank (pool indices [rank*m, (rank+1)*m) of
    the generator's stream), strong = a contiguous 1/world of the config's pools.  The generators are pure functions of
    (seed, pool index), so the shards of all ranks concatenated ARE the global market (`build_global`)."""
    _, n, parts = WORKLOADS[name]
    out = []
    for gen, m, kw in parts:
        if scaling == "strong":
            lo, hi = shard_range(m, rank, world)
        else:
            lo, hi = rank * m, (rank + 1) * m
        out.append(gen(hi - lo, n, seed=1234, first=lo, **kw))
    return out


def build_global(name, world, scaling):
    _, n, parts = WORKLOADS[name]
    return [gen(m if scaling == "strong" else world * m, n, seed=1234, first=0, **kw) for gen, m, kw in parts]


def sweep_prices_for(name, n):
    v = synth.sweep_prices(n, seed=1234)
    if name == "univ3_ticks":     # the ladders are quoted around the token price vector: sweep a few per cent off it
        v = v * synth.token_price_vector(n, seed=1234)
    return v


def objective_for(name, n):
    if name.startswith("config5") or name == "univ3_ticks":
        return cr.BasketLiquidation(1, synth.basket(n, seed=1234))
    return cr.LinearNonnegative(synth.linear_prices(n, seed=1234))


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
    threads = orc.lib().oracle_max_threads()
    if "TORCHELASTIC_RUN_ID" in os.environ:   # torchrun exports OMP_NUM_THREADS=1; rank 0 is the only rank doing host work here
        threads = max(threads, min(128, max(1, len(os.sched_getaffinity(0)) // 2)))   # one per physical core (SMT siblings only add contention to this memory-bound loop: measured 4e6 pools/s on 256 threads vs 2.3e7 on 128)
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
        # the KERNEL isolated from the solver: the HIP sweep at the oracle's v* against the oracle's Ψ* (callback given by
        # the caller: it owns the device contexts), and how far the two solvers' v* are apart
        at = route_gpu.get("_sweep_at")
        if at is not None:
            parity["sweep_at_oracle_vstar_rel_err"] = float(np.max(np.abs(at(ref["v"]) - ref["psi"])) / scale)
        for key in ("_v", "_v_native", "_v_sharded"):
            if route_gpu.get(key) is not None:
                parity["vstar_rel_diff" + key[2:]] = float(np.max(np.abs(route_gpu[key] - ref["v"]) / ref["v"]))
        # how far apart two runs of the ALGORITHM itself end up when v0 moves by 1e-16 .. 1e-13 (relative): the yardstick for
        # the route-level figures above (interior optima -- BasketLiquidation on a consistent market -- are pinned to ~2e-6
        # of max|Psi| only; arbitrage markets to 1e-9 .. 1e-6)
        if base["route_ms"] < 15e3:
            v0p = np.ones(n) if isinstance(obj, cr.LinearNonnegative) else np.ones(n) / n
            first = 0 if isinstance(obj, cr.LinearNonnegative) else 1
            hull = 0.0
            for eps in (1e-16, 1e-13):
                alt = orc.route_oracle(oracle_objective(obj), ps, v0=v0p * (1 + eps * np.arange(n)), nthreads=threads)
                hull = max(hull, float(np.max(np.abs(alt["psi"][first:] - ref["psi"][first:])) / scale))
            parity["oracle_scatter_hull"] = hull
            parity["oracle_scatter_hull_is"] = ("max|dPsi|/max|Psi| between CPU-restatement route! runs whose v0 differs by "
                                               "1e-16 / 1e-13 relative: what the algorithm itself pins Psi* to")
    return base, parity


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


def single_process_main(args):
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


class ShardBench:
    """One rank's timed machinery for one workload: backend, stream, peer buffers (N > 1), the step."""

    def __init__(self, args, name, scaling, rank, world, local_rank, use_dist):
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
        if use_dist and not args.rccl and os.environ.get("CFMM_AMD_NO_PEER", "0") != "1":
            self.setup_peers()

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
            be.ctx.set_peers(self.peer_ptrs, world, rank, self.n_fused)

    def step(self):
        if self.ring is not None:     # rotate over enough copies of the market to exceed the 256 MB Infinity Cache
            b_ = self.ring[self.ring_pos % len(self.ring)]
            self.ring_pos += 1
            b_.ctx.sweep_dev(self.v_t.data_ptr(), self.out_t.data_ptr(), self.materialize)
            return
        self.be.ctx.sweep_dev(self.v_t.data_ptr(), self.out_t.data_ptr(), self.materialize)   # sharded context: already the global {Ψ, acc}
        if self.use_dist and not self.fused_peer:
            dist.all_reduce(self.out_t)  # Ψ and the dual scalar: one small RCCL collective per evaluation
        self.steps_run += 1

    def market_copies(self):
        per_copy = alg_bytes(self.batches, True) + 16 * sum(len(b) for b in self.batches if b.kind == KIND_GEOMEAN)
        return per_copy, int(np.ceil(320e6 / per_copy)) + 1

    def use_ring(self):
        """--cold-only: the TIMED steps rotate over > 300 MB of market copies (no collective: local sweeps)."""
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
        """HBM-resident figure: every working set here (<= 100 MB) fits the 256 MB Infinity Cache, so the
        timed passes are "warm" (what a running route! sees).  Rotating LOCAL sweeps over enough distinct copies of this
        rank's shard to exceed 300 MB makes every sweep read its pool state from HBM.  Every rank runs it (N > 1: the
        slowest rank's kernel time is reported)."""
        per_copy, copies = self.market_copies()
        extra = [cr.DeviceBackend(self.n, self.batches, device=self.local_rank) for _ in range(copies - 1)]
        sharded = self.fused_peer
        if sharded:
            self.be.ctx.set_peers([], 0, 0, 0)
        ring = [self.be] + extra
        outs = [torch.zeros(self.n + 1, dtype=torch.float64, device="cuda") for _ in ring]
        for b_ in extra:
            b_.ctx.set_stream(self.stream.cuda_stream)
            self.apply_options(b_)
        for k in range(2 * copies):
            ring[k % copies].ctx.sweep_dev(self.v_t.data_ptr(), outs[k % copies].data_ptr(), self.materialize)
        torch.cuda.synchronize()
        for b_ in ring:
            b_.ctx.set_option("time_kernels", 1)
            b_.ctx.kernel_times()
        cold_steps = max(steps, 60)     # a stable average: at the driver's K = 20 the figure moves by +-0.02
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
        cold = {"copies": copies, "bytes_rotated": copies * per_copy, "kernel_ms": sw,
                "achieved": ab / (sw * 1e-3) / 1e9 if sw > 0 else 0.0,
                "ms_per_step_with_kernel_events": 1e3 * cold_elapsed / cold_steps, "sweeps": cold_steps}
        cold["frac"] = cold["achieved"] / HBM_PEAK_GBS
        return cold

    def collective_check(self):
        """sharded runs: the timed path's global {Ψ, acc} against a plain RCCL all-reduce of the local ones"""
        self.step()
        got = self.out_t.clone()
        if self.fused_peer:
            self.be.ctx.set_peers([], 0, 0, 0)           # a LOCAL sweep for the reference
        self.be.ctx.sweep_dev(self.v_t.data_ptr(), self.out_t.data_ptr(), self.materialize)
        ref = self.out_t.clone()
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
        return f"pools x{self.world}, RCCL all-reduce of n_tokens+1 f64 per step"

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

    sr, err, out, psi, v_star = None, None, None, None, None
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
            out = {"ms": 1e3 * float(tmax.item()), "evaluations": sr.info.get("funcalls"),
                   "pools_total": sb.world * sb.m_rank, "ranks_agree_on_v": bool(torch.equal(vmax, vmin)),
                   "max_netflow": float(np.max(np.abs(psi))),
                   "pre_armed": bool(in_lib and sr._backend.ctx.get_option("armed")),
                   "collective": ("peer all-reduce inside the library (cfmm_set_peers), route! = one call per rank"
                                  if in_lib else "rccl via torch.distributed")}
    if out is None:
        out = {"error": err or "another rank failed"}
    return out, psi, v_star, sr


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="config3", choices=sorted(WORKLOADS))
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = one config-sized shard per GPU (default); strong = the config's pools divided among the GPUs")
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline / parity / route / host-boundary legs")
    ap.add_argument("--no-strong", action="store_true", help="N > 1: skip the additional strong-scaling leg (config 4)")
    ap.add_argument("--fused", action="store_true", help="time the fused evaluation (no Δ/Λ write-back)")
    ap.add_argument("--opt", action="append", default=[], help="library option key=value")
    ap.add_argument("--no-cold", action="store_true", help="skip the cache-cold pass")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="do not measure roofline.traffic with rocprofv3 PMC passes of this command (two bounded child runs); "
                         "use the committed profiles/traffic.json")
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

    # pass 1 -- THE timed region: K steps between barrier+synchronize, nothing else on the stream or the host
    elapsed, _ = sb.timed_pass(args.steps)
    elapsed = sb.max_over_ranks(elapsed)
    # pass 2 -- kernel events (roofline), out of the timed region
    kt, elapsed2 = sb.kernel_pass(args.steps)
    sweep_ms = sb.max_over_ranks(kt["sweep_ms"] / max(args.steps, 1))      # all sweep launches of one step, slowest rank
    reduce_ms = kt["reduce_ms"] / max(args.steps, 1)

    psi_timed = None if sb.ring is not None else sb.out_t.cpu().numpy().copy()   # what the timed path left behind (global at N > 1)
    cold = None
    if not args.no_cold and not args.cold_only:
        cold = sb.cold_pass(args.steps)
    collective_check = sb.collective_check() if use_dist else None
    if psi_timed is None:
        be.ctx.sweep_dev(sb.v_t.data_ptr(), sb.out_t.data_ptr(), materialize)
        torch.cuda.synchronize()
        psi_timed = sb.out_t.cpu().numpy().copy()

    route_sharded, psi_sharded, v_sharded, sr = None, None, None, None
    if use_dist:
        route_sharded, psi_sharded, v_sharded, sr = sharded_route(sb, local_rank)

    # the synchronous host-pointer boundary (what a ccall from Julia pays per evaluation):
    # pageable v in, Ψ/acc out over PCIe -- never the headline value
    host = {}
    if world == 1 and not use_dist and not args.no_cpu:
        be.ctx.reset_stream()
        v = sb.v

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

    ms_per_step = 1e3 * elapsed / args.steps
    value = world * sb.m_rank * args.steps / elapsed
    bytes_per_launch = alg_bytes(sb.batches, materialize, sb.v)
    achieved = bytes_per_launch / (sweep_ms * 1e-3) / 1e9 if sweep_ms > 0 else 0.0
    traffic, traffic_src, traffic_detail = None, None, None
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    if world == 1 and not use_dist and not args.no_cpu and not args.no_live_traffic:
        traffic, traffic_detail = live_traffic(args.workload, args.fused, args.opt)
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

    # Roofline of the dominant kernel (the sweep launch).  Headline = pool state resident in HBM (the cold
    # pass, or the whole timed region with --cold-only): every working set here fits the 256 MB Infinity
    # Cache, so the warm figure (same market every step, as inside route!) is a cache number and is reported
    # beside it, never as `frac`.  step_frac prices the WHOLE step (sweep + fold + boundaries) the same way.
    warm = {"achieved": achieved, "frac": achieved / HBM_PEAK_GBS, "kernel_ms": sweep_ms}
    if args.cold_only and sb.ring is not None:
        hbm, resid = dict(warm), "hbm-resident: the timed steps rotate over > 300 MB of market copies (--cold-only)"
        warm = None
    elif cold is not None:
        hbm = {"achieved": cold["achieved"], "frac": cold["frac"], "kernel_ms": cold["kernel_ms"]}
        resid = ("hbm-resident: cold pass over %d market copies (%.0f MB rotated) after the timed region; the timed "
                 "region itself sweeps one cache-resident market (see `warm`)" % (cold["copies"], cold["bytes_rotated"] / 1e6))
    else:
        hbm, resid = dict(warm), "cache-warm only (no cold pass in this run: --no-cold)"
    # `achieved` prices the REFERENCE's per-pool bytes (pool state + 32 B of Δ/Λ rows).  With the packed pool
    # records and the compact trade records the launch moves fewer bytes than that, so the rate over the bytes this layout
    # really has to move is reported beside it (the smaller number: read THAT one as "fraction of the bus").
    compact = bool(materialize and be.ctx.get_option("compact_trades"))
    packed = bool(be.ctx.get_option("pack")) and n <= 8192
    m_all = sum(len(b) for b in sb.batches)
    moved = bytes_per_launch - (16 * m_all if compact else 0) - (8 * m_all if packed else 0)
    layout = {"compact_trades": compact, "packed_records": packed, "bytes_per_launch": moved,
              "achieved": moved / (hbm["kernel_ms"] * 1e-3) / 1e9 if hbm["kernel_ms"] > 0 else 0.0}
    layout["frac"] = layout["achieved"] / HBM_PEAK_GBS
    roofline = {"bound": "hbm", "achieved": hbm["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": hbm["frac"],
                "traffic": traffic, "traffic_source": traffic_src, "traffic_detail": traffic_detail,
                "kernel": "cfmm::sweep_multi / cfmm::sweep_kernel (the sweep launch of one step)",
                "alg_bytes_per_launch": bytes_per_launch, "kernel_ms": hbm["kernel_ms"], "residency": resid,
                "layout": layout, "warm": warm, "cold": cold,
                "step_frac": bytes_per_launch / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "reduce_kernel_ms": reduce_ms, "ms_per_step_with_kernel_events": 1e3 * elapsed2 / args.steps,
                "how": "kernel_ms = mean duration of the sweep launches (slowest rank), from hipEvent pairs written by the "
                       "command processor at each kernel's start and stop (hipExtLaunchKernel) on the launch stream; compare "
                       "profiles/r03_*_kernel_stats.csv (warm: --no-cold runs; hbm-resident: --cold-only runs)"}

    line = {
        "metric": METRIC, "value": value, "unit": "pools/s",
        "value_is": "find_arb! pool-evaluations per second (materialising sweep + Ψ/dual reduction); "
                    "route! wall-clock is reported under `route`", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {sb.desc}", "pools_per_gpu": sb.m_rank, "pools_total": world * sb.m_rank,
                   "n_tokens": n, "variant": "materialising" if materialize else "fused", "segments": be.ctx.segments(),
                   "sharding": sb.sharding_text(), **({"rehearsal": "ranks share a GPU over gloo: not a measurement"} if share else {})},
        "roofline": roofline,
        "library_options": {k: be.ctx.get_option(k) for k in ("pack", "compact_trades", "alternate", "fast_math", "armed",
                                                               "stop_in_noise", "host_flag", "zero_copy")},
    }
    if collective_check is not None:
        line["collective_check_rel_err"] = collective_check
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
                                      "scaling": "strong", "sharding": s4.sharding_text(), "collective_check_rel_err": chk}
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
            route_gpu = {"_psi_sharded": psi_sharded, "_v_sharded": v_sharded}
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
    if rank == 0:
        print(json.dumps(line))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
