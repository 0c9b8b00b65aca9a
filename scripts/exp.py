"""Option A/B runs on the GPU box: python scripts/exp.py WORKLOAD "k=v,k=v" "k=v" ...
Prints, per option set: wall µs per step (K back-to-back device-resident sweeps, stream-synced),
sweep-kernel and fold-kernel µs (hipExtLaunchKernel events), materialising and fused.
COLD=1: the sweeps rotate over market copies whose TOUCHED bytes are >= 2 x the 256 MiB Infinity Cache (pool state from HBM)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import numpy as np, torch
import cfmmrouter_amd as cr
from cfmmrouter_amd import synth
from benchlib import workloads as bench

name = sys.argv[1]
desc, n, _ = bench.WORKLOADS[name]
batches = bench.build_market(name, 0, 1, "weak")
v = bench.sweep_prices_for(name, n)
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
v_t = torch.from_numpy(v).to("cuda"); out_t = torch.zeros(n + 1, dtype=torch.float64, device="cuda")
K = int(os.environ.get("K", "200"))
print(f"# {name}: {desc}")
COLD = os.environ.get("COLD", "0") == "1"
per_copy = bench.touched_bytes(batches, True)     # the packed layout's own bytes (a lower bound): ring >= 2 x the Infinity Cache
copies = bench.ring_copies(per_copy) if COLD else 1
for spec in sys.argv[2:] or [""]:
    ring = [cr.DeviceBackend(n, batches) for _ in range(copies)]
    for b_ in ring:
        b_.ctx.set_stream(stream.cuda_stream)
        for kv in filter(None, spec.split(",")):
            k, val = kv.split("="); b_.ctx.set_option(k, int(val))
    be = ring[0]
    res = []
    if COLD:
        for mat in (True, False):
            for k in range(2 * copies): ring[k % copies].ctx.sweep_dev(v_t.data_ptr(), out_t.data_ptr(), mat)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(K): ring[k % copies].ctx.sweep_dev(v_t.data_ptr(), out_t.data_ptr(), mat)
            torch.cuda.synchronize()
            wall = 1e6 * (time.perf_counter() - t0) / K
            for b_ in ring: b_.ctx.set_option("time_kernels", 1); b_.ctx.kernel_times()
            for k in range(K): ring[k % copies].ctx.sweep_dev(v_t.data_ptr(), out_t.data_ptr(), mat)
            kts = [b_.ctx.kernel_times() for b_ in ring]
            for b_ in ring: b_.ctx.set_option("time_kernels", 0)
            res.append((wall, 1e3 * sum(k_["sweep_ms"] for k_ in kts) / K, 1e3 * sum(k_["reduce_ms"] for k_ in kts) / K))
        seg = [(s["block"], s["grid"]) for s in be.ctx.segments()]
        (w1, s1, r1), (w2, s2, r2) = res
        print(f"COLD {spec or '(default)':40s} mat: step {w1:6.2f} sweep {s1:6.2f} fold {r1:5.2f} | fused: step {w2:6.2f} sweep {s2:6.2f} fold {r2:5.2f}  {seg}", flush=True)
        for b_ in ring: b_.close()
        continue
    warm_s = float(os.environ.get("WARM_S", "0"))
    if warm_s > 0:   # sustained load first: lets the clock governor ramp (DVFS) before anything is timed
        t_end = time.perf_counter() + warm_s
        while time.perf_counter() < t_end:
            for _ in range(200): be.ctx.sweep_dev(v_t.data_ptr(), out_t.data_ptr(), True)
            torch.cuda.synchronize()
    for mat in (True, False):
        for _ in range(20): be.ctx.sweep_dev(v_t.data_ptr(), out_t.data_ptr(), mat)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(K): be.ctx.sweep_dev(v_t.data_ptr(), out_t.data_ptr(), mat)
        torch.cuda.synchronize()
        wall = 1e6 * (time.perf_counter() - t0) / K
        be.ctx.set_option("time_kernels", 1); be.ctx.kernel_times()
        for _ in range(K): be.ctx.sweep_dev(v_t.data_ptr(), out_t.data_ptr(), mat)
        kt = be.ctx.kernel_times(); be.ctx.set_option("time_kernels", 0)
        res.append((wall, 1e3 * kt["sweep_ms"] / K, 1e3 * kt["reduce_ms"] / K))
    seg = [(s["block"], s["grid"]) for s in be.ctx.segments()]
    (w1, s1, r1), (w2, s2, r2) = res
    for b_ in ring[1:]: b_.close()
    print(f"{spec or '(default)':45s} mat: step {w1:6.2f} sweep {s1:6.2f} fold {r1:5.2f} | fused: step {w2:6.2f} sweep {s2:6.2f} fold {r2:5.2f}  {seg}", flush=True)
    be.close()
