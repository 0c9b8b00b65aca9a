"""The ProductTwoCoin sweep against the roofline as the market grows (HBM-resident throughout: the sweeps rotate over market
copies whose TOUCHED bytes are >= 2 x the 256 MiB Infinity Cache, benchlib.workloads.ring_copies) -- where does the launch
floor stop mattering?  python scripts/size_scaling.py [m ...]   (256 tokens, materialising sweep, kernel span by CP events)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import numpy as np, torch
import cfmmrouter_amd as cr
from cfmmrouter_amd import synth
from benchlib import workloads as W

n = 256
sizes = [int(a) for a in sys.argv[1:]] or [250_000, 500_000, 1_000_000, 2_000_000, 4_000_000, 8_000_000, 16_000_000, 32_000_000]
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
v_t = torch.from_numpy(synth.sweep_prices(n, seed=1234)).to("cuda")
out_t = torch.zeros(n + 1, dtype=torch.float64, device="cuda")
print("# ProductTwoCoin, 256 tokens, materialising sweep, HBM-resident (ring of market copies: touched bytes >= 2 x 256 MiB)")
print("#        pools  copies   sweep us   step us   pool-evals/s   frac (64 B/pool ref. layout)   bus (40 B/pool moved) of 8 TB/s")
for m in sizes:
    batch = [synth.product_pools(m, n, seed=1234)]
    per_copy = W.touched_bytes(batch, True)
    copies = W.ring_copies(per_copy)
    ring = [cr.DeviceBackend(n, batch) for _ in range(copies)]
    for b in ring:
        b.ctx.set_stream(stream.cuda_stream)
        for kv in filter(None, os.environ.get("OPTS", "").split(",")):      # OPTS="max_grid=512,block=1024": library options
            k_, val = kv.split("=")
            b.ctx.set_option(k_, int(val))
    K = max(3 * copies, 24)
    for k in range(2 * copies):
        ring[k % copies].ctx.sweep_dev(v_t.data_ptr(), out_t.data_ptr(), True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        ring[k % copies].ctx.sweep_dev(v_t.data_ptr(), out_t.data_ptr(), True)
    torch.cuda.synchronize()
    step = 1e6 * (time.perf_counter() - t0) / K
    for b in ring:
        b.ctx.set_option("time_kernels", 1); b.ctx.kernel_times()
    for k in range(K):
        ring[k % copies].ctx.sweep_dev(v_t.data_ptr(), out_t.data_ptr(), True)
    torch.cuda.synchronize()
    sw = 1e3 * sum(b.ctx.kernel_times()["sweep_ms"] for b in ring) / K
    print(f"{m:14d} {copies:7d} {sw:10.2f} {step:9.2f} {m / (step * 1e-6):14.3e} {64.0 * m / (sw * 1e-6) / 8e12:18.3f} {40.0 * m / (sw * 1e-6) / 8e12:28.3f}", flush=True)
    for b in ring:
        b.close()
    del ring, batch
