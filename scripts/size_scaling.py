"""Sweep duration vs market size (ProductTwoCoin, 256 tokens, materialising): how much of the
distance to the HBM roofline is fixed ramp/tail cost of a ~10 us kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cfmmrouter_amd as cr
from cfmmrouter_amd import synth

n = 256
v = synth.sweep_prices(n, seed=1234)
for m in (250_000, 500_000, 1_000_000, 2_000_000, 4_000_000, 8_000_000):
    b = synth.product_pools(m, n, seed=1234)
    be = cr.DeviceBackend(n, [b])
    stream = torch.cuda.Stream(); torch.cuda.set_stream(stream); be.ctx.set_stream(stream.cuda_stream)
    v_t = torch.from_numpy(v).to("cuda"); out_t = torch.zeros(n + 1, dtype=torch.float64, device="cuda")
    for _ in range(5):
        be.ctx.sweep_dev(v_t.data_ptr(), out_t.data_ptr(), True)
    torch.cuda.synchronize()
    be.ctx.set_option("time_kernels", 1); be.ctx.kernel_times()
    for _ in range(30):
        be.ctx.sweep_dev(v_t.data_ptr(), out_t.data_ptr(), True)
    kt = be.ctx.kernel_times()
    us = 1e3 * kt["sweep_ms"] / 30
    print(f"m={m:>9d}  sweep {us:7.2f} us (hipEvents)  {64 * m / us / 1e6:6.2f} TB/s algorithmic  frac {64 * m / us / 1e6 / 8:.3f}", flush=True)
    be.close()
