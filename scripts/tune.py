"""Launch-geometry sweep for the find_arb! kernels (run on the GPU box): prints event-timed kernel
durations for block / unroll / max_grid / bin_copies combinations."""
import itertools
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import cfmmrouter_amd as cr
from cfmmrouter_amd import synth

what = sys.argv[1] if len(sys.argv) > 1 else "product"
m = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
n = int(sys.argv[3]) if len(sys.argv) > 3 else 256
gen = {"product": synth.product_pools, "geomean": synth.geomean_pools, "bounded": synth.bounded_product_pools}[what]
b = gen(m, n, seed=1234)
v = synth.sweep_prices(n, seed=1234)
be = cr.DeviceBackend(n, [b])
stream = torch.cuda.Stream()
torch.cuda.set_stream(stream)
be.ctx.set_stream(stream.cuda_stream)
v_t = torch.from_numpy(v).to("cuda")
out_t = torch.zeros(n + 1, dtype=torch.float64, device="cuda")


def run(mat, steps=30, **opts):
    for k, val in opts.items():
        be.ctx.set_option(k, val)
    be.ctx.set_option("time_kernels", 0)
    for _ in range(5):
        be.ctx.sweep_dev(v_t.data_ptr(), out_t.data_ptr(), mat)
    torch.cuda.synchronize()
    be.ctx.set_option("time_kernels", 1)
    be.ctx.kernel_times()
    for _ in range(steps):
        be.ctx.sweep_dev(v_t.data_ptr(), out_t.data_ptr(), mat)
    kt = be.ctx.kernel_times()
    return 1e3 * kt["sweep_ms"] / steps, 1e3 * kt["reduce_ms"] / steps, be.ctx.segments()[0]


print(f"# {what} m={m} n={n}  (hipExtLaunchKernel start/stop events)")
print("block unroll max_grid copies nt | sweep_us(mat) sweep_us(fused) reduce_us grid")
for block, unroll, mg, cp, nt in itertools.product([256, 512, 1024], [1, 2, 4], [256, 512, 1024, 2048], [1, 2], [0, 1]):
    if block >= 512 and mg > 1024:
        continue
    if block == 256 and mg < 1024:
        continue
    if nt and cp == 1:
        continue
    try:
        a, r, seg = run(True, block=block, unroll=unroll, max_grid=mg, bin_copies=cp, nt_stores=nt)
        f, _, _ = run(False, block=block, unroll=unroll, max_grid=mg, bin_copies=cp, nt_stores=nt)
        print(f"{block:5d} {unroll:6d} {mg:8d} {cp:6d} {nt:2d} | {a:9.2f} {f:9.2f} {r:9.2f} {seg['grid']}", flush=True)
    except Exception as e:
        print(block, unroll, mg, cp, nt, "ERR", e, flush=True)
be.close()
