"""Would ordering the pools of a multi-tick UniV3 segment by ladder length pay?  (suggestion (a).)  The same 1M
ragged pools uploaded in generator order, sorted by tick count, and sorted by (tick count, current-tick position); sweep kernel
time warm and HBM-resident.  python scripts/univ3_sort_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import numpy as np, torch
import cfmmrouter_amd as cr
from cfmmrouter_amd import synth
from cfmmrouter_amd.cfmms import PoolBatch
from benchlib import workloads as W

n = 256
b = W.build_market("univ3_ticks", 0, 1, "weak")[0]
v = W.sweep_prices_for("univ3_ticks", n)


def permuted(batch, idx):
    lens = np.diff(batch.tick_off)[idx]
    off = np.zeros(len(idx) + 1, dtype=np.int64)
    np.cumsum(lens, out=off[1:])
    src = np.repeat(batch.tick_off[:-1][idx] - off[:-1], lens) + np.arange(off[-1])
    return PoolBatch(batch.kind, current_price=batch.current_price[idx], tick_off=off, lower_ticks=batch.lower_ticks[src],
                     liquidity=batch.liquidity[src], γ=batch.γ[idx], Ai=batch.Ai[idx])


lens = np.diff(b.tick_off)
cur = np.array([np.count_nonzero(b.lower_ticks[b.tick_off[i]:b.tick_off[i + 1]] >= b.current_price[i]) for i in range(0, len(b), 1)]) if len(b) <= 0 else None
orders = {"generator order": np.arange(len(b)), "sorted by tick count": np.argsort(lens, kind="stable")}
stream = torch.cuda.Stream(); torch.cuda.set_stream(stream)
v_t = torch.from_numpy(v).to("cuda"); out_t = torch.zeros(n + 1, dtype=torch.float64, device="cuda")
for name, idx in orders.items():
    pb = permuted(b, idx)
    for cold in (0, 1):
        copies = W.ring_copies(int(0.7 * 246e6)) if cold else 1
        ring = [cr.DeviceBackend(n, [pb]) for _ in range(copies)]
        for be in ring:
            be.ctx.set_stream(stream.cuda_stream)
        K = 60
        for k in range(2 * copies):
            ring[k % copies].ctx.sweep_dev(v_t.data_ptr(), out_t.data_ptr(), True)
        torch.cuda.synchronize()
        for be in ring:
            be.ctx.set_option("time_kernels", 1); be.ctx.kernel_times()
        for k in range(K):
            ring[k % copies].ctx.sweep_dev(v_t.data_ptr(), out_t.data_ptr(), True)
        torch.cuda.synchronize()
        sw = 1e3 * sum(be.ctx.kernel_times()["sweep_ms"] for be in ring) / K
        psi = out_t.cpu().numpy()
        print(f"{name:24s} {'HBM-resident' if cold else 'cache-warm  '}  sweep kernel {sw:6.2f} us   sum psi {psi[:n].sum():.6e}", flush=True)
        for be in ring:
            be.close()
