"""Lane-per-pool vs wavefront-cooperative UniV3 walks on deep tick ladders (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import cfmmrouter_amd as cr
from cfmmrouter_amd import synth

for m, t, spread, stragglers in ((100_000, 200, 8.0, False), (100_000, 200, 0.5, False), (100_000, 200, 0.02, True),
                                 (100_000, 32, 8.0, False), (1_000_000, 2, 0.2, False)):
    n = 64
    b = synth.univ3_pools(m, n, t, seed=1) if t > 2 else synth.bounded_product_pools(m, n, seed=1)
    v = synth.sweep_prices(n, seed=1, spread=spread)
    if stragglers:            # one token far off: ~3 % of the pools walk their whole ladder, the rest stay put
        v[0] *= np.exp(8.0)
    be = cr.DeviceBackend(n, [b])
    stream = torch.cuda.Stream(); torch.cuda.set_stream(stream); be.ctx.set_stream(stream.cuda_stream)
    v_t = torch.from_numpy(v).to("cuda"); out_t = torch.zeros(n + 1, dtype=torch.float64, device="cuda")
    res = {}
    for mode in (0, 1):
        be.ctx.set_option("univ3_coop", mode)
        for _ in range(3):
            be.ctx.sweep_dev(v_t.data_ptr(), out_t.data_ptr(), True)
        torch.cuda.synchronize()
        be.ctx.set_option("time_kernels", 1); be.ctx.kernel_times()
        for _ in range(20):
            be.ctx.sweep_dev(v_t.data_ptr(), out_t.data_ptr(), True)
        kt = be.ctx.kernel_times(); be.ctx.set_option("time_kernels", 0)
        res[mode] = (1e3 * kt["sweep_ms"] / 20, out_t.cpu().numpy().copy())
    same = np.array_equal(res[0][1], res[1][1])
    print(f"m={m} ticks={t} spread={spread} stragglers={stragglers}: lane-per-pool {res[0][0]:.1f} us, wave-cooperative {res[1][0]:.1f} us, identical Psi: {same}")
    be.close()
