"""A/B of the trade-store policy (option stream_stores: 1 write-through, 2 non-temporal, 0 auto by size) on 8M / 16M ProductTwoCoin
pools, HBM-resident by size (ring of 3 / 2 market copies): sweep-kernel span by CP events, step wall clock, fraction of the bus over the
40 B per pool the layout moves.  profiles/r06_ab_stream_stores.txt.  usage: python scripts/atscale_stores.py"""
import sys, os, time; sys.path.insert(0, os.getcwd())
os.environ.setdefault("HIP_FORCE_DEV_KERNARG","1")
import numpy as np, torch, cfmmrouter_amd as cr
from cfmmrouter_amd import synth
n=256
for m in (8_000_000, 16_000_000):
    batch=[synth.product_pools(m,n,seed=1234)]
    for opt in (1,2,1,2,0):
        copies=3 if m==8_000_000 else 2
        ring=[cr.DeviceBackend(n,batch) for _ in range(copies)]
        st=torch.cuda.current_stream()
        v_t=torch.from_numpy(synth.sweep_prices(n,seed=1234)).cuda(); out=torch.zeros(n+1,dtype=torch.float64,device="cuda")
        for b in ring: b.ctx.set_stream(st.cuda_stream); b.ctx.set_option("stream_stores",opt)
        K=8*copies
        for k in range(2*copies): ring[k%copies].ctx.sweep_dev(v_t.data_ptr(),out.data_ptr(),True)
        torch.cuda.synchronize()
        for b in ring: b.ctx.set_option("time_kernels",1); b.ctx.kernel_times()
        t0=time.perf_counter()
        for k in range(K): ring[k%copies].ctx.sweep_dev(v_t.data_ptr(),out.data_ptr(),True)
        torch.cuda.synchronize(); wall=1e6*(time.perf_counter()-t0)/K
        sw=1e3*sum(b.ctx.kernel_times()["sweep_ms"] for b in ring)/K
        print(m, "stream_stores",opt, "kernel us %.1f step %.1f bus %.3f"%(sw,wall,40.0*m/(sw*1e-6)/8e12), flush=True)
        for b in ring: b.close()
