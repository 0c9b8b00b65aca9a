"""One-off robustness check for markets whose per-pool arrays exceed 2^31 bytes (64-bit indexing everywhere:
upload, sweep, compact trade records, download): M ProductTwoCoin pools (default 150M: R = 2.4 GB, trade records
2.4 GB), every row compared bit for bit with the CPU restatement (test infrastructure: scripts/ is not the
product path).  usage: python scripts/big_market.py [M] > profiles/rNN_big_market.txt"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import cfmmrouter_amd as cr
from cfmmrouter_amd import synth
from helpers import oracle_sweep

M = int(float(sys.argv[1])) if len(sys.argv) > 1 else 150_000_000
n = 256
t0 = time.perf_counter()
b = synth.product_pools(M, n, seed=99)
v = synth.sweep_prices(n, seed=98)
print(f"market: {M} ProductTwoCoin pools, {n} tokens; R = {b.R.nbytes / 2**30:.2f} GiB (built in {time.perf_counter() - t0:.1f} s)", flush=True)
be = cr.DeviceBackend(n, [b])
try:
    for rep in range(2):
        t0 = time.perf_counter()
        psi, acc = be.find_arb(v)
        t1 = time.perf_counter() - t0
    print(f"host-pointer find_arb!: {1e3 * t1:.2f} ms = {M / t1:.3e} pools/s (PCIe-inclusive call, pool state from HBM)", flush=True)
    t0 = time.perf_counter()
    D, L = be.trades()
    print(f"trades downloaded and decoded in {time.perf_counter() - t0:.1f} s", flush=True)
    t0 = time.perf_counter()
    Do, Lo, psio, acco = oracle_sweep([b], n, v, nthreads=max(1, (os.cpu_count() or 2) // 2))
    print(f"oracle sweep + reductions in {time.perf_counter() - t0:.1f} s", flush=True)
    same_D, same_L = bool(np.array_equal(D, Do)), bool(np.array_equal(L, Lo))
    print("Delta rows bit-equal:", same_D, " Lambda rows bit-equal:", same_L)
    print("last rows:", D[-1], L[-1], "| oracle:", Do[-1], Lo[-1])
    print("netflow rel err:", float(np.max(np.abs(psi - psio)) / np.max(np.abs(psio))), " dual rel err:", abs(acc - acco) / abs(acco))
    assert same_D and same_L
    print("OK")
finally:
    be.close()
