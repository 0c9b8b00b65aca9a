"""Regenerates tests/golden/golden.npz from the CPU oracle (oracle/cfmm_oracle.c).

The reference is Julia and cannot run in this image, so these vectors are NOT reference output:
they freeze the oracle (which is pinned on the reference's KATs, tests/test_oracle_kat.py) so that
neither the oracle nor the HIP path can drift silently.  Inputs come from the documented
counter-mode generator (cfmmrouter.jl_amd/synth.py), so a Julia script can rebuild identical pools.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import cfmmrouter_amd as cr
from cfmmrouter_amd import synth
from oracle import cfmm_oracle as orc
from helpers import oracle_objective, oracle_poolset, oracle_sweep

out = {}

# fixed-v sweeps, one per family
n = 7
bp = synth.product_pools(257, n, seed=11)
v = synth.sweep_prices(n, seed=11, spread=0.5)
D, L, psi, acc = oracle_sweep([bp], n, v)
out.update(prod_v=v, prod_D=D, prod_L=L, prod_psi=psi, prod_acc=acc)

n = 5
bg = synth.geomean_pools(129, n, seed=12)
v = synth.sweep_prices(n, seed=12, spread=0.5)
D, L, psi, acc = oracle_sweep([bg], n, v)
out.update(geo_v=v, geo_D=D, geo_L=L, geo_psi=psi, geo_acc=acc)

n = 9
bu = synth.univ3_pools(64, n, 7, seed=13)
v = synth.sweep_prices(n, seed=13, spread=1.5)
D, L, psi, acc = oracle_sweep([bu], n, v)
out.update(uni_v=v, uni_D=D, uni_L=L, uni_psi=psi, uni_acc=acc)

# the reference's UniV3 fixture (test/cfmms.jl:117-120) at its 7+7 scenarios (:127-199)
rows = []
for g in (1.0, 0.997):
    pool = orc.UniV3(15.0, [30.0, 20, 10, 5], [1.0, 2.0, 1.5, 0.0], g)
    for p in (15.0 if g == 1.0 else 15.0 * (1 + 0.997) / 2, 16.0, 14.0, 25.0, 7.5, 4.0, 35.0):
        Dd, Ll = pool.find_arb([p, 1.0])
        rows.append([g, p, Dd[0], Dd[1], Ll[0], Ll[1]])
out["uni_fixture"] = np.array(rows)

# route!-level results of the CPU restatement (SciPy L-BFGS-B): v*, Ψ*
for name, obj, m, n, v0 in (
        ("arb", cr.LinearNonnegative(synth.linear_prices(10, seed=21)), 100, 10, np.ones(10)),
        ("basket", cr.BasketLiquidation(1, synth.basket(10, seed=22)), 100, 10, None)):
    b = synth.product_pools(m, n, seed=21)
    ref = orc.route_oracle(oracle_objective(obj), oracle_poolset([b], n), v0=v0)
    out[f"route_{name}_v"] = ref["v"]
    out[f"route_{name}_psi"] = ref["psi"]

np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden.npz"), **out)
print("wrote golden.npz with", sorted(out))
