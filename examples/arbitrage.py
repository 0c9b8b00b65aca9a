"""Arbitrage -- the reference's examples/arbitrage.jl on the MI355X path (same pools, same calls).

Three pools of the same two tokens, no fees (γ = 1): two constant-product pools and one weighted
(geometric-mean) pool; LinearNonnegative objective with unit prices."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import cfmmrouter_amd as cr


def main(solver="scipy"):
    equal_pool = cr.ProductTwoCoin([1e6, 1e6], 1, [1, 2])
    unequal_small_pool = cr.ProductTwoCoin([1e3, 2e3], 1, [1, 2])
    weighted_pool = cr.GeometricMeanTwoCoin([1e4, 2e4], [.4, .6], 1, [1, 2])

    prices = np.ones(2)
    router = cr.Router(cr.LinearNonnegative(prices), [equal_pool, unequal_small_pool, weighted_pool], 2)
    cr.route_(router, solver=solver)

    Ψ = np.round(cr.netflows(router)).astype(int)
    print(f"Net trade: {Ψ}")
    print(f"Profit: {prices @ Ψ}")
    eps = np.finfo(float).eps
    for i, (Δ, Λ) in enumerate(zip(router.Δs, router.Λs)):
        tokens = router.cfmms[i].Ai
        print(f"CFMM {i + 1}:")
        print("\tTendered basket:", ", ".join(f"{tokens[k]}: {round(δ)}" for k, δ in enumerate(Δ) if δ > eps))
        print("\tReceived basket:", ", ".join(f"{tokens[k]}: {round(λ)}" for k, λ in enumerate(Λ) if λ > eps))
    out = cr.netflows(router).copy(), router.v.copy(), router.Δs.copy(), router.Λs.copy()
    router.close()
    return out


if __name__ == "__main__":
    main(*sys.argv[1:])
