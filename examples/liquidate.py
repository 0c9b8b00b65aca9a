"""Liquidating a basket of tokens -- the reference's examples/liquidate.jl on the MI355X path."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import cfmmrouter_amd as cr


def main(solver="scipy"):
    cfmms = [
        cr.ProductTwoCoin([1e3, 1e4], 0.997, [1, 2]),
        cr.ProductTwoCoin([1e3, 1e2], 0.997, [2, 3]),
        cr.ProductTwoCoin([1e3, 2e4], 0.997, [1, 3]),
    ]
    n = max(max(c.Ai) for c in cfmms)
    results = []
    # liquidate a basket of tokens 2 & 3 into token 1, then the special case token 1 -> token 2
    for i, Δin in ((1, [0, 1e1, 1e2]), (2, [1e1, 0.0, 0.0])):
        router = cr.Router(cr.BasketLiquidation(i, Δin), cfmms, n)
        cr.route_(router, solver=solver)
        Ψ = cr.netflows(router)
        print(f"Input Basket: {np.round(Δin).astype(int)}")
        print(f"Net trade: {np.round(Ψ).astype(int)}")
        print(f"Amount received: {round(Ψ[i - 1])}")
        for k, (Δ, Λ) in enumerate(zip(router.Δs, router.Λs)):
            print(f"CFMM {k + 1}: tendered {np.round(Δ, 3)}  received {np.round(Λ, 3)}")
        results.append((Ψ.copy(), router.v.copy()))
        router.close()
    return results


if __name__ == "__main__":
    main(*sys.argv[1:])
