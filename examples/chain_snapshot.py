"""A market taken from a snapshot of on-chain pool state (raw integer reserves, decimals, sqrtPriceX96 + initialized
ticks): cfmmrouter_amd.chain converts it into the reference's pool types, route! finds the arbitrage.

The reference has no loader -- its examples build pools by hand (examples/arbitrage.jl:8-17); this is the data
format on the caller's side of the path."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import cfmmrouter_amd as cr
from cfmmrouter_amd import chain

HERE = os.path.dirname(os.path.abspath(__file__))


def main(solver="native", path=os.path.join(HERE, "data", "snapshot.jsonl")):
    tokens, batches = chain.load_snapshot(path)
    n = len(tokens)
    usd = {"USDC": 1.0, "DAI": 1.0, "USDT": 1.0, "FRAX": 0.998, "LUSD": 1.004}
    c = np.array([usd[t] for t in tokens])                      # value the output in USD: LinearNonnegative(c)
    router = cr.Router(cr.LinearNonnegative(c), batches, n)
    cr.route_(router, v=c.copy(), solver=solver)
    Ψ = cr.netflows(router)
    print(f"{sum(len(b) for b in batches)} pools, {n} tokens")
    for t, x in zip(tokens, Ψ):
        print(f"  {t:5s} net {x:+.6f}")
    print(f"Profit: {float(c @ Ψ):.2f} USD")
    out = (tokens, Ψ.copy(), router.v.copy(), batches, np.array(router.Δs), np.array(router.Λs))
    router.close()
    return out


if __name__ == "__main__":
    main()
