"""Sequential routing on a sharded router -- what the reference's docs describe (docs/src/method.md:81:
route, apply the trades to the pools, route again) but cannot run, because its `update_reserves!`
(src/router.jl:127-132) calls a per-pool method that does not exist.

A 200k-pool market (ProductTwoCoin + GeometricMeanTwoCoin) is split over the GPUs given on the command
line -- from this ONE process, through the C ABI's multi-device context (cfmm_ctx_create_multi; an ordinal
may repeat, `0,0` = two shards on one GPU) -- arbitraged, updated in place on the devices, and arbitraged
again: the second pass finds (almost) nothing.

    python examples/sequential_sharded.py [device,device,...]
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

import cfmmrouter_amd as cr
from cfmmrouter_amd import synth


def main(devices="0,0"):
    devs = [int(d) for d in str(devices).split(",")]
    n = 64
    market = [synth.product_pools(150_000, n, seed=1), synth.geomean_pools(50_000, n, seed=2)]
    prices = synth.linear_prices(n, seed=3)
    router = cr.Router(cr.LinearNonnegative(prices), market, n, device=devs if len(devs) > 1 else devs[0])
    profits = []
    for round_ in (1, 2):
        cr.route_(router, v=np.ones(n) if round_ == 1 else router.v.copy(), solver="native")
        Ψ = cr.netflows(router)
        profits.append(float(prices @ Ψ))
        traded = int(np.count_nonzero(router.Δs.sum(axis=1) > 0))
        print(f"round {round_}: profit {profits[-1]:.6g} over {traded} trading pools "
              f"({router.info['funcalls']} evaluations on {len(devs)} shard(s))")
        cr.update_reserves_(router, sync_host=False)      # R <- R + γΔ − Λ on the devices, no per-pool host traffic
    router.close()
    return profits


if __name__ == "__main__":
    main(*sys.argv[1:])
