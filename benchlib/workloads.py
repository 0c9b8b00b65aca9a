"""The bench's workloads (BASELINE.json configs on synthetic markets) and their byte accounting."""
import numpy as np

import cfmmrouter_amd as cr
from cfmmrouter_amd import synth
from cfmmrouter_amd._lib import KIND_GEOMEAN, KIND_PRODUCT, KIND_UNIV3

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
METRIC = "find_arb! pools/sec + route! wall-clock, 1M-pool arbitrage, 1/2/4/8 GPU"

# algorithmic bytes per pool-evaluation, materialising sweep (SURVEY.md §8d / DESIGN.md):
#   read pool state + write Δ(16 B) + Λ(16 B)
ALG_BYTES = {KIND_PRODUCT: 32 + 32, KIND_GEOMEAN: 48 + 32}
ALG_BYTES_FUSED = {KIND_PRODUCT: 32, KIND_GEOMEAN: 48}


def alg_bytes(batches, materialize=True, v=None):
    """SURVEY §8d bytes of one launch.  UniV3: 32 B header + 16 B per tick (+ 32 B of trades); multi-tick ladders
    (`v` given): per tick VISITED by the walk at these prices, not per tick stored."""
    tot = 0
    for b in batches:
        if b.kind == KIND_UNIV3:
            ticks = b.lower_ticks.size
            if v is not None and ticks > 2 * len(b):
                ticks = int(np.sum(np.maximum(synth.univ3_ticks_visited(b, v), 1)))   # an idle pool still reads its current tick
            tot += len(b) * 32 + 16 * ticks + (32 * len(b) if materialize else 0)
        else:
            tot += len(b) * (ALG_BYTES if materialize else ALG_BYTES_FUSED)[b.kind]
    return tot


L3_BYTES = 256 * 2 ** 20      # Infinity Cache (MI355X_MICROARCH.md, memory hierarchy)


def touched_bytes(batches, materialize=True):
    """Bytes ONE sweep moves over the L2 <-> fabric boundary in THIS library's packed layout, by construction: ProductTwoCoin
    24 B read + 16 B trade record, GeometricMean 48 + 16, UniV3 56 (+ 16 B walk spans on multi-tick segments) + 16 -- a LOWER
    bound (the scattered walk records of multi-tick ladders are not counted).  Sizes the HBM-resident ring."""
    mat = 16 if materialize else 0
    tot = 0
    for b in batches:
        if b.kind == KIND_PRODUCT:
            tot += len(b) * (24 + mat)
        elif b.kind == KIND_GEOMEAN:
            tot += len(b) * (48 + mat)
        else:
            tot += len(b) * (56 + (16 if b.lower_ticks.size > 2 * len(b) else 0) + mat)
    return int(tot)


def ring_copies(per_copy):
    """copies of a market such that the ring's touched bytes are >= 2 x the Infinity Cache"""
    return int(np.ceil(2 * L3_BYTES / per_copy)) + 1


# name: (description, n_tokens, [(generator, pools per GPU (weak) = pools in total (strong), kwargs)])
WORKLOADS = {
    "config2": ("100k ProductTwoCoin pools, 64 tokens, LinearNonnegative arbitrage", 64,
                [(synth.product_pools, 100_000, {})]),
    "config3": ("1M mixed ProductTwoCoin + GeometricMeanTwoCoin pools (500k each), 256 tokens, "
                "LinearNonnegative arbitrage", 256,
                [(synth.product_pools, 500_000, {}), (synth.geomean_pools, 500_000, {})]),
    "config4shard": ("500k ProductTwoCoin pools per GPU (4M over 8 GPUs), 512 tokens", 512,
                     [(synth.product_pools, 500_000, {})]),
    "config4": ("4M ProductTwoCoin pools in total, 512 tokens (BASELINE config 4; --scaling strong: 4M / N per GPU)", 512,
                [(synth.product_pools, 4_000_000, {})]),
    "config5": ("1M BoundedProduct (2-tick UniV3) pools quoted around one token price vector (1 % noise), 256 tokens, "
                "BasketLiquidation (interior dual optimum)", 256,
                [(synth.bounded_product_pools, 1_000_000, {"consistent": True})]),
    "config5corner": ("1M BoundedProduct pools with independent random prices (arbitrage-rich: route! ends at the box "
                      "corner after 2 evaluations), 256 tokens, BasketLiquidation", 256,
                      [(synth.bounded_product_pools, 1_000_000, {})]),
    "univ3_ticks": ("1M UniV3 pools with ragged ladders of 2..64 initialised ticks (17 on average) quoted around one token "
                    "price vector, 256 tokens; at the sweep's prices 3/4 of the pools walk through more than one tick", 256,
                    [(synth.univ3_ragged_pools, 1_000_000, {})]),
    "large_n": ("1M ProductTwoCoin pools, 65536 tokens (global-bin path), LinearNonnegative arbitrage", 65536,
                [(synth.product_pools, 1_000_000, {})]),
    "product1m": ("1M ProductTwoCoin pools, 256 tokens, LinearNonnegative arbitrage", 256,
                  [(synth.product_pools, 1_000_000, {})]),
}


def shard_range(m, rank, world):
    base, rem = divmod(int(m), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def build_market(name, rank, world, scaling):
    """This rank's shard of the workload: weak = one config-sized shard per rank (pool indices [rank*m, (rank+1)*m) of
    the generator's stream), strong = a contiguous 1/world of the config's pools.  The generators are pure functions of
    (seed, pool index), so the shards of all ranks concatenated ARE the global market (`build_global`)."""
    _, n, parts = WORKLOADS[name]
    out = []
    for gen, m, kw in parts:
        if scaling == "strong":
            lo, hi = shard_range(m, rank, world)
        else:
            lo, hi = rank * m, (rank + 1) * m
        out.append(gen(hi - lo, n, seed=1234, first=lo, **kw))
    return out


def build_global(name, world, scaling):
    _, n, parts = WORKLOADS[name]
    return [gen(m if scaling == "strong" else world * m, n, seed=1234, first=0, **kw) for gen, m, kw in parts]


def sweep_prices_for(name, n):
    v = synth.sweep_prices(n, seed=1234)
    if name == "univ3_ticks":     # the ladders are quoted around the token price vector: sweep a few per cent off it
        v = v * synth.token_price_vector(n, seed=1234)
    return v


def objective_for(name, n):
    if name.startswith("config5") or name == "univ3_ticks":
        return cr.BasketLiquidation(1, synth.basket(n, seed=1234))
    return cr.LinearNonnegative(synth.linear_prices(n, seed=1234))


