"""Synthetic markets with the reference benchmark's distributions (benchmark/scaling.jl:13-34).

Julia's `rand` stream cannot be reproduced outside Julia, so "identical pools" come from a
generator this repo owns: SplitMix64 in counter mode, u(i) = mix(seed*GOLD + i*GOLD2) mapped to
[0,1) by taking the top 53 bits -- a pure function of (seed, stream, i), so any implementation
(the bench/reference.jl script included) regenerates the same bits in any order.

    R        = 1000 * U[0,1)^2              (scaling.jl:24)
    γ        ∈ {0.997, 1.0} equiprobable    (scaling.jl:25)
    Ai       = two distinct tokens, uniform (scaling.jl:22, sample(..., replace=false))
    c        = U[0,1)^n, floored at 2^-53   (scaling.jl:31; LinearNonnegative needs c > 0)
    w₁       = U(0,1), w₂ = 1 − w₁          (test/cfmms.jl:101)
"""
from __future__ import annotations

import numpy as np

from ._lib import KIND_GEOMEAN, KIND_PRODUCT, KIND_UNIV3
from .cfmms import PoolBatch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def uniform(seed: int, stream: int, count: int, first: int = 0):
    """count doubles in [0,1): element i depends only on (seed, stream, first+i)."""
    with np.errstate(over="ignore"):
        base = _splitmix64(np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(stream))
        ctr = np.arange(first, first + count, dtype=np.uint64)
        bits = _splitmix64(base + ctr * np.uint64(0xD1342543DE82EF95))
    return (bits >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def uniform_at(seed: int, stream: int, counters):
    """doubles in [0,1) at explicit counter values (same function of (seed, stream, counter) as `uniform`)."""
    with np.errstate(over="ignore"):
        base = _splitmix64(np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(stream))
        bits = _splitmix64(base + np.asarray(counters, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95))
    return (bits >> np.uint64(11)).astype(np.float64) * (1.0 / (1 << 53))


def token_pairs(seed, stream, m, n_tokens, first=0):
    """Two distinct 1-based token ids per pool, uniform over ordered pairs."""
    a = np.minimum((uniform(seed, stream, m, first) * n_tokens).astype(np.int64), n_tokens - 1)
    b = np.minimum((uniform(seed, stream + 1, m, first) * (n_tokens - 1)).astype(np.int64), n_tokens - 2)
    b = b + (b >= a)
    return np.stack([a + 1, b + 1], axis=1)


def product_pools(m, n_tokens, seed=1234, first=0):
    """m ProductTwoCoin pools, distributions of benchmark/scaling.jl:21-28."""
    R = 1000.0 * np.stack([uniform(seed, 10, m, first), uniform(seed, 11, m, first)], axis=1)
    R = np.maximum(R, 1e-9)  # U[0,1) can return exactly 0; reserves must be > 0
    γ = np.where(uniform(seed, 12, m, first) < 0.5, 0.997, 1.0)
    return PoolBatch(KIND_PRODUCT, R=R, γ=γ, Ai=token_pairs(seed, 13, m, n_tokens, first))


def geomean_pools(m, n_tokens, seed=1234, first=0):
    R = 1000.0 * np.stack([uniform(seed, 20, m, first), uniform(seed, 21, m, first)], axis=1)
    R = np.maximum(R, 1e-9)
    γ = np.where(uniform(seed, 22, m, first) < 0.5, 0.997, 1.0)
    w1 = np.clip(uniform(seed, 25, m, first), 0.02, 0.98)  # keep η = w₁/w₂ in [1/49, 49]
    w = np.stack([w1, 1.0 - w1], axis=1)
    return PoolBatch(KIND_GEOMEAN, R=R, w=w, γ=γ, Ai=token_pairs(seed, 23, m, n_tokens, first))


def token_price_vector(n_tokens, seed=1234):
    """The common price vector π behind `bounded_product_pools(..., consistent=True)`: exp(U[-1,1))."""
    return np.exp(2.0 * uniform(seed, 36, n_tokens) - 1.0)


def bounded_product_pools(m, n_tokens, seed=1234, first=0, consistent=False, noise=0.01):
    """m stand-alone BoundedProduct pools as 2-tick UniV3 (second tick empty), config 5.

    The reference gives no distribution for these (only the hand fixture test/cfmms.jl:117-119);
    this generator's choice: price band [p/(1+a), p(1+b)] around the current price p with
    a, b ~ U[0.05, 0.55), liquidity (the squared invariant k) ~ 1e6·U[0.01,1), and
      * consistent=False: p ~ exp(U[-1,1)) independently per pool -- an arbitrage-rich market whose dual
        optimum sits on the box corner (route! ends after 2 evaluations);
      * consistent=True: p = π[i₁]/π[i₂]·exp(noise·U[-1,1)) for one token price vector π
        (`token_price_vector`), i.e. a market close to no-arbitrage as real pools are, in which a
        BasketLiquidation has an INTERIOR dual optimum and route! has to find π (tens of evaluations)."""
    Ai = token_pairs(seed, 35, m, n_tokens, first)
    if consistent:
        π = token_price_vector(n_tokens, seed)
        p = π[Ai[:, 0] - 1] / π[Ai[:, 1] - 1] * np.exp(noise * (2.0 * uniform(seed, 30, m, first) - 1.0))
    else:
        p = np.exp(2.0 * uniform(seed, 30, m, first) - 1.0)
    a = 0.05 + 0.5 * uniform(seed, 31, m, first)
    b = 0.05 + 0.5 * uniform(seed, 32, m, first)
    k = 1e6 * (0.01 + 0.99 * uniform(seed, 33, m, first))
    γ = np.where(uniform(seed, 34, m, first) < 0.5, 0.997, 1.0)
    lower_ticks = np.stack([p * (1 + b), p / (1 + a)], axis=1).reshape(-1)
    liquidity = np.stack([k, np.zeros(m)], axis=1).reshape(-1)
    tick_off = 2 * np.arange(m + 1, dtype=np.int64)
    return PoolBatch(KIND_UNIV3, current_price=p, tick_off=tick_off, lower_ticks=lower_ticks,
                     liquidity=liquidity, γ=γ, Ai=Ai)


def univ3_pools(m, n_tokens, ticks_per_pool, seed=1234, first=0):
    """m UniV3 pools with `ticks_per_pool` geometric ticks each (last interval reaches price 0)."""
    t = int(ticks_per_pool)
    p = np.exp(2.0 * uniform(seed, 40, m, first) - 1.0)
    step = 1.0 + 0.01 + 0.1 * uniform(seed, 41, m, first)
    pos = 0.5 + (t - 1) * uniform(seed, 42, m, first)  # where the current price sits in the ladder
    j = np.arange(t)[None, :]
    lower_ticks = p[:, None] * step[:, None] ** (pos[:, None] - j)
    liq = 1e6 * (0.01 + uniform(seed, 43, m * t, first * t).reshape(m, t))
    empty = uniform(seed, 44, m * t, first * t).reshape(m, t) < 0.1
    liq = np.where(empty, 0.0, liq)
    γ = np.where(uniform(seed, 45, m, first) < 0.5, 0.997, 1.0)
    tick_off = t * np.arange(m + 1, dtype=np.int64)
    return PoolBatch(KIND_UNIV3, current_price=p, tick_off=tick_off, lower_ticks=lower_ticks.reshape(-1),
                     liquidity=liq.reshape(-1), γ=γ, Ai=token_pairs(seed, 46, m, n_tokens, first))


def univ3_ragged_pools(m, n_tokens, min_ticks=2, max_ticks=64, seed=1234, first=0, noise=0.01):
    """m UniV3 pools with RAGGED tick ladders (the multi-tick workload of bench.py --workload univ3_ticks).

    The reference gives no distribution (only the hand fixture test/cfmms.jl:117-119); this generator's choice, made to
    look like a concentrated-liquidity venue: the number of initialised ticks per pool is skewed towards few
    (t = min + floor((max − min + 1)·u³): half of the pools have <= 9 ticks at 2..64), ticks are geometric with a
    per-pool spacing of 1 %..6 %, one tick in ten is empty, liquidity ~ 1e6·U[0.01, 1.01) per tick, and the current price
    is the quote p = π[i₁]/π[i₂]·exp(noise·U[-1,1)) of one token price vector π (`token_price_vector`) placed anywhere
    inside the ladder -- so at prices a few per cent off π most pools trade inside their current tick and a sizeable
    minority walks through several.  Every value is a pure function of (seed, pool index, tick index): shards of a
    market regenerate the same pools."""
    Ai = token_pairs(seed, 76, m, n_tokens, first)
    π = token_price_vector(n_tokens, seed)
    p = π[Ai[:, 0] - 1] / π[Ai[:, 1] - 1] * np.exp(noise * (2.0 * uniform(seed, 70, m, first) - 1.0))
    span = max_ticks - min_ticks + 1
    t = np.minimum(min_ticks + np.floor(span * uniform(seed, 71, m, first) ** 3).astype(np.int64), max_ticks)
    step = 1.0 + 0.01 + 0.05 * uniform(seed, 72, m, first)
    pos = 0.5 + (t - 1) * uniform(seed, 73, m, first)          # where the current price sits in the ladder
    tick_off = np.concatenate([[0], np.cumsum(t)]).astype(np.int64)
    pool = np.repeat(np.arange(m, dtype=np.int64), t)
    j = np.arange(tick_off[-1], dtype=np.int64) - tick_off[pool]
    ctr = (np.uint64(first) + pool.astype(np.uint64)) * np.uint64(max_ticks) + j.astype(np.uint64)
    lower_ticks = p[pool] * step[pool] ** (pos[pool] - j)
    liq = 1e6 * (0.01 + uniform_at(seed, 74, ctr))
    liq = np.where(uniform_at(seed, 75, ctr) < 0.1, 0.0, liq)
    γ = np.where(uniform(seed, 77, m, first) < 0.5, 0.997, 1.0)
    return PoolBatch(KIND_UNIV3, current_price=p, tick_off=tick_off, lower_ticks=lower_ticks, liquidity=liq, γ=γ, Ai=Ai)


def univ3_ticks_visited(b, v):
    """Per pool of a UniV3 batch: how many ticks find_arb! (src/cfmms.jl:339-395) looks at for prices v -- 0 inside the
    no-arbitrage band (:347-349), else the ticks from the current one to the one that holds the target price p/γ
    (falling) or γ·p (rising), inclusive.  (Bench bookkeeping: the algorithmic bytes of a sweep are 32 B per pool + 16 B
    per tick VISITED)"""
    m = len(b)
    lt, off = b.lower_ticks, b.tick_off
    nt = np.diff(off)
    pool = np.repeat(np.arange(m, dtype=np.int64), nt)

    def index_of(x):   # searchsortedlast(lower_ticks, x, rev = true), 1-based, per pool (:235)
        return np.add.reduceat((lt >= x[pool]).astype(np.int64), off[:-1])

    pr = v[b.Ai[:, 0] - 1] / v[b.Ai[:, 1] - 1]
    g, cp = b.γ, b.current_price
    ct = index_of(cp)
    idle = (g * cp <= pr) & (pr <= cp / g)
    target = np.where(pr < g * cp, pr / g, g * pr)
    it = np.clip(index_of(target), 1, nt)
    return np.where(idle, 0, np.abs(it - ct) + 1)


def linear_prices(n_tokens, seed=1234):
    """c for LinearNonnegative(rand(n)) -- benchmark/scaling.jl:31."""
    return np.maximum(uniform(seed, 50, n_tokens), 2.0 ** -53)


def basket(n_tokens, seed=1234):
    """Δin = [0; 100·rand(n−1)] -- test/swap.jl:34."""
    d = 100.0 * uniform(seed, 51, n_tokens)
    d[0] = 0.0
    return d


def sweep_prices(n_tokens, seed=1234, spread=0.2):
    """A strictly positive price vector off the no-arbitrage manifold (for fixed-v sweeps)."""
    return np.exp(spread * (2.0 * uniform(seed, 60, n_tokens) - 1.0))
