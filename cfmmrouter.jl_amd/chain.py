"""Chain-data intake: on-chain pool state -> the pool batches of a Router.

The reference constructs its pools by hand in Julia (`examples/*.jl`); it has no loader.  Real markets arrive as
snapshots of contract state: raw integer reserves, token decimals, fees in basis points or pips, and -- for
concentrated liquidity -- `sqrtPriceX96`, the initialized ticks and their `liquidityNet`.  This module converts one
such snapshot (JSON lines, one pool per line; integers may be decimal strings, as RPC clients deliver them) into
the package's `PoolBatch`es, in human units, with the token universe numbered in order of first appearance.

One line per pool:

    {"type": "constant_product", "tokens": ["0xA..", "0xB.."], "decimals": [18, 6],
     "reserves": ["123..", "456.."], "fee_bps": 30}                                    Uniswap-v2 style pairs
    {"type": "weighted", "tokens": [...], "decimals": [...], "balances": [...],
     "weights": [0.8, 0.2], "fee": 0.001}                                              Balancer-style 2-token pools
    {"type": "concentrated", "tokens": [token0, token1], "decimals": [d0, d1], "fee_pips": 3000,
     "sqrt_price_x96": "...", "liquidity": "...", "ticks": [[index, liquidity_net], ...]}   Uniswap-v3 style pools

Fees: exactly one of "fee" (fraction), "fee_bps" (1e-4) or "fee_pips" (1e-6); γ = 1 − fee.

Concentrated liquidity -> the reference's `UniV3(current_price, lower_ticks, liquidity, γ, Ai)` (src/cfmms.jl:226-245):
  * price = amount of token1 per token0 = (sqrtPriceX96 / 2^96)², in human units × 10^(d0 − d1); the reference's
    coin 1 is token0 and coin 2 is token1 (`curr_price = (R₂+β)/(R₁+α)`, :283);
  * between two adjacent initialized ticks t_j < t_{j+1} the active liquidity is L_j = Σ_{i ≤ j} liquidityNet_i; the
    reference's per-interval `liquidity` is k = L² (`α = sqrt(k/p₊)`, `β = sqrt(k·p₋)`, :294-300), in human units
    L² / 10^(d0 + d1);
  * the reference lists, in DESCENDING order, the UPPER price of every interval (`tick_high_price`, :249; the field is
    called `lower_ticks`), the last interval reaching down to price 0 (:252-257): one entry per pair of adjacent
    initialized ticks, plus an empty interval below the lowest tick (and one above the highest when the pool's price
    sits there).
"""
from __future__ import annotations

import json
from fractions import Fraction

import numpy as np

from ._lib import ArgumentError
from .cfmms import GeometricMeanTwoCoin, ProductTwoCoin, UniV3

Q96 = 1 << 96
TICK_BASE = 1.0001


def _int(x, what):
    try:
        return int(x)
    except (TypeError, ValueError):
        raise ArgumentError(f"{what}: not an integer: {x!r}") from None


def _gamma(rec, where):
    keys = [k for k in ("fee", "fee_bps", "fee_pips") if k in rec]
    if len(keys) != 1:
        raise ArgumentError(f"{where}: give exactly one of fee / fee_bps / fee_pips")
    fee = float(rec[keys[0]]) * {"fee": 1.0, "fee_bps": 1e-4, "fee_pips": 1e-6}[keys[0]]
    if not 0.0 <= fee < 1.0:
        raise ArgumentError(f"{where}: fee {fee} outside [0, 1)")
    return 1.0 - fee


def _amount(raw, decimals, where):
    v = _int(raw, where)
    if v <= 0:
        raise ArgumentError(f"{where}: reserves must be > 0")
    return float(Fraction(v, 10 ** int(decimals)))          # exact rational, one rounding


def tick_price(index, d0=0, d1=0):
    """Price (token1 per token0, human units) at a tick index: 1.0001^index · 10^(d0 − d1)."""
    return TICK_BASE ** int(index) * 10.0 ** (int(d0) - int(d1))


def concentrated_to_univ3(sqrt_price_x96, ticks, d0, d1, liquidity=None, where="pool"):
    """-> (current_price, upper_prices descending, k per interval) in the reference's UniV3 parametrisation."""
    s = _int(sqrt_price_x96, where + ".sqrt_price_x96")
    if s <= 0:
        raise ArgumentError(f"{where}: sqrt_price_x96 must be > 0")
    scale = 10.0 ** (int(d0) - int(d1))
    price = float(Fraction(s * s, Q96 * Q96)) * scale
    init = sorted((int(t), _int(net, where + ".ticks")) for t, net in ticks)
    if len(init) < 2:
        raise ArgumentError(f"{where}: at least two initialized ticks are needed")
    if any(a[0] == b[0] for a, b in zip(init, init[1:])):
        raise ArgumentError(f"{where}: duplicate tick index")
    L, active = [], 0
    for t, net in init[:-1]:
        active += net
        if active < 0:
            raise ArgumentError(f"{where}: liquidityNet sums to a negative liquidity at tick {t}")
        L.append(active)
    if active + init[-1][1] != 0:
        raise ArgumentError(f"{where}: liquidityNet does not sum to zero over the initialized ticks")
    unit = 10.0 ** (-(int(d0) + int(d1)))
    uppers = [tick_price(t, d0, d1) for t, _ in init[1:]]          # upper price of interval j = price at t_{j+1}
    ks = [float(l) * float(l) * unit for l in L]
    uppers, ks = uppers[::-1], ks[::-1]                             # descending, as the reference stores them
    uppers.append(tick_price(init[0][0], d0, d1))                   # below the lowest tick: empty, down to price 0
    ks.append(0.0)
    if price > uppers[0]:                                           # above the highest tick: empty as well
        uppers.insert(0, price * TICK_BASE)
        ks.insert(0, 0.0)
    if liquidity is not None:                                       # cross-check against the pool's own `liquidity` slot
        want = float(_int(liquidity, where + ".liquidity")) ** 2 * unit
        idx = int(np.searchsorted(-np.asarray(uppers), -price, side="right")) - 1   # searchsortedlast(rev=true), 0-based
        have = ks[max(idx, 0)]
        if abs(have - want) > 1e-9 * max(have, want, 1e-300):
            raise ArgumentError(f"{where}: active liquidity {want:g} (liquidity slot) does not match the ticks ({have:g})")
    return price, uppers, ks


def load_snapshot(source):
    """source: path to a JSON-lines file, or an iterable of dicts / JSON strings.
    -> (tokens, batches): the token identifiers in index order (index k+1 is the reference's 1-based token id) and the
    pool batches [ProductTwoCoin..., GeometricMeanTwoCoin..., UniV3...] (families that do not occur are omitted)."""
    if isinstance(source, (str, bytes)):
        with open(source) as f:
            records = [json.loads(line) for line in f if line.strip() and not line.lstrip().startswith("#")]
    else:
        records = [json.loads(r) if isinstance(r, (str, bytes)) else r for r in source]
    tokens, index = [], {}

    def tid(name):
        if name not in index:
            index[name] = len(tokens) + 1
            tokens.append(name)
        return index[name]

    prod, geo, conc = [], [], []
    for k, rec in enumerate(records):
        where = f"pool {k}"
        toks = rec.get("tokens")
        if not isinstance(toks, (list, tuple)) or len(toks) != 2 or toks[0] == toks[1]:
            raise ArgumentError(f"{where}: tokens must be two distinct identifiers")
        dec = rec.get("decimals", [18, 18])
        if len(dec) != 2:
            raise ArgumentError(f"{where}: decimals must have two entries")
        ai = [tid(toks[0]), tid(toks[1])]
        g = _gamma(rec, where)
        kind = rec.get("type")
        if kind == "constant_product":
            r = rec.get("reserves")
            prod.append(([_amount(r[0], dec[0], where), _amount(r[1], dec[1], where)], g, ai))
        elif kind == "weighted":
            r, w = rec.get("balances"), [float(x) for x in rec.get("weights", ())]
            if len(w) != 2 or min(w) <= 0:
                raise ArgumentError(f"{where}: two positive weights are needed")
            tot = w[0] + w[1]
            geo.append(([_amount(r[0], dec[0], where), _amount(r[1], dec[1], where)], [w[0] / tot, w[1] / tot], g, ai))
        elif kind == "concentrated":
            p, up, ks = concentrated_to_univ3(rec.get("sqrt_price_x96"), rec.get("ticks", ()), dec[0], dec[1],
                                              rec.get("liquidity"), where)
            conc.append((p, up, ks, g, ai))
        else:
            raise ArgumentError(f"{where}: unknown pool type {kind!r}")
    batches = []
    if prod:
        batches.append(ProductTwoCoin.batch([p[0] for p in prod], [p[1] for p in prod], [p[2] for p in prod]))
    if geo:
        batches.append(GeometricMeanTwoCoin.batch([p[0] for p in geo], [p[1] for p in geo], [p[2] for p in geo],
                                                  [p[3] for p in geo]))
    if conc:
        off = np.zeros(len(conc) + 1, dtype=np.int64)
        np.cumsum([len(c[1]) for c in conc], out=off[1:])
        batches.append(UniV3.batch([c[0] for c in conc], off, np.concatenate([c[1] for c in conc]),
                                   np.concatenate([c[2] for c in conc]), [c[3] for c in conc], [c[4] for c in conc]))
    return tokens, batches
