"""Flat little-endian market files, so that every implementation (this package, the CPU checker in tests,
and the reference itself through bench/reference.jl) reads IDENTICAL pool bits (SURVEY §7.1:
Julia's `rand` stream cannot be reproduced outside Julia).

Layout (all little-endian; int = int64, real = float64; pair arrays are [m][2] row-major, which
Julia reads as a 2×m column-major Matrix):

    magic      8 bytes  "CFMMAMD1"
    n_tokens   int
    n_segments int
    objective  int kind (0 LinearNonnegative, 1 BasketLiquidation), int i (1-based, 0 if unused),
               real[n_tokens]  (c or Δin)
    has_v0     int (0/1), then real[n_tokens] if 1
    per segment: int kind (0 product, 1 geomean, 2 univ3), int m, then
        product: real R[m][2], real γ[m], int Ai[m][2] (1-based)
        geomean: real R[m][2], real w[m][2], real γ[m], int Ai[m][2]
        univ3:   real current_price[m], real γ[m], int Ai[m][2], int tick_off[m+1] (0-based CSR),
                 real lower_ticks[T], real liquidity[T]
"""
from __future__ import annotations

import numpy as np

from . import objectives as _obj
from ._lib import KIND_GEOMEAN, KIND_PRODUCT, KIND_UNIV3, ArgumentError
from .cfmms import PoolBatch

MAGIC = b"CFMMAMD1"


def _w(f, a, dtype):
    f.write(np.ascontiguousarray(a, dtype=dtype).tobytes())


def save_market(path, batches, n_tokens, objective, v0=None):
    with open(path, "wb") as f:
        f.write(MAGIC)
        _w(f, [n_tokens, len(batches)], "<i8")
        if isinstance(objective, _obj.LinearNonnegative):
            _w(f, [0, 0], "<i8")
            _w(f, objective.c, "<f8")
        elif isinstance(objective, _obj.BasketLiquidation):
            _w(f, [1, objective.i], "<i8")
            _w(f, objective.Δin, "<f8")
        else:
            raise ArgumentError("unknown objective")
        _w(f, [0 if v0 is None else 1], "<i8")
        if v0 is not None:
            _w(f, v0, "<f8")
        for b in batches:
            _w(f, [b.kind, len(b)], "<i8")
            if b.kind == KIND_PRODUCT:
                _w(f, b.R, "<f8"); _w(f, b.γ, "<f8"); _w(f, b.Ai, "<i8")
            elif b.kind == KIND_GEOMEAN:
                _w(f, b.R, "<f8"); _w(f, b.w, "<f8"); _w(f, b.γ, "<f8"); _w(f, b.Ai, "<i8")
            else:
                _w(f, b.current_price, "<f8"); _w(f, b.γ, "<f8"); _w(f, b.Ai, "<i8")
                _w(f, b.tick_off, "<i8"); _w(f, b.lower_ticks, "<f8"); _w(f, b.liquidity, "<f8")


def load_market(path):
    """-> (batches, n_tokens, objective, v0)"""
    buf = memoryview(open(path, "rb").read())
    if bytes(buf[:8]) != MAGIC:
        raise ArgumentError("not a CFMMAMD1 market file")
    pos = 8

    def rd(count, dtype):
        nonlocal pos
        a = np.frombuffer(buf, dtype=dtype, count=count, offset=pos)
        pos += a.nbytes
        return a.copy()

    n, nseg = (int(x) for x in rd(2, "<i8"))
    kind, idx = (int(x) for x in rd(2, "<i8"))
    vec = rd(n, "<f8")
    objective = _obj.LinearNonnegative(vec) if kind == 0 else _obj.BasketLiquidation(idx, vec)
    v0 = rd(n, "<f8") if int(rd(1, "<i8")[0]) else None
    batches = []
    for _ in range(nseg):
        k, m = (int(x) for x in rd(2, "<i8"))
        if k == KIND_PRODUCT:
            batches.append(PoolBatch(k, R=rd(2 * m, "<f8"), γ=rd(m, "<f8"), Ai=rd(2 * m, "<i8")))
        elif k == KIND_GEOMEAN:
            batches.append(PoolBatch(k, R=rd(2 * m, "<f8"), w=rd(2 * m, "<f8"), γ=rd(m, "<f8"), Ai=rd(2 * m, "<i8")))
        elif k == KIND_UNIV3:
            cp, g, ai, off = rd(m, "<f8"), rd(m, "<f8"), rd(2 * m, "<i8"), rd(m + 1, "<i8")
            T = int(off[-1])
            batches.append(PoolBatch(k, current_price=cp, γ=g, Ai=ai, tick_off=off, lower_ticks=rd(T, "<f8"),
                                     liquidity=rd(T, "<f8")))
        else:
            raise ArgumentError(f"unknown segment kind {k}")
    return batches, n, objective, v0
