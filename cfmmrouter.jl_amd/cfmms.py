"""Host-side mirror of the reference's pool types (src/cfmms.jl).

Same names, same constructor arguments, same error behaviour as the Julia structs, so code and
tests written against the reference read the same here:

    ProductTwoCoin(R, γ, idx)                src/cfmms.jl:101-111
    GeometricMeanTwoCoin(R, w, γ, idx)       src/cfmms.jl:152-165
    UniV3(current_price, lower_ticks, liquidity, γ, Ai)   src/cfmms.jl:226-245

Token indices are 1-BASED, exactly as in the reference (`Ai[j]` is the global id of the pool's
j-th coin); they are converted to 0-based int32 once, when a Router packs the pools for the
device.  The objects here hold data only: all arithmetic of `find_arb!` happens on the GPU
through the C ABI (`find_arb_(Δ, Λ, cfmm, v)` below is a one-pool device sweep).

For large markets, building m Python objects is the slow part, so every family also has a
structure-of-arrays batch (`ProductTwoCoin.batch(R[m,2], γ[m], idx[m,2])` ...) that a Router
accepts directly; `batch[i]` materialises the i-th pool object on demand.
"""
from __future__ import annotations

import numpy as np

from ._lib import KIND_GEOMEAN, KIND_PRODUCT, KIND_UNIV3, ArgumentError


class CFMM:
    """abstract type CFMM{T} -- src/cfmms.jl:5"""

    kind = -1

    def __len__(self):  # Base.length(c::CFMM) = length(c.Ai), src/cfmms.jl:19
        return len(self.Ai)


def _two_coin_check_cast(R, γ, idx):
    """two_coin_check_cast -- src/cfmms.jl:76-90"""
    R = np.asarray(R)
    idx = np.asarray(idx)
    if R.size != 2:
        raise ArgumentError("length of R must be 2 for *TwoCoin constructors")
    if idx.size != 2:
        raise ArgumentError("length of idx must be 2 for *TwoCoin constructors")
    if np.any(idx < 0):  # convert.(UInt, idx) throws InexactError on negatives
        raise ArgumentError("idx must be non-negative")
    return np.array(R, dtype=np.float64).reshape(2), float(γ), np.array(idx, dtype=np.int64).reshape(2)


class ProductTwoCoin(CFMM):
    """ProductTwoCoin(R, γ, idx): φ(R) = R₁R₂ -- src/cfmms.jl:92-111"""

    kind = KIND_PRODUCT

    def __init__(self, R, γ, idx):
        self.R, self.γ, self.Ai = _two_coin_check_cast(R, γ, idx)

    gamma = property(lambda self: self.γ)

    @staticmethod
    def batch(R, γ, idx):
        return PoolBatch(KIND_PRODUCT, R=R, γ=γ, Ai=idx)


class GeometricMeanTwoCoin(CFMM):
    """GeometricMeanTwoCoin(R, w, γ, idx): φ(R) = R₁^w₁ R₂^w₂ -- src/cfmms.jl:142-165"""

    kind = KIND_GEOMEAN

    def __init__(self, R, w, γ, idx):
        self.R, self.γ, self.Ai = _two_coin_check_cast(R, γ, idx)
        w = np.array(w, dtype=np.float64)
        if w.size != 2:
            raise ArgumentError("length of w must be 2")  # SVector{2,T}(w) would throw
        self.w = w.reshape(2)

    gamma = property(lambda self: self.γ)

    @staticmethod
    def batch(R, w, γ, idx):
        return PoolBatch(KIND_GEOMEAN, R=R, w=w, γ=γ, Ai=idx)


def ϕ(cfmm, R=None):
    """ϕ(c::CFMM; R=nothing): the trading function -- src/cfmms.jl:36-42, :113-116 (ProductTwoCoin:
    R₁R₂), :167-171 (GeometricMeanTwoCoin: R₁^w₁ R₂^w₂).  The reference defines no method for UniV3.
    Host-side definition (O(1) per pool, used by the optimality tests, not by the sweep)."""
    if not isinstance(cfmm, (ProductTwoCoin, GeometricMeanTwoCoin)):
        raise ArgumentError("ϕ has no method for this pool type (as in the reference)")
    R = cfmm.R if R is None else np.asarray(R, dtype=np.float64)
    if isinstance(cfmm, ProductTwoCoin):
        return R[0] * R[1]
    if isinstance(cfmm, GeometricMeanTwoCoin):
        return R[0] ** cfmm.w[0] * R[1] ** cfmm.w[1]
    raise ArgumentError("ϕ has no method for this pool type (as in the reference)")


def ϕ_grad_(out, cfmm, R=None):
    """∇ϕ!(x, c::CFMM; R=nothing): gradient of the trading function, stored in `out` --
    src/cfmms.jl:44-50, :117-122, :172-178."""
    if not isinstance(cfmm, (ProductTwoCoin, GeometricMeanTwoCoin)):
        raise ArgumentError("∇ϕ! has no method for this pool type (as in the reference)")
    R = cfmm.R if R is None else np.asarray(R, dtype=np.float64)
    if isinstance(cfmm, ProductTwoCoin):
        out[0], out[1] = R[1], R[0]
        return None
    if isinstance(cfmm, GeometricMeanTwoCoin):
        w = cfmm.w
        out[0] = w[0] * (R[1] / R[0]) ** w[1]
        out[1] = w[1] * (R[0] / R[1]) ** w[0]
        return None
    raise ArgumentError("∇ϕ! has no method for this pool type (as in the reference)")


phi, grad_phi_ = ϕ, ϕ_grad_   # ASCII spellings ("∇" is not a valid Python identifier character, hence ϕ_grad_ for ∇ϕ!)


class UniV3(CFMM):
    """UniV3(current_price, lower_ticks, liquidity, γ, Ai) -- src/cfmms.jl:206-245.

    `lower_ticks` is in decreasing order; `current_tick` is
    searchsortedlast(lower_ticks, current_price, rev=true) (:235), 1-based."""

    kind = KIND_UNIV3

    def __init__(self, current_price, lower_ticks, liquidity, γ, Ai):
        self.current_price = float(current_price)
        self.lower_ticks = np.array(lower_ticks, dtype=np.float64).reshape(-1)
        self.liquidity = np.array(liquidity, dtype=np.float64).reshape(-1)
        if self.lower_ticks.size != self.liquidity.size:
            raise ArgumentError("lower_ticks and liquidity must have the same length")
        self.γ = float(γ)
        self.Ai = np.array(Ai, dtype=np.int64).reshape(-1)
        if self.Ai.size != 2:
            raise ArgumentError("length of Ai must be 2")
        # number of ticks >= current_price in the descending vector (== searchsortedlast, rev=true)
        self.current_tick = int(np.count_nonzero(self.lower_ticks >= self.current_price))

    gamma = property(lambda self: self.γ)

    @staticmethod
    def batch(current_price, tick_off, lower_ticks, liquidity, γ, Ai):
        return PoolBatch(KIND_UNIV3, current_price=current_price, tick_off=tick_off,
                         lower_ticks=lower_ticks, liquidity=liquidity, γ=γ, Ai=Ai)


def BoundedProduct(current_price, p_lower, p_upper, liquidity, γ, Ai):
    """A stand-alone bounded-liquidity pool φ(R) = (R₁+α)(R₂+β) (src/cfmms.jl:261-289) on the
    price interval [p_lower, p_upper].  The reference's BoundedProduct struct is not a CFMM
    subtype and cannot enter a Router; the routable form is a UniV3 with two ticks whose second
    interval is empty (cf. the trailing 0.0 of the fixture at test/cfmms.jl:118-119)."""
    return UniV3(current_price, [p_upper, p_lower], [liquidity, 0.0], γ, Ai)


class PoolBatch:
    """m pools of one family, structure-of-arrays (the HBM layout, on the host).

    Ai is 1-based [m, 2] like the reference's per-pool `Ai`."""

    def __init__(self, kind, **a):
        self.kind = kind
        self.γ = np.ascontiguousarray(a["γ"], dtype=np.float64).reshape(-1)
        m = self.γ.size
        self.Ai = np.ascontiguousarray(a["Ai"], dtype=np.int64).reshape(m, 2)
        if kind in (KIND_PRODUCT, KIND_GEOMEAN):
            self.R = np.ascontiguousarray(a["R"], dtype=np.float64).reshape(m, 2)
        if kind == KIND_GEOMEAN:
            self.w = np.ascontiguousarray(a["w"], dtype=np.float64).reshape(m, 2)
        if kind == KIND_UNIV3:
            self.current_price = np.ascontiguousarray(a["current_price"], dtype=np.float64).reshape(m)
            self.tick_off = np.ascontiguousarray(a["tick_off"], dtype=np.int64).reshape(m + 1)
            self.lower_ticks = np.ascontiguousarray(a["lower_ticks"], dtype=np.float64).reshape(-1)
            self.liquidity = np.ascontiguousarray(a["liquidity"], dtype=np.float64).reshape(-1)

    def __len__(self):
        return self.γ.size

    def __getitem__(self, i):
        if isinstance(i, slice):
            return self.slice(*i.indices(len(self))[:2])
        i = int(i)
        if i < 0:
            i += len(self)
        if self.kind == KIND_PRODUCT:
            return ProductTwoCoin(self.R[i], self.γ[i], self.Ai[i])
        if self.kind == KIND_GEOMEAN:
            return GeometricMeanTwoCoin(self.R[i], self.w[i], self.γ[i], self.Ai[i])
        o, e = self.tick_off[i], self.tick_off[i + 1]
        return UniV3(self.current_price[i], self.lower_ticks[o:e], self.liquidity[o:e], self.γ[i], self.Ai[i])

    def slice(self, lo, hi):
        """Pools [lo, hi) as a new batch (used to shard a market across GPUs)."""
        if self.kind == KIND_PRODUCT:
            return PoolBatch(self.kind, R=self.R[lo:hi], γ=self.γ[lo:hi], Ai=self.Ai[lo:hi])
        if self.kind == KIND_GEOMEAN:
            return PoolBatch(self.kind, R=self.R[lo:hi], w=self.w[lo:hi], γ=self.γ[lo:hi], Ai=self.Ai[lo:hi])
        o, e = self.tick_off[lo], self.tick_off[hi]
        return PoolBatch(self.kind, current_price=self.current_price[lo:hi],
                         tick_off=self.tick_off[lo:hi + 1] - o, lower_ticks=self.lower_ticks[o:e],
                         liquidity=self.liquidity[o:e], γ=self.γ[lo:hi], Ai=self.Ai[lo:hi])

    @staticmethod
    def concat(batches):
        """One batch holding the pools of several same-family batches, in order."""
        batches = list(batches)
        kind = batches[0].kind
        if any(b.kind != kind for b in batches):
            raise ArgumentError("concat needs batches of one pool family")
        cat = lambda name: np.concatenate([getattr(b, name) for b in batches])
        if kind == KIND_PRODUCT:
            return PoolBatch(kind, R=cat("R"), γ=cat("γ"), Ai=cat("Ai"))
        if kind == KIND_GEOMEAN:
            return PoolBatch(kind, R=cat("R"), w=cat("w"), γ=cat("γ"), Ai=cat("Ai"))
        off, base = [np.zeros(1, dtype=np.int64)], 0
        for b in batches:
            off.append(b.tick_off[1:] + base)
            base += int(b.tick_off[-1])
        return PoolBatch(kind, current_price=cat("current_price"), tick_off=np.concatenate(off),
                         lower_ticks=cat("lower_ticks"), liquidity=cat("liquidity"), γ=cat("γ"), Ai=cat("Ai"))

    @staticmethod
    def from_pools(kind, pools):
        if kind == KIND_PRODUCT:
            return PoolBatch(kind, R=[p.R for p in pools], γ=[p.γ for p in pools], Ai=[p.Ai for p in pools])
        if kind == KIND_GEOMEAN:
            return PoolBatch(kind, R=[p.R for p in pools], w=[p.w for p in pools], γ=[p.γ for p in pools],
                             Ai=[p.Ai for p in pools])
        off = np.zeros(len(pools) + 1, dtype=np.int64)
        np.cumsum([p.lower_ticks.size for p in pools], out=off[1:])
        return PoolBatch(kind, current_price=[p.current_price for p in pools], tick_off=off,
                         lower_ticks=np.concatenate([p.lower_ticks for p in pools]) if pools else [],
                         liquidity=np.concatenate([p.liquidity for p in pools]) if pools else [],
                         γ=[p.γ for p in pools], Ai=[p.Ai for p in pools])


def zerotrade(c):
    """zerotrade(c) -- src/cfmms.jl:73,248"""
    return np.zeros(2)


def find_arb_(Δ, Λ, cfmm, v, device=0):
    """find_arb!(Δ, Λ, cfmm, v) -- src/cfmms.jl:35 and the methods at :130, :185, :339.

    Solves one pool's arbitrage problem at local prices `v` (length 2) ON THE DEVICE and
    overwrites Δ and Λ.  Convenience for tests and examples; routers sweep all pools at once."""
    from ._lib import Context

    v = np.asarray(v, dtype=np.float64).reshape(2)
    ctx = Context(2, device)
    try:
        _upload(ctx, PoolBatch.from_pools(cfmm.kind, [_with_local_idx(cfmm)]))
        ctx.find_arb(v)
        D, Lm = ctx.trades()
    finally:
        ctx.close()
    Δ[:] = D[0]
    Λ[:] = Lm[0]
    return None


def _with_local_idx(c):
    if c.kind == KIND_PRODUCT:
        return ProductTwoCoin(c.R, c.γ, [1, 2])
    if c.kind == KIND_GEOMEAN:
        return GeometricMeanTwoCoin(c.R, c.w, c.γ, [1, 2])
    return UniV3(c.current_price, c.lower_ticks, c.liquidity, c.γ, [1, 2])


def _upload(ctx, batch: PoolBatch):
    """Append one homogeneous batch to the device pool store (1-based -> 0-based here)."""
    Ai0 = (batch.Ai - 1).astype(np.int32)
    if np.any(batch.Ai < 1) or np.any(batch.Ai > ctx.n_tokens):
        raise ArgumentError(f"token index out of range 1:{ctx.n_tokens}")
    if batch.kind == KIND_PRODUCT:
        ctx.add_product(batch.R, batch.γ, Ai0)
    elif batch.kind == KIND_GEOMEAN:
        ctx.add_geomean(batch.R, batch.w, batch.γ, Ai0)
    elif batch.kind == KIND_UNIV3:
        ctx.add_univ3(batch.current_price, batch.γ, Ai0, batch.tick_off, batch.lower_ticks, batch.liquidity)
    else:
        raise ArgumentError("unknown pool family")
