"""Host-side mirror of the reference's objectives (src/objectives.jl).

O(n_tokens) arithmetic that sets the L-BFGS-B box and adds ∇f to the gradient; it runs on the
host exactly as in the reference (out of scope as a kernel, must be restated exactly).
Generic functions keep the reference's names: `f`, `grad_` (Julia `grad!`), `lower_limit`,
`upper_limit`.  Token index `i` of BasketLiquidation / Swap is 1-based like the reference.
"""
from __future__ import annotations

import math

import numpy as np

from ._lib import ArgumentError

_SQRT_EPS = math.sqrt(np.finfo(np.float64).eps)


class Objective:
    """abstract type Objective -- src/objectives.jl:3"""


class LinearNonnegative(Objective):
    """U(Ψ) = cᵀΨ − I(Ψ ≥ 0) -- src/objectives.jl:41-59"""

    def __init__(self, c):
        c = np.array(c, dtype=np.float64).reshape(-1)  # Float64.(c), :59
        if not np.all(c > 0):
            raise ArgumentError("all elements must be strictly positive")  # :54
        self.c = c


class BasketLiquidation(Objective):
    """Ψ_i − I(Ψ_{-i} + Δin_{-i} = 0, Ψ_i ≥ 0) -- src/objectives.jl:82-103 (i is 1-based)"""

    def __init__(self, i, Δin):
        Δin = np.array(Δin, dtype=np.float64).reshape(-1)  # :103
        if not (i > 0 and i <= Δin.size):
            raise ArgumentError("Invalid index i")  # :97
        self.i = int(i)
        self.Δin = Δin

    Din = property(lambda self: self.Δin)


def Swap(i, j, δ, n):
    """Swap(i, j, δ, n): one-hot BasketLiquidation -- src/objectives.jl:131-146"""
    Δin = np.zeros(int(n))
    Δin[j - 1] = δ
    return BasketLiquidation(i, Δin)


def f(obj, v):
    """f(obj, v): conjugate of the utility at v -- src/objectives.jl:62-67, :106-111"""
    v = np.asarray(v, dtype=np.float64)
    if isinstance(obj, LinearNonnegative):
        return 0.0 if np.all(obj.c <= v) else math.inf
    if isinstance(obj, BasketLiquidation):
        if v[obj.i - 1] >= 1.0:
            s = 0.0
            for j in range(v.size):  # left-to-right like Base.sum below its pairwise block size
                s += 0.0 if j == obj.i - 1 else obj.Δin[j] * v[j]
            return s
        return math.inf
    raise TypeError(f"no method f for {type(obj).__name__}")


def grad_(g, obj, v):
    """grad!(g, obj, v) -- src/objectives.jl:69-76, :113-121"""
    v = np.asarray(v, dtype=np.float64)
    if isinstance(obj, LinearNonnegative):
        g[:] = 0.0 if np.all(obj.c <= v) else math.inf
        return None
    if isinstance(obj, BasketLiquidation):
        if v[obj.i - 1] >= 1.0:
            g[:] = obj.Δin
            g[obj.i - 1] = 0.0
        else:
            g[:] = math.inf
        return None
    raise TypeError(f"no method grad! for {type(obj).__name__}")


def lower_limit(obj):
    """lower_limit(obj) -- src/objectives.jl:78, :123-128"""
    if isinstance(obj, LinearNonnegative):
        return obj.c + 1e-8
    if isinstance(obj, BasketLiquidation):
        ret = np.full(obj.Δin.size, _SQRT_EPS)
        ret[obj.i - 1] = 1.0 + _SQRT_EPS
        return ret
    raise TypeError(f"no method lower_limit for {type(obj).__name__}")


def upper_limit(obj):
    """upper_limit(obj) -- src/objectives.jl:79, :129"""
    if isinstance(obj, LinearNonnegative):
        return math.inf + np.zeros_like(obj.c)
    if isinstance(obj, BasketLiquidation):
        return math.inf + np.zeros_like(obj.Δin)
    raise TypeError(f"no method upper_limit for {type(obj).__name__}")


def n_tokens_of(obj):
    return obj.c.size if isinstance(obj, LinearNonnegative) else obj.Δin.size
