"""ctypes front-end for the CPU ORACLE (oracle/cfmm_oracle.c).

TEST INFRASTRUCTURE, NOT PRODUCT CODE: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this module.  The shipped package (cfmmrouter.jl_amd/) never does.

All index arguments of this wrapper are 0-based (the harness converts the reference's 1-based
`Ai`).  Pair arrays are float64 / int32 of shape [m, 2], C-contiguous.

`route_oracle` restates route! (src/router.jl:58-108) on top of the running interpreter's SciPy
L-BFGS-B (1.15 here: a C translation of the Fortran L-BFGS-B 3.0 that LBFGSB.jl wraps).  The
outer loop's PIN is elsewhere: tests/golden/route_fortran.npz, runs of the Fortran code itself
(tests/golden/make_route_golden.py, tests/test_route_fortran_pin.py).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

_f64p = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)


def build(force: bool = False) -> str:
    """Compile liboracle.so with gcc (oracle/Makefile)."""
    if force or not os.path.exists(_LIB_PATH) or (
        os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "cfmm_oracle.c"))
    ):
        subprocess.run(["make", "-C", _HERE, "-B", "liboracle.so"], check=True,
                       stdout=subprocess.DEVNULL)
    return _LIB_PATH


class BoundedProduct(C.Structure):
    _fields_ = [("k", C.c_double), ("alpha", C.c_double), ("beta", C.c_double),
                ("R_1", C.c_double), ("R_2", C.c_double)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        L.oracle_product_find_arb.argtypes = [_f64p, C.c_double, _f64p, _f64p, _f64p]
        L.oracle_product_phi.argtypes = [_f64p]
        L.oracle_product_phi.restype = C.c_double
        L.oracle_product_grad_phi.argtypes = [_f64p, _f64p]
        L.oracle_geomean_find_arb.argtypes = [_f64p, _f64p, C.c_double, _f64p, _f64p, _f64p]
        L.oracle_geomean_phi.argtypes = [_f64p, _f64p]
        L.oracle_geomean_phi.restype = C.c_double
        L.oracle_geomean_grad_phi.argtypes = [_f64p, _f64p, _f64p]
        L.oracle_univ3_current_tick.argtypes = [_f64p, C.c_int64, C.c_double]
        L.oracle_univ3_current_tick.restype = C.c_int64
        L.oracle_univ3_compute_at_tick.argtypes = [C.c_double, C.c_int64, _f64p, _f64p, C.c_int64, C.c_int64]
        L.oracle_univ3_compute_at_tick.restype = BoundedProduct
        L.oracle_flip_sides.argtypes = [BoundedProduct]
        L.oracle_flip_sides.restype = BoundedProduct
        L.oracle_find_arb_pos.argtypes = [BoundedProduct, C.c_double, _f64p, _f64p]
        L.oracle_univ3_find_arb.argtypes = [C.c_double, C.c_int64, _f64p, _f64p, C.c_int64, C.c_double,
                                            _f64p, _f64p, _f64p]
        L.oracle_univ3_forward_trade.argtypes = [_f64p, C.c_double, C.c_int64, _f64p, _f64p, C.c_int64,
                                                 C.c_double]
        L.oracle_univ3_forward_trade.restype = C.c_double
        L.oracle_sweep_product.argtypes = [C.c_int64, _f64p, _f64p, _i32p, _f64p, _f64p, _f64p, C.c_int]
        L.oracle_sweep_geomean.argtypes = [C.c_int64, _f64p, _f64p, _f64p, _i32p, _f64p, _f64p, _f64p, C.c_int]
        L.oracle_sweep_univ3.argtypes = [C.c_int64, _f64p, _i64p, _f64p, _i32p, _i64p, _f64p, _f64p, _f64p,
                                         _f64p, _f64p, C.c_int]
        L.oracle_dual_acc.argtypes = [C.c_int64, _f64p, _f64p, _i32p, _f64p]
        L.oracle_dual_acc.restype = C.c_double
        L.oracle_grad_scatter.argtypes = [C.c_int64, _f64p, _f64p, _i32p, _f64p]
        L.oracle_netflows.argtypes = [C.c_int64, _f64p, _f64p, _i32p, C.c_int64, _f64p]
        L.oracle_linear_nonneg_f.argtypes = [_f64p, _f64p, C.c_int64]
        L.oracle_linear_nonneg_f.restype = C.c_double
        L.oracle_linear_nonneg_grad.argtypes = [_f64p, _f64p, _f64p, C.c_int64]
        L.oracle_linear_nonneg_lower.argtypes = [_f64p, _f64p, C.c_int64]
        L.oracle_basket_liq_f.argtypes = [C.c_int64, _f64p, _f64p, C.c_int64]
        L.oracle_basket_liq_f.restype = C.c_double
        L.oracle_basket_liq_grad.argtypes = [_f64p, C.c_int64, _f64p, _f64p, C.c_int64]
        L.oracle_basket_liq_lower.argtypes = [_f64p, C.c_int64, C.c_int64]
        L.oracle_max_threads.restype = C.c_int
        _lib = L
    return _lib


def _f64(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if shape is not None:
        a = a.reshape(shape)
    return a


def _p(a):
    if a.dtype == np.float64:
        return a.ctypes.data_as(_f64p)
    if a.dtype == np.int32:
        return a.ctypes.data_as(_i32p)
    if a.dtype == np.int64:
        return a.ctypes.data_as(_i64p)
    raise TypeError(a.dtype)


# ---- per-pool ---------------------------------------------------------------------------------

def product_find_arb(R, gamma, v):
    R, v = _f64(R, (2,)), _f64(v, (2,))
    D, Lm = np.empty(2), np.empty(2)
    lib().oracle_product_find_arb(_p(R), float(gamma), _p(v), _p(D), _p(Lm))
    return D, Lm


def product_phi(R):
    return lib().oracle_product_phi(_p(_f64(R, (2,))))


def product_grad_phi(R):
    out = np.empty(2)
    lib().oracle_product_grad_phi(_p(_f64(R, (2,))), _p(out))
    return out


def geomean_find_arb(R, w, gamma, v):
    R, w, v = _f64(R, (2,)), _f64(w, (2,)), _f64(v, (2,))
    D, Lm = np.empty(2), np.empty(2)
    lib().oracle_geomean_find_arb(_p(R), _p(w), float(gamma), _p(v), _p(D), _p(Lm))
    return D, Lm


def geomean_phi(R, w):
    return lib().oracle_geomean_phi(_p(_f64(R, (2,))), _p(_f64(w, (2,))))


def geomean_grad_phi(R, w):
    out = np.empty(2)
    lib().oracle_geomean_grad_phi(_p(_f64(R, (2,))), _p(_f64(w, (2,))), _p(out))
    return out


class UniV3:
    """Plain data holder mirroring the reference struct (src/cfmms.jl:226-245)."""

    def __init__(self, current_price, lower_ticks, liquidity, gamma):
        self.current_price = float(current_price)
        self.lower_ticks = _f64(lower_ticks)
        self.liquidity = _f64(liquidity)
        self.gamma = float(gamma)
        self.n_ticks = int(self.lower_ticks.size)
        self.current_tick = int(lib().oracle_univ3_current_tick(_p(self.lower_ticks), self.n_ticks,
                                                                self.current_price))

    def compute_at_tick(self, idx):
        return lib().oracle_univ3_compute_at_tick(self.current_price, self.current_tick,
                                                  _p(self.lower_ticks), _p(self.liquidity),
                                                  self.n_ticks, int(idx))

    def find_arb(self, v):
        v = _f64(v, (2,))
        D, Lm = np.empty(2), np.empty(2)
        lib().oracle_univ3_find_arb(self.current_price, self.current_tick, _p(self.lower_ticks),
                                    _p(self.liquidity), self.n_ticks, self.gamma, _p(v), _p(D), _p(Lm))
        return D, Lm

    def forward_trade(self, Delta):
        Delta = _f64(Delta, (2,))
        return lib().oracle_univ3_forward_trade(_p(Delta), self.current_price, self.current_tick,
                                                _p(self.lower_ticks), _p(self.liquidity), self.n_ticks,
                                                self.gamma)


def find_arb_pos(t: BoundedProduct, price: float):
    d, l = C.c_double(), C.c_double()
    lib().oracle_find_arb_pos(t, float(price), C.byref(d), C.byref(l))
    return d.value, l.value


# ---- sweeps -----------------------------------------------------------------------------------

def sweep_product(R, gamma, Ai, v, nthreads=1):
    R, gamma, v = _f64(R), _f64(gamma), _f64(v)
    Ai = np.ascontiguousarray(Ai, dtype=np.int32)
    m = gamma.size
    D, Lm = np.empty((m, 2)), np.empty((m, 2))
    lib().oracle_sweep_product(m, _p(R), _p(gamma), _p(Ai), _p(v), _p(D), _p(Lm), int(nthreads))
    return D, Lm


def sweep_geomean(R, w, gamma, Ai, v, nthreads=1):
    R, w, gamma, v = _f64(R), _f64(w), _f64(gamma), _f64(v)
    Ai = np.ascontiguousarray(Ai, dtype=np.int32)
    m = gamma.size
    D, Lm = np.empty((m, 2)), np.empty((m, 2))
    lib().oracle_sweep_geomean(m, _p(R), _p(w), _p(gamma), _p(Ai), _p(v), _p(D), _p(Lm), int(nthreads))
    return D, Lm


def univ3_current_ticks(current_price, tick_off, lower_ticks):
    current_price, lower_ticks = _f64(current_price), _f64(lower_ticks)
    tick_off = np.ascontiguousarray(tick_off, dtype=np.int64)
    m = current_price.size
    out = np.empty(m, dtype=np.int64)
    L = lib()
    base = lower_ticks.ctypes.data
    for i in range(m):
        o = int(tick_off[i])
        out[i] = L.oracle_univ3_current_tick(C.cast(base + 8 * o, _f64p), int(tick_off[i + 1]) - o,
                                             float(current_price[i]))
    return out


def sweep_univ3(current_price, current_tick, gamma, Ai, tick_off, lower_ticks, liquidity, v, nthreads=1):
    current_price, gamma, v = _f64(current_price), _f64(gamma), _f64(v)
    lower_ticks, liquidity = _f64(lower_ticks), _f64(liquidity)
    current_tick = np.ascontiguousarray(current_tick, dtype=np.int64)
    tick_off = np.ascontiguousarray(tick_off, dtype=np.int64)
    Ai = np.ascontiguousarray(Ai, dtype=np.int32)
    m = gamma.size
    D, Lm = np.empty((m, 2)), np.empty((m, 2))
    lib().oracle_sweep_univ3(m, _p(current_price), _p(current_tick), _p(gamma), _p(Ai), _p(tick_off),
                             _p(lower_ticks), _p(liquidity), _p(v), _p(D), _p(Lm), int(nthreads))
    return D, Lm


# ---- serial reductions ------------------------------------------------------------------------

def dual_acc(D, Lm, Ai, v):
    D, Lm, v = _f64(D), _f64(Lm), _f64(v)
    Ai = np.ascontiguousarray(Ai, dtype=np.int32)
    return lib().oracle_dual_acc(D.shape[0], _p(D), _p(Lm), _p(Ai), _p(v))


def grad_scatter(G, D, Lm, Ai):
    D, Lm = _f64(D), _f64(Lm)
    Ai = np.ascontiguousarray(Ai, dtype=np.int32)
    assert G.dtype == np.float64 and G.flags.c_contiguous
    lib().oracle_grad_scatter(D.shape[0], _p(D), _p(Lm), _p(Ai), _p(G))


def netflows(D, Lm, Ai, n_tokens):
    D, Lm = _f64(D), _f64(Lm)
    Ai = np.ascontiguousarray(Ai, dtype=np.int32)
    psi = np.empty(int(n_tokens))
    lib().oracle_netflows(D.shape[0], _p(D), _p(Lm), _p(Ai), int(n_tokens), _p(psi))
    return psi


# ---- objectives -------------------------------------------------------------------------------

class LinearNonnegative:
    """src/objectives.jl:51-79"""

    def __init__(self, c):
        self.c = _f64(c)
        if not np.all(self.c > 0):
            raise ValueError("all elements must be strictly positive")
        self.n = self.c.size

    def f(self, v):
        return lib().oracle_linear_nonneg_f(_p(self.c), _p(_f64(v)), self.n)

    def grad(self, v):
        g = np.empty(self.n)
        lib().oracle_linear_nonneg_grad(_p(g), _p(self.c), _p(_f64(v)), self.n)
        return g

    def lower_limit(self):
        lo = np.empty(self.n)
        lib().oracle_linear_nonneg_lower(_p(lo), _p(self.c), self.n)
        return lo

    def upper_limit(self):
        return np.full(self.n, np.inf)


class BasketLiquidation:
    """src/objectives.jl:92-129; `i` is 0-based here."""

    def __init__(self, i, Din):
        self.Din = _f64(Din)
        self.n = self.Din.size
        if not (0 <= i < self.n):
            raise ValueError("Invalid index i")
        self.i = int(i)

    def f(self, v):
        return lib().oracle_basket_liq_f(self.i, _p(self.Din), _p(_f64(v)), self.n)

    def grad(self, v):
        g = np.empty(self.n)
        lib().oracle_basket_liq_grad(_p(g), self.i, _p(self.Din), _p(_f64(v)), self.n)
        return g

    def lower_limit(self):
        lo = np.empty(self.n)
        lib().oracle_basket_liq_lower(_p(lo), self.i, self.n)
        return lo

    def upper_limit(self):
        return np.full(self.n, np.inf)


def Swap(i, j, delta, n):
    """src/objectives.jl:142-146 (0-based i, j)."""
    Din = np.zeros(n)
    Din[j] = delta
    return BasketLiquidation(i, Din)


# ---- route! restated (src/router.jl:58-108) ----------------------------------------------------

class PoolSet:
    """Pools in ROUTER ORDER for the oracle: a list of homogeneous segments
    ("product", dict(R, gamma, Ai)) / ("geomean", dict(R, w, gamma, Ai)) /
    ("univ3", dict(current_price, gamma, Ai, tick_off, lower_ticks, liquidity)).
    Segments are swept one after the other and concatenated, which equals the reference's
    pool-index order when the router's cfmms vector is grouped by family."""

    def __init__(self, segments, n_tokens):
        self.segments = segments
        self.n_tokens = int(n_tokens)
        self.Ai = np.ascontiguousarray(
            np.concatenate([np.asarray(s[1]["Ai"], dtype=np.int32).reshape(-1, 2) for s in segments]),
            dtype=np.int32)
        for kind, s in segments:
            if kind == "univ3" and "current_tick" not in s:
                s["current_tick"] = univ3_current_ticks(s["current_price"], s["tick_off"], s["lower_ticks"])
        self.m = self.Ai.shape[0]

    def sweep_into(self, v, D, Lm, nthreads=1):
        """find_arb!(r, v) into the caller's [m, 2] arrays (the reference overwrites r.Δs / r.Λs in place,
        src/router.jl:40): no allocation, no concatenation -- bench.py's cpu_baseline times this."""
        v = _f64(v)
        L, lo = lib(), 0
        for kind, s in self.segments:
            m = np.asarray(s["gamma"]).size
            d, l = D[lo:lo + m], Lm[lo:lo + m]
            c = s.setdefault("_c", {})     # contiguous typed views of the inputs, converted once
            def arr(name, dt=np.float64):
                if name not in c:
                    c[name] = np.ascontiguousarray(s[name], dtype=dt)
                return c[name]
            if kind == "product":
                L.oracle_sweep_product(m, _p(arr("R")), _p(arr("gamma")), _p(arr("Ai", np.int32)), _p(v), _p(d), _p(l), int(nthreads))
            elif kind == "geomean":
                L.oracle_sweep_geomean(m, _p(arr("R")), _p(arr("w")), _p(arr("gamma")), _p(arr("Ai", np.int32)), _p(v), _p(d), _p(l),
                                       int(nthreads))
            else:
                L.oracle_sweep_univ3(m, _p(arr("current_price")), _p(arr("current_tick", np.int64)), _p(arr("gamma")),
                                     _p(arr("Ai", np.int32)), _p(arr("tick_off", np.int64)), _p(arr("lower_ticks")),
                                     _p(arr("liquidity")), _p(v), _p(d), _p(l), int(nthreads))
            lo += m
        return D, Lm

    def sweep(self, v, nthreads=1):
        Ds, Ls = [], []
        for kind, s in self.segments:
            if kind == "product":
                D, Lm = sweep_product(s["R"], s["gamma"], s["Ai"], v, nthreads)
            elif kind == "geomean":
                D, Lm = sweep_geomean(s["R"], s["w"], s["gamma"], s["Ai"], v, nthreads)
            elif kind == "univ3":
                D, Lm = sweep_univ3(s["current_price"], s["current_tick"], s["gamma"], s["Ai"],
                                    s["tick_off"], s["lower_ticks"], s["liquidity"], v, nthreads)
            else:
                raise ValueError(kind)
            Ds.append(D)
            Ls.append(Lm)
        return np.concatenate(Ds), np.concatenate(Ls)


BOXED_INF = 1e100   # "u = Inf with nbd = 2": see route_oracle


def route_oracle(objective, pools: PoolSet, v0=None, m=5, factr=1e1, pgtol=1e-5, maxfun=15000,
                 maxiter=15000, nthreads=1):
    """route!(r; v, m, factr, pgtol, maxfun, maxiter) -- src/router.jl:58-108.

    Returns dict(v, Delta, Lambda, psi, n_sweeps, info)."""
    from scipy.optimize import fmin_l_bfgs_b

    n = pools.n_tokens
    state = {"v": np.ones(n) / n if v0 is None else np.array(v0, dtype=np.float64), "sweeps": 0}  # :61-65

    def sweep(v):
        state["D"], state["L"] = pools.sweep(v, nthreads)
        state["sweeps"] += 1

    lo = objective.lower_limit()  # :67-70 (nbd=2 with u=Inf == SciPy's nbd=1)
    # The reference passes nbd = 2 for every variable with an INFINITE upper bound (src/router.jl:67-70); the
    # Fortran code then treats the problem as "boxed" and takes a unit first step.  SciPy maps an infinite
    # bound to "no bound" (nbd = 1, first step min(1/|d|, 1)), so the reference's call shape is reproduced
    # with a finite upper bound no iterate can reach.
    up = objective.upper_limit() if hasattr(objective, "upper_limit") else np.full(n, np.inf)
    bounds = [(lo[j], up[j] if np.isfinite(up[j]) else BOXED_INF) for j in range(n)]

    def fg(v):
        if not np.all(v == state["v"]):  # :74-77 / :92-95 (one sweep per evaluation)
            sweep(v)
            state["v"] = v.copy()
        acc = dual_acc(state["D"], state["L"], pools.Ai, v)  # :79-83
        fval = objective.f(v) + acc                          # :85
        G = objective.grad(v)                                # :90,96
        grad_scatter(G, state["D"], state["L"], pools.Ai)    # :98-100
        return fval, G

    sweep(state["v"])  # :104
    v, fmin, info = fmin_l_bfgs_b(fg, state["v"].copy(), bounds=bounds, m=m, factr=factr, pgtol=pgtol,
                                  maxfun=maxfun, maxiter=maxiter, iprint=-1)  # :105
    state["v"] = np.array(v)  # :106
    sweep(state["v"])         # :107
    psi = netflows(state["D"], state["L"], pools.Ai, n)
    return {"v": state["v"], "Delta": state["D"], "Lambda": state["L"], "psi": psi,
            "n_sweeps": state["sweeps"], "f": fmin, "info": info}
