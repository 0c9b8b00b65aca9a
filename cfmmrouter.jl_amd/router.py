"""Host-side mirror of the reference's Router / route! (src/router.jl).

    Router(objective, cfmms, n_tokens)      src/router.jl:4-36
    find_arb_(r, v)   (Julia find_arb!)     src/router.jl:38-42   -> one device sweep
    route_(r; ...)    (Julia route!)        src/router.jl:58-108
    netflows_(ψ, r), netflows(r)            src/router.jl:111-125
    update_reserves_(r)                     src/router.jl:127-132 (unimplemented upstream)

Julia's `!` cannot appear in a Python identifier; mutating verbs carry a trailing underscore.
`r.Δs` / `r.Λs` / `r.v` / `r.cfmms` / `r.objective` are the reference's fields.

What runs where: pools are packed once into the device pool store (type-partitioned segments,
SoA-of-pairs); every evaluation of the dual function is ONE call through the C ABI which sweeps
all pools on the GPU and returns Ψ (n_tokens doubles) and the dual scalar.  The outer L-BFGS-B
iteration stays on the host, as in the reference (LBFGSB.jl there, SciPy's translation of the same
L-BFGS-B 3.0 here); its cost is O(n_tokens) per step.
"""
from __future__ import annotations

import math

import numpy as np

from . import objectives as _obj
from ._lib import KIND_GEOMEAN, KIND_PRODUCT, KIND_UNIV3, ArgumentError, CFMMDeviceError, Context
from .cfmms import CFMM, PoolBatch, _upload


_BOXED_INF = 1e100   # stands in for the reference's u = Inf under nbd = 2 (see route_)


class DeviceBackend:
    """The pool store + sweeps of one GPU, behind the C ABI (include/cfmm_amd.h)."""

    def __init__(self, n_tokens, batches, device=0):
        self.ctx = Context(n_tokens, device)
        self.n_tokens = int(n_tokens)
        for b in batches:
            if len(b):
                _upload(self.ctx, b)

    def eval(self, v):
        """fn/g! evaluation without trade write-back -> (Ψ, acc)."""
        return self.ctx.eval(v)

    def find_arb(self, v):
        """find_arb!(r, v): trades materialised on the device -> (Ψ, acc)."""
        self.ctx.find_arb(v)
        return self.ctx.netflows(), self.ctx.dual_value()

    def trades(self, out=None):
        return self.ctx.trades(out)

    def reload(self, batches):
        """Replace the device pool store (used after update_reserves_)."""
        self.ctx.clear()
        for b in batches:
            if len(b):
                _upload(self.ctx, b)

    def close(self):
        peer = getattr(self, "peer", None)
        if peer is not None and hasattr(peer, "close"):
            peer.close()       # unmap / free the IPC peer buffers while the context still exists
            self.peer = None
        self.ctx.close()


def _segments_of(cfmms):
    """Pack a pool container into homogeneous batches.

    Returns (batches, order, host) where `order[k]` is the router index of the k-th packed pool (None when packing
    preserved the router order) and `host` lists the router indices of pools whose TYPE has no device kernel -- any
    other CFMM subclass with its own `find_arb_(Δ, Λ, v)` method, the reference's plugin seam (src/cfmms.jl:35,56:
    `find_arb!(Δ, Λ, cfmm::CFMM, v)` is dispatched on the pool's type, so a user's own CFMM subtype enters a Router
    by defining that one method).  They are evaluated on the host every evaluation (HostSegment)."""
    if isinstance(cfmms, PoolBatch):
        return [cfmms], None, []
    cfmms = list(cfmms)
    if cfmms and all(isinstance(c, PoolBatch) for c in cfmms):
        return cfmms, None, []
    for c in cfmms:
        if not isinstance(c, CFMM):
            raise ArgumentError("cfmms must hold CFMM objects or PoolBatch containers")
    batches, order = [], []
    for kind in (KIND_PRODUCT, KIND_GEOMEAN, KIND_UNIV3):
        idx = [i for i, c in enumerate(cfmms) if c.kind == kind]
        if idx:
            batches.append(PoolBatch.from_pools(kind, [cfmms[i] for i in idx]))
            order.extend(idx)
    host = [i for i, c in enumerate(cfmms) if c.kind not in (KIND_PRODUCT, KIND_GEOMEAN, KIND_UNIV3)]
    for i in host:
        if not callable(getattr(cfmms[i], "find_arb_", None)) or not hasattr(cfmms[i], "Ai"):
            raise ArgumentError(f"cfmms[{i}] ({type(cfmms[i]).__name__}): a pool type without a device kernel needs its own "
                                f"find_arb_(Δ, Λ, v) method and an `Ai` field (src/cfmms.jl:35,56)")
    order = np.array(order, dtype=np.int64)
    if not host and np.array_equal(order, np.arange(len(cfmms))):
        order = None
    return batches, order, host


class HostSegment:
    """Pools whose type has no device kernel: the CALLER's own `find_arb_(Δ, Λ, v_local)` (Julia: the user's
    `find_arb!(Δ, Λ, cfmm, v)` method, src/cfmms.jl:35) runs on the host at every evaluation, exactly as
    `find_arb!(r::Router, v)` would call it (src/router.jl:40: `find_arb!(r.Δs[i], r.Λs[i], r.cfmms[i], v[r.cfmms[i].Ai])`),
    and their part of Ψ (src/router.jl:98-100, :114-115) and of the dual scalar (:82) is added to what the device returns.
    Any number of coins per pool (`len(pool.Ai)`)."""

    def __init__(self, pools, n_tokens):
        self.pools = list(pools)
        self.n_tokens = int(n_tokens)
        self.Ai0 = []
        for c in self.pools:
            ai = np.asarray(c.Ai, dtype=np.int64).reshape(-1) - 1
            if ai.size == 0 or np.any(ai < 0) or np.any(ai >= n_tokens):
                raise ArgumentError(f"token index out of range 1:{n_tokens}")
            self.Ai0.append(ai)
        self.Δs = [np.zeros(ai.size) for ai in self.Ai0]      # zerotrade(c), src/router.jl:23-26
        self.Λs = [np.zeros(ai.size) for ai in self.Ai0]

        self._sΔ = [np.zeros(ai.size) for ai in self.Ai0]     # scratch trades of non-materialising evaluations
        self._sΛ = [np.zeros(ai.size) for ai in self.Ai0]

    def sweep(self, v, materialize=True):
        """materialize=False (a bare evaluation: the device half writes no trades either): the pools' find_arb_ runs into
        scratch vectors, so self.Δs / self.Λs stay those of the latest find_arb -- the same price vector as the device
        pools' trades (the reference overwrites ALL pools' trades in every evaluation; here NONE move between find_arb!s)."""
        psi, acc = np.zeros(self.n_tokens), 0.0
        Ds, Ls = (self.Δs, self.Λs) if materialize else (self._sΔ, self._sΛ)
        for c, ai, D, L in zip(self.pools, self.Ai0, Ds, Ls):
            vl = v[ai]                                   # v[r.cfmms[i].Ai]
            c.find_arb_(D, L, vl)                        # the user's method: overwrites Δ, Λ
            acc += float(np.dot(L, vl) - np.dot(D, vl))  # src/router.jl:82
            np.add.at(psi, ai, L - D)                    # src/router.jl:99
        return psi, acc


class MixedBackend:
    """A device (or any other) backend plus a HostSegment: one evaluation = the backend's sweep + the host pools' own
    find_arb_, their Ψ and dual parts summed -- what L-BFGS-B sees is the dual of the WHOLE router."""

    def __init__(self, inner, host: HostSegment):
        self.inner, self.host = inner, host
        self.n_tokens = host.n_tokens

    def _add(self, res, v, materialize):
        psi, acc = res
        ph, ah = self.host.sweep(np.asarray(v, dtype=np.float64), materialize)
        return psi + ph, acc + ah

    def eval(self, v):
        return self._add(self.inner.eval(v), v, False)

    def find_arb(self, v):
        return self._add(self.inner.find_arb(v), v, True)

    def trades(self, out=None):
        return self.inner.trades() if out is None else self.inner.trades(out)

    def reload(self, batches):
        self.inner.reload(batches)

    ctx = property(lambda self: self.inner.ctx)     # library options / introspection of the device half

    def close(self):
        if hasattr(self.inner, "close"):
            self.inner.close()


class _PoolView:
    """`r.cfmms` for batch-built routers: indexable / iterable pool objects, built on demand."""

    def __init__(self, batches):
        self._batches = batches
        self._offsets = np.cumsum([0] + [len(b) for b in batches])

    def __len__(self):
        return int(self._offsets[-1])

    def __getitem__(self, i):
        i = int(i)
        if i < 0:
            i += len(self)
        s = int(np.searchsorted(self._offsets, i, side="right") - 1)
        return self._batches[s][i - int(self._offsets[s])]

    def __iter__(self):
        for b in self._batches:
            for i in range(len(b)):
                yield b[i]


class Router:
    """Router(objective, cfmms, n_tokens) -- src/router.jl:4-36."""

    def __init__(self, objective, cfmms, n_tokens, device=0, _backend=None):
        if not isinstance(objective, _obj.Objective):
            raise ArgumentError("objective must be an Objective")
        self.objective = objective
        self.n_tokens = int(n_tokens)
        if not isinstance(cfmms, PoolBatch):
            cfmms = list(cfmms)
        batches, self._order, host = _segments_of(cfmms)
        self._batches = batches
        from_batches = isinstance(cfmms, PoolBatch) or (
            len(batches) > 0 and not isinstance(cfmms, PoolBatch) and all(isinstance(c, PoolBatch) for c in cfmms))
        self.cfmms = _PoolView(batches) if from_batches else list(cfmms)
        self._m = sum(len(b) for b in batches)       # pools with a device kernel
        self.v = np.zeros(self.n_tokens)  # :33
        self._backend = _backend if _backend is not None else DeviceBackend(self.n_tokens, batches, device)
        # the plugin seam: pools of any other CFMM subclass are evaluated by their own find_arb_ on the host
        self._host = HostSegment([cfmms[i] for i in host], self.n_tokens) if host else None
        self._host_idx = list(host)
        if self._host is not None:
            self._backend = MixedBackend(self._backend, self._host)
        self._psi = np.zeros(self.n_tokens)
        self._acc = 0.0
        self._Δs = np.zeros((self._m, 2))  # zerotrade per pool, :23-26
        self._Λs = np.zeros((self._m, 2))
        self._trades_stale = False
        self.n_sweeps = 0
        self.info = None

    # r.Δs / r.Λs: [m, 2] arrays in router order (rows are the reference's per-pool vectors); routers with host-evaluated
    # pools: a list of per-pool vectors in router order (the reference's Vector{Vector}), host pools' vectors included
    def _fetch(self):
        if self._trades_stale:
            if self._host is not None:        # packed order on the device; _rows() maps router index -> row
                self._Δs, self._Λs = self._backend.trades()
            elif self._order is not None:
                D, Lm = self._backend.trades()
                self._Δs[self._order] = D
                self._Λs[self._order] = Lm
            else:           # find_arb! overwrites r.Δs / r.Λs in place (src/router.jl:40): the router's own arrays are filled
                if isinstance(self._backend, DeviceBackend):
                    self._backend.trades(out=(self._Δs, self._Λs))
                else:                   # test-injected backends
                    self._Δs, self._Λs = self._backend.trades()
            self._trades_stale = False

    def _rows(self, dev, host):
        out = [None] * (self._m + len(self._host_idx))
        for k, i in enumerate(self._order if self._order is not None else range(self._m)):
            out[int(i)] = dev[k]
        for j, i in enumerate(self._host_idx):
            out[i] = host[j]
        return out

    @property
    def Δs(self):
        self._fetch()
        return self._Δs if self._host is None else self._rows(self._Δs, self._host.Δs)

    @property
    def Λs(self):
        self._fetch()
        return self._Λs if self._host is None else self._rows(self._Λs, self._host.Λs)

    Deltas = Δs
    Lambdas = Λs

    def close(self):
        if hasattr(self._backend, "close"):
            self._backend.close()


def find_arb_(r: Router, v):
    """find_arb!(r::Router, v) -- src/router.jl:38-42: every pool's arbitrage at prices v."""
    v = np.asarray(v, dtype=np.float64)
    r._psi, r._acc = r._backend.find_arb(v)
    r._trades_stale = True
    r.n_sweeps += 1
    return None


def route_(r: Router, v=None, verbose=False, m=5, factr=1e1, pgtol=1e-5, maxfun=15_000, maxiter=15_000,
           solver="scipy"):
    """route!(r; v, verbose, m, factr, pgtol, maxfun, maxiter) -- src/router.jl:58-108.

    Overwrites r.Δs, r.Λs and r.v.  The dual function g(ν) = f(ν) + Σᵢ arbᵢ(Aᵢᵀν) is minimised
    with L-BFGS-B over the objective's box; each evaluation is one device sweep.

    solver="scipy"  (default): SciPy's translation of the Fortran L-BFGS-B 3.0 the reference calls
                    through LBFGSB.jl drives the loop from Python, one C-ABI call per evaluation.
    solver="native": the library's own L-BFGS-B (csrc/lbfgsb.cpp).  On a single-GPU router the whole
                    of route! then runs inside ONE C-ABI call (cfmm_route): no interpreter between
                    two device sweeps."""
    if solver == "native":
        return _route_native(r, v, m, factr, pgtol, maxfun, maxiter)
    if solver != "scipy":
        raise ArgumentError("solver must be 'scipy' or 'native'")
    from scipy.optimize import fmin_l_bfgs_b

    n = r.v.size
    if v is None:
        r.v[:] = np.ones(n) / n  # :62
    else:
        r.v[:] = v  # :64
    lo = _obj.lower_limit(r.objective)  # :67-70 (nbd = 2 with an infinite upper bound)
    up = _obj.upper_limit(r.objective)
    # nbd = 2 with u = Inf makes the Fortran code take its "boxed" unit first step; SciPy would turn an
    # infinite bound into "no bound" (first step min(1/|d|, 1)), so the reference's call shape is kept
    # with a finite upper bound no iterate can reach (solver="native" emulates the same thing directly).
    bounds = [(lo[j], _BOXED_INF if math.isinf(up[j]) else up[j]) for j in range(n)]

    def sweep(x):
        r._psi, r._acc = r._backend.eval(x)
        r.n_sweeps += 1

    def fg(x):
        if not np.all(x == r.v):  # :74-77 / :92-95: one sweep per evaluation
            sweep(x)
            r.v[:] = x
        fval = _obj.f(r.objective, x) + r._acc  # :79-85
        G = np.zeros(n)  # :90
        _obj.grad_(G, r.objective, x)  # :96
        G += r._psi  # :98-100
        return fval, G

    sweep(r.v)  # :104
    kw = dict(bounds=bounds, m=m, factr=factr, pgtol=pgtol, maxfun=maxfun, maxiter=maxiter)
    if verbose:
        kw["iprint"] = 1
    x, fmin, info = fmin_l_bfgs_b(fg, r.v.copy(), **kw)  # :105
    r.v[:] = x  # :106
    r.info = {"f": fmin, **{k: info[k] for k in ("warnflag", "task", "funcalls", "nit") if k in info}}
    find_arb_(r, r.v)  # :107
    return None


def _route_native(r: Router, v, m, factr, pgtol, maxfun, maxiter):
    from ._lib import OBJ_BASKET_LIQUIDATION, OBJ_LINEAR_NONNEGATIVE, lbfgsb_minimize

    n = r.v.size
    obj = r.objective
    if isinstance(r._backend, DeviceBackend):   # everything in one library call
        if isinstance(obj, _obj.LinearNonnegative):
            kind, vec, idx = OBJ_LINEAR_NONNEGATIVE, obj.c, 0
        elif isinstance(obj, _obj.BasketLiquidation):
            kind, vec, idx = OBJ_BASKET_LIQUIDATION, obj.Δin, obj.i - 1
        else:
            raise ArgumentError("solver='native' knows LinearNonnegative and BasketLiquidation objectives")
        ctx, guard = r._backend.ctx, getattr(r, "_guard", None)
        call = lambda: ctx.route(kind, vec, idx, v0=v, m=m, factr=factr, pgtol=pgtol, maxfun=maxfun, maxiter=maxiter)
        if guard is None:
            vout, psi, info = call()
        else:
            # sharded (cfmm_set_peers): the ranks run this call in lockstep.  If ANY rank's call fails -- a lost pre-armed
            # hand-over, a peer that did not publish within the time limit -- every rank learns it through one vote,
            # the exchange is re-aligned, pre-arming goes off and the route is repeated launch-when-ready on all ranks.
            res = err = None
            try:
                res = call()
            except RuntimeError as e:      # CFMM_ERR_STATE (RuntimeError) or a HIP failure (CFMMDeviceError, its subclass)
                err = e
            if not guard.vote(res is not None):
                guard.resync()
                ctx.set_option("armed", 0)
                res = err = None
                try:
                    res = call()
                except RuntimeError as e:
                    err = e
                if not guard.vote(res is not None):
                    raise err if err is not None else CFMMDeviceError("sharded route! failed on another rank")
                r.collective_retries = getattr(r, "collective_retries", 0) + 1
            vout, psi, info = res
        r.v[:] = vout
        r._psi, r._acc = psi, r._backend.ctx.dual_value()
        r._trades_stale = True
        r.n_sweeps += info["sweeps"]
        r.info = {"f": info["f"], "funcalls": info["evaluations"], "nit": info["iterations"],
                  "warnflag": 0 if info["status"] in (0, 1) else 2, "task": info["status"], "solver": "native",
                  "sweep_seconds": info["sweep_seconds"], "total_seconds": info["total_seconds"]}
        return None
    # any other backend (sharded, test-injected): same solver, Python callback per evaluation
    r.v[:] = np.ones(n) / n if v is None else v
    lo, up = _obj.lower_limit(r.objective), _obj.upper_limit(r.objective)
    bounds = [(lo[j], None if math.isinf(up[j]) else up[j]) for j in range(n)]

    def sweep(x):
        r._psi, r._acc = r._backend.eval(x)
        r.n_sweeps += 1

    def fg(x):
        if not np.all(x == r.v):
            sweep(x)
            r.v[:] = x
        G = np.zeros(n)
        _obj.grad_(G, r.objective, x)
        return _obj.f(r.objective, x) + r._acc, G + r._psi

    sweep(r.v)
    x, info = lbfgsb_minimize(fg, r.v.copy(), bounds, m=m, factr=factr, pgtol=pgtol, maxfun=maxfun, maxiter=maxiter,
                              reference_boxed=True)   # nbd = 2 with u = Inf, as src/router.jl:67-70
    r.v[:] = x
    r.info = {"f": info["f"], "funcalls": info["evaluations"], "nit": info["iterations"],
              "warnflag": 0 if info["status"] in (0, 1) else 2, "task": info["status"], "solver": "native"}
    find_arb_(r, r.v)
    return None


def _dual_gradient(r: Router, x):
    """G(ν) = ∇f(obj, ν) + Ψ(ν) of g! (src/router.jl:89-102) at ν = x: one fused sweep."""
    psi, acc = r._backend.eval(x)
    r.n_sweeps += 1
    G = np.zeros(x.size)
    _obj.grad_(G, r.objective, x)
    return G + psi, psi, acc


def dual_jacobian(r: Router, v=None, rel_step=1e-7):
    """Forward-difference Jacobian of the dual gradient G(ν) = ∇f(obj, ν) + Ψ(ν) (src/router.jl:89-102) at ν = v
    (default r.v): column j = (G(ν + hⱼeⱼ) − G(ν)) / hⱼ with hⱼ = rel_step·νⱼ -- n_tokens + 1 fused sweeps.  It is the
    Hessian of the dual where that exists (the dual is C¹ and piecewise C²: pools enter and leave their no-trade band).
    `polish_` uses it as the matrix of a chord-Newton iteration, where only its rough accuracy matters."""
    x = np.array(r.v if v is None else v, dtype=np.float64)
    G0, _, _ = _dual_gradient(r, x)
    J = np.empty((x.size, x.size))
    for j in range(x.size):
        xj = x.copy()
        xj[j] = x[j] * (1.0 + rel_step)
        J[:, j] = (_dual_gradient(r, xj)[0] - G0) / (xj[j] - x[j])
    return J


def polish_(r: Router, iters=8, jacobian=None, rel_step=1e-7, native=None):
    """Tighten a route!'s result beyond what L-BFGS-B's stopping rules can: a projected chord-Newton iteration on the
    optimality conditions of the dual problem route! solves (src/router.jl:58-108),

        Gⱼ(ν) = 0 for lⱼ < νⱼ,    Gⱼ(ν) ≥ 0 for νⱼ = lⱼ,    G(ν) = ∇f(obj, ν) + Ψ(ν),  l = lower_limit(obj),

    starting from r.v.  NOT part of the reference (its route! ends where L-BFGS-B ends); it exists because L-BFGS-B's
    line search works on the dual VALUE, whose rounding noise (a sum over 10⁶ pools) hides decreases below ~1e-15
    relative -- on interior optima that leaves a stationarity residual of ~1e-6·max|Ψ| -- while the GRADIENT Ψ is
    resolved to ~1e-12·max|Ψ|.  The iteration uses gradients only:

        free set F = {j : not (νⱼ = lⱼ and Gⱼ > 0)},   ν_F ← max(l_F, ν_F − J_FF⁻¹ G_F),

    J = `jacobian` (any reasonable approximation of the dual's Hessian: the limit point is where THIS backend's G
    satisfies the conditions, whatever J is; only the rate depends on it) or `dual_jacobian(r)` at the start
    (n_tokens + 1 sweeps).  Steps that do not reduce the residual are halved; the best iterate is kept.  Ends with
    find_arb!(r, ν) like route! does (:106-107), so r.v / r.Δs / r.Λs / netflows(r) describe the polished point.
    r.info["polish"] = {"residual0", "residual", "iterations", "sweeps"} (residual = max |G_F|, and −Gⱼ where a
    variable on its bound has Gⱼ < 0).

    native (default: True on an unsharded single-device router when no `jacobian` is given): the same iteration inside
    the library (cfmm_polish, one C-ABI call: what a Julia or C caller uses)."""
    if native is None:
        native = jacobian is None and isinstance(r._backend, DeviceBackend) and getattr(r, "_guard", None) is None
    if native:
        if jacobian is not None or not isinstance(r._backend, DeviceBackend):
            raise ArgumentError("native polish computes its own Jacobian on a DeviceBackend")
        from ._lib import OBJ_BASKET_LIQUIDATION, OBJ_LINEAR_NONNEGATIVE
        obj = r.objective
        if isinstance(obj, _obj.LinearNonnegative):
            kind, vec, idx = OBJ_LINEAR_NONNEGATIVE, obj.c, 0
        else:
            kind, vec, idx = OBJ_BASKET_LIQUIDATION, obj.Δin, obj.i - 1
        vout, psi, pinfo = r._backend.ctx.polish(kind, vec, idx, r.v, max_iters=iters, rel_step=rel_step)
        r.v[:] = vout
        r._psi, r._acc = psi, r._backend.ctx.dual_value()
        r._trades_stale = True
        r.n_sweeps += pinfo["sweeps"]
        info = dict(r.info) if isinstance(r.info, dict) else {}
        info["polish"] = {k: pinfo[k] for k in ("residual0", "residual", "iterations", "sweeps")}
        info["polish"]["native_seconds"] = pinfo["total_seconds"]
        r.info = info
        return None
    n = r.v.size
    lo = _obj.lower_limit(r.objective)
    sweeps0 = r.n_sweeps
    x = np.maximum(np.array(r.v, dtype=np.float64), lo)
    J = dual_jacobian(r, x, rel_step) if jacobian is None else np.asarray(jacobian, dtype=np.float64)

    def residual(x, G):
        free = ~((x <= lo) & (G > 0.0))
        return free, (float(np.max(np.abs(G[free]))) if free.any() else 0.0)

    G, _, _ = _dual_gradient(r, x)
    free, res = residual(x, G)
    res0, done = res, 0
    for _ in range(int(iters)):
        if res == 0.0:
            break
        step = np.zeros(n)
        step[free] = np.linalg.lstsq(J[np.ix_(free, free)], -G[free], rcond=1e-13)[0]
        t, improved = 1.0, False
        while t >= 1.0 / 64:
            xt = np.maximum(x + t * step, lo)
            Gt, _, _ = _dual_gradient(r, xt)
            freet, rest = residual(xt, Gt)
            if rest < res:
                x, G, free, res, improved = xt, Gt, freet, rest, True
                break
            t *= 0.5
        done += 1
        if not improved:   # the rounding-noise floor of Ψ (or a kink the chord matrix cannot cross): keep the best iterate
            break
    r.v[:] = x
    find_arb_(r, r.v)
    info = dict(r.info) if isinstance(r.info, dict) else {}
    info["polish"] = {"residual0": res0, "residual": res, "iterations": done, "sweeps": r.n_sweeps - sweeps0}
    r.info = info
    return None


def netflows_(ψ, r: Router, exact=False):
    """netflows!(ψ, r) -- src/router.jl:111-119: ψ = Σᵢ Aᵢ(Λᵢ − Δᵢ) for the latest sweep.

    exact=False (default): the device's reduction of the sweep that produced r.Δs / r.Λs (per-wavefront LDS bins, fixed-order
    folds): within 1e-12·max|ψ| of the reference's value, no per-pool data moved.
    exact=True: the reference's OWN loop -- `ψ[c.Ai] .+= Λ - Δ` pool after pool in router order (src/router.jl:113-116) --
    over the fetched trade rows, i.e. bit for bit what `netflows(r)` returns upstream, so that the reference's router test
    `all_flows .== netflows(r)` (test/arb.jl:16) holds unedited.  O(m) on the host (as upstream), and it fetches r.Δs / r.Λs."""
    if not exact:
        ψ[:] = r._psi
        return None
    Δs, Λs = r.Δs, r.Λs
    ψ[:] = 0.0
    if r._host is not None:                      # per-pool vectors of any length, router order
        for Δ, Λ, c in zip(Δs, Λs, r.cfmms):
            ai = np.asarray(c.Ai, dtype=np.int64).reshape(-1) - 1
            for k in range(ai.size):             # broadcast assignment, element after element
                ψ[ai[k]] += Λ[k] - Δ[k]
        return None
    Ai = np.concatenate([b.Ai for b in r._batches]) if r._batches else np.zeros((0, 2), dtype=np.int64)
    if r._order is not None:                     # packed (family-grouped) order -> router order
        Ar = np.empty_like(Ai)
        Ar[r._order] = Ai
        Ai = Ar
    # np.bincount adds its weights to each bin one after another in input order: pool 1 coin 1, pool 1 coin 2, pool 2 coin 1, ...
    # -- the reference's loop, starting from zeros
    ψ[:] = np.bincount((Ai.astype(np.int64) - 1).ravel(), weights=(Λs - Δs).ravel(), minlength=r.n_tokens)[:r.n_tokens]
    return None


def netflows(r: Router, exact=False):
    """netflows(r) -- src/router.jl:121-125 (exact: see netflows_)"""
    ψ = np.zeros_like(r.v)
    netflows_(ψ, r, exact)
    return ψ


def update_reserves_(r: Router, sync_host=True):
    """update_reserves!(r) -- src/router.jl:127-132.

    The reference's router method calls `update_reserves!(c, Δ, Λ, v)` per pool, a method that is
    defined nowhere (its own test marks it "borked", test/arb.jl:30-39), so there is no reference
    behaviour to match.  What is implemented is the update the routing problem itself prescribes
    (find_arb! docstring, src/cfmms.jl:26-31: the pool ends at R + γΔ − Λ), applied in place on the
    device from the trades of the latest find_arb!/route! (cfmm_update_reserves): one kernel for the
    two-coin families; UniV3 pools move to the price the arbitrage left them at.  The trades are
    consumed.  sync_host=True (default) also refreshes the host mirror (`r.cfmms[i].R`, batch.R,
    batch.current_price: 16 / 8 bytes per pool device-to-host); sequential routing that never reads
    them can pass sync_host=False and moves no per-pool data at all."""
    if r._host is not None:
        # host-evaluated pools: the pool type's own update_reserves_(Δ, Λ, v) -- the per-pool method the reference's
        # router calls (src/router.jl:129: update_reserves!(c, Δ, Λ, r.v[c.Ai])) -- then zero trades
        for c in r._host.pools:
            if not callable(getattr(c, "update_reserves_", None)):
                raise ArgumentError(f"{type(c).__name__} has no update_reserves_(Δ, Λ, v) method (src/router.jl:129)")

    def host_pools():
        # AFTER the device half (which can still refuse: no trades, a UniV3 segment without host prices) -- a refused
        # update then leaves every pool, host-evaluated ones included, as it was
        if r._host is None:
            return
        for c, ai, D, L in zip(r._host.pools, r._host.Ai0, r._host.Δs, r._host.Λs):
            c.update_reserves_(D, L, r.v[ai])
            D[:] = 0.0
            L[:] = 0.0

    ctx = getattr(r._backend, "ctx", None)
    if ctx is None:   # test-injected / sharded backends: host-side update of the two-coin families
        _update_reserves_host(r)
        return host_pools()
    ctx.update_reserves()
    host_pools()
    if sync_host:
        seg = 0
        for b in r._batches:
            if len(b) == 0:
                continue
            if b.kind == KIND_UNIV3:
                b.current_price[:] = ctx.prices(seg, len(b))
            else:
                b.R[:] = ctx.reserves(seg, len(b))
            seg += 1
        if isinstance(r.cfmms, list):                     # keep the per-pool objects in step
            it = iter(range(r._m)) if r._order is None else iter(r._order)
            for b in r._batches:
                for k in range(len(b)):
                    pool = r.cfmms[next(it)]
                    if b.kind == KIND_UNIV3:
                        pool.current_price = float(b.current_price[k])
                        pool.current_tick = int(np.count_nonzero(pool.lower_ticks >= pool.current_price))
                    else:
                        pool.R[:] = b.R[k]
    r._Δs = np.zeros((r._m, 2))
    r._Λs = np.zeros((r._m, 2))
    r._psi = np.zeros(r.n_tokens)
    r._acc = 0.0
    r._trades_stale = False
    return None


def _update_reserves_host(r: Router):
    if any(b.kind == KIND_UNIV3 for b in r._batches):
        raise NotImplementedError("update_reserves! is not defined for UniV3 pools (nor in the reference)")
    if not hasattr(getattr(r._backend, "inner", r._backend), "reload"):    # (MixedBackend forwards to its inner backend)
        raise NotImplementedError("this backend cannot reload pools")
    D, Lm = r._backend.trades()                       # packed (segment) order
    off = 0
    for b in r._batches:
        m = len(b)
        b.R[:] = b.R + b.γ[:, None] * D[off:off + m] - Lm[off:off + m]
        off += m
    r._backend.reload(r._batches)
    if isinstance(r.cfmms, list):                     # keep the per-pool objects in step
        it = iter(range(r._m)) if r._order is None else iter(r._order)
        for b in r._batches:
            for k in range(len(b)):
                r.cfmms[next(it)].R[:] = b.R[k]
    r._Δs = np.zeros((r._m, 2))
    r._Λs = np.zeros((r._m, 2))
    r._psi = np.zeros(r.n_tokens)
    r._acc = 0.0
    r._trades_stale = False
    return None
