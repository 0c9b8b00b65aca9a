"""Shared test helpers: bridge between the host mirror's PoolBatch containers and the CPU oracle."""
import numpy as np

import cfmmrouter_amd as cr
from cfmmrouter_amd._lib import KIND_GEOMEAN, KIND_PRODUCT, KIND_UNIV3
from oracle import cfmm_oracle as orc


def oracle_poolset(batches, n_tokens):
    segs = []
    for b in batches:
        Ai0 = (b.Ai - 1).astype(np.int32)
        if b.kind == KIND_PRODUCT:
            segs.append(("product", dict(R=b.R, gamma=b.γ, Ai=Ai0)))
        elif b.kind == KIND_GEOMEAN:
            segs.append(("geomean", dict(R=b.R, w=b.w, gamma=b.γ, Ai=Ai0)))
        else:
            segs.append(("univ3", dict(current_price=b.current_price, gamma=b.γ, Ai=Ai0, tick_off=b.tick_off,
                                       lower_ticks=b.lower_ticks, liquidity=b.liquidity)))
    return orc.PoolSet(segs, n_tokens)


def oracle_objective(obj):
    if isinstance(obj, cr.LinearNonnegative):
        return orc.LinearNonnegative(obj.c)
    return orc.BasketLiquidation(obj.i - 1, obj.Δin)


def oracle_sweep(batches, n_tokens, v, nthreads=1):
    ps = oracle_poolset(batches, n_tokens)
    D, L = ps.sweep(np.asarray(v, dtype=np.float64), nthreads)
    psi = orc.netflows(D, L, ps.Ai, n_tokens)
    acc = orc.dual_acc(D, L, ps.Ai, v)
    return D, L, psi, acc


def rel_to_max(a, b):
    """max|a-b| / max|b| : the netflow parity measure (component-wise relative error is
    unattainable on constraint-active ~0 components)."""
    scale = max(np.max(np.abs(b)), np.finfo(float).tiny)
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b))) / scale)


class OracleBackend:
    """TEST-ONLY stand-in for DeviceBackend so the host logic (route loop, packing order, sharded
    reduction) can be exercised on a box with no GPU.  Lives in tests/, never in the package."""

    def __init__(self, n_tokens, batches, nthreads=1):
        self.ps = oracle_poolset(batches, n_tokens)
        self.n_tokens = n_tokens
        self.nthreads = nthreads

    def eval(self, v):
        self.D, self.L = self.ps.sweep(np.asarray(v, dtype=np.float64), self.nthreads)
        return orc.netflows(self.D, self.L, self.ps.Ai, self.n_tokens), orc.dual_acc(self.D, self.L, self.ps.Ai, v)

    find_arb = eval

    def trades(self):
        return self.D, self.L

    def reload(self, batches):
        self.ps = oracle_poolset(batches, self.n_tokens)


def route_converged(obj, market, n, v0=None, nthreads=1, solver="native", router=None):
    """Route-level parity BY CONVERGENCE.  Device side: route! (`solver`) on the HIP path, then
    polish_ with the device's own finite-difference Jacobian.  CPU side: route_oracle (SciPy L-BFGS-B on the CPU
    restatement), then the SAME polish_ code driven through the OracleBackend -- with the device's Jacobian as the chord
    matrix, which only sets the rate: the point it converges to is where the RESTATEMENT's gradient satisfies the
    optimality conditions (tests/test_polish_cpu.py::test_polish_with_a_foreign_chord_matrix...).  Returns the default-
    tolerance distance, the converged distance (both max|ΔΨ| / max|Ψ|) and the polish records."""
    r = router if router is not None else cr.Router(obj, market, n)
    try:
        cr.route_(r, v=v0, solver=solver)
        psi_default, v_default, evals = cr.netflows(r).copy(), r.v.copy(), r.info["funcalls"]
        J = cr.dual_jacobian(r)
        cr.polish_(r, jacobian=J)
        psi_dev, v_dev, pol_dev = cr.netflows(r).copy(), r.v.copy(), dict(r.info["polish"])
    finally:
        if router is None:
            r.close()
    ref = orc.route_oracle(oracle_objective(obj), oracle_poolset(market, n), v0=v0, nthreads=nthreads)
    ro = cr.Router(obj, market, n, _backend=OracleBackend(n, market, nthreads))
    ro.v[:] = ref["v"]
    cr.polish_(ro, jacobian=J)
    psi_ref = cr.netflows(ro)
    return {"default": rel_to_max(psi_default, ref["psi"]), "converged": rel_to_max(psi_dev, psi_ref),
            "v_converged": float(np.max(np.abs(v_dev - ro.v) / ro.v)),
            "device_moved": rel_to_max(psi_default, psi_dev), "oracle_moved": rel_to_max(ref["psi"], psi_ref),
            "evaluations_device": int(evals), "evaluations_oracle": int(ref["info"]["funcalls"]),
            "polish_device": pol_dev, "polish_oracle": dict(ro.info["polish"]), "psi_scale": float(np.max(np.abs(psi_ref)))}
