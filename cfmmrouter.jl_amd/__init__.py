"""cfmmrouter.jl_amd -- MI355X-native drop-in for CFMMRouter.jl's arbitrage-sweep hot path.

Exports the reference's names (src/CFMMRouter.jl, src/cfmms.jl:1-3, src/objectives.jl:1,
src/router.jl:1-2).  Julia's `f!` is spelled `f_` here.  All pool arithmetic runs on the GPU
through libcfmm_amd.so (include/cfmm_amd.h); there is no CPU fallback in this package.
"""
from ._lib import ArgumentError, CFMMDeviceError, Context, build, lib
from .cfmms import (CFMM, BoundedProduct, GeometricMeanTwoCoin, PoolBatch, ProductTwoCoin, UniV3, find_arb_ as _find_arb_pool,
                    grad_phi_, phi, zerotrade, ϕ, ϕ_grad_)
from .objectives import (BasketLiquidation, LinearNonnegative, Objective, Swap, f, grad_, lower_limit,
                         upper_limit)
from .router import (DeviceBackend, Router, dual_jacobian, find_arb_ as _find_arb_router, netflows, netflows_, polish_, route_,
                     update_reserves_)


def find_arb_(*args, **kw):
    """find_arb!(r::Router, v)  or  find_arb!(Δ, Λ, cfmm, v)  (multiple dispatch on arity)."""
    if len(args) == 2:
        return _find_arb_router(*args, **kw)
    return _find_arb_pool(*args, **kw)


__all__ = [
    "CFMM", "ProductTwoCoin", "GeometricMeanTwoCoin", "UniV3", "BoundedProduct", "PoolBatch", "find_arb_",
    "update_reserves_", "Objective", "LinearNonnegative", "BasketLiquidation", "Swap", "f", "grad_",
    "lower_limit", "upper_limit", "Router", "route_", "netflows_", "netflows", "ArgumentError",
    "CFMMDeviceError", "Context", "DeviceBackend", "build", "lib", "zerotrade", "ϕ", "ϕ_grad_", "phi", "grad_phi_",
    "polish_", "dual_jacobian",
]
