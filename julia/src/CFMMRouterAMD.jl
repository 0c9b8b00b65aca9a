# CFMMRouterAMD.jl -- the Julia host side of the MI355X drop-in: a `Router`-compatible type whose
# `find_arb!` sweep and Ψ/dual reductions run in libcfmm_amd.so (include/cfmm_amd.h) while
# LBFGSB.jl keeps driving the outer loop exactly as in CFMMRouter.jl's src/router.jl:58-108.
#
# STATUS: EXPERIMENTAL -- written against Julia 1.7+/CFMMRouter v0.3.1 but NEVER EXECUTED: there is no
# Julia toolchain in the build image.  Every ccall below has a twin that IS executed on the MI355X:
# the plain-C clients tests/c/abi_smoke.c and tests/c/abi_multi.c (same entry points, same argument
# order) and the Python mirror (cfmmrouter.jl_amd/router.py, tests/test_gpu_*.py).  The one-command check for whoever has
# Julia and an MI355X:   CFMM_AMD_LIB=$PWD/cfmmrouter.jl_amd/libcfmm_amd.so julia --project=julia -e 'using Pkg; Pkg.test()'
# (julia/test/runtests.jl: the reference's own router / CFMM tests through AMDRouter).
#
# Usage (drop-in for the README quick start):
#     using CFMMRouter, CFMMRouterAMD
#     router = AMDRouter(LinearNonnegative(prices), [equal_pool, unequal_small_pool], 2)
#     route!(router); Ψ = netflows(router)
module CFMMRouterAMD

using CFMMRouter
using CFMMRouter: CFMM, ProductTwoCoin, GeometricMeanTwoCoin, UniV3, Objective
using LBFGSB
import CFMMRouter: route!, netflows, netflows!, find_arb!, update_reserves!

export AMDRouter, route_native!, polish!

const LIB = get(ENV, "CFMM_AMD_LIB", "libcfmm_amd.so")

struct CFMMAMDError <: Exception
    code::Cint
    msg::String
end

function check(ctx::Ptr{Cvoid}, rc::Cint)
    rc == 0 && return nothing
    msg = unsafe_string(ccall((:cfmm_last_error, LIB), Cstring, (Ptr{Cvoid},), ctx))
    rc == -1 && throw(ArgumentError(msg))          # CFMM_ERR_INVALID_ARG == the reference's ArgumentError
    throw(CFMMAMDError(rc, msg))
end

# Same readable fields as CFMMRouter.Router (src/router.jl:4-10): objective, cfmms, Δs, Λs, v.
mutable struct AMDRouter{O,T}
    objective::O
    cfmms::Vector{CFMM{T}}
    Δs::Vector{Vector{T}}
    Λs::Vector{Vector{T}}
    v::Vector{T}
    ctx::Ptr{Cvoid}
    order::Vector{Int}          # packed position -> index into cfmms (pools are grouped by family)
    Ψ::Vector{T}
    acc::Base.RefValue{T}
    host::Vector{Int}           # indices of pools whose TYPE has no device kernel: evaluated by their own find_arb! on the host
end

# Router(objective, cfmms, n_tokens) -- src/router.jl:18-36
# `device` is a HIP ordinal, or a vector of ordinals: the pools are then split in contiguous blocks over
# those GPUs from this one Julia task (cfmm_ctx_create_multi: one L-BFGS-B drives all shards, the
# shards' Ψ are summed on the host in device order; no MPI, no RCCL, nothing else to set up).
function AMDRouter(objective::O, cfmms::Vector{C}, n_tokens; device=0) where {O<:Objective,C<:CFMM{Float64}}
    ctxref = Ref{Ptr{Cvoid}}(C_NULL)
    rc = if device isa Integer
        ccall((:cfmm_ctx_create, LIB), Cint, (Cint, Int32, Ref{Ptr{Cvoid}}), device, n_tokens, ctxref)
    else
        ids = Int32.(collect(device))
        GC.@preserve ids ccall((:cfmm_ctx_create_multi, LIB), Cint, (Int32, Ptr{Int32}, Int32, Ref{Ptr{Cvoid}}),
                               length(ids), ids, n_tokens, ctxref)
    end
    check(Ptr{Cvoid}(C_NULL), rc)
    ctx = ctxref[]
    try
        return build_router(objective, cfmms, n_tokens, ctx)
    catch
        ccall((:cfmm_ctx_destroy, LIB), Cvoid, (Ptr{Cvoid},), ctx)    # a refused upload (ArgumentError) must not leak the device context
        rethrow()
    end
end

function build_router(objective::O, cfmms::Vector{C}, n_tokens, ctx::Ptr{Cvoid}) where {O<:Objective,C<:CFMM{Float64}}
    order = Int[]
    # --- ProductTwoCoin segment (src/cfmms.jl:101-111) ---
    idx = findall(c -> c isa ProductTwoCoin, cfmms)
    if !isempty(idx)
        R = Float64[c.R[j] for j in 1:2, c in cfmms[idx]]            # 2×m column-major == [m][2] row-major
        γ = Float64[c.γ for c in cfmms[idx]]
        Ai = Int32[c.Ai[j] - 1 for j in 1:2, c in cfmms[idx]]        # 1-based -> 0-based
        GC.@preserve R γ Ai check(ctx, ccall((:cfmm_pools_add_product, LIB), Cint,
            (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}), ctx, length(idx), R, γ, Ai))
        append!(order, idx)
    end
    # --- GeometricMeanTwoCoin segment (src/cfmms.jl:152-165) ---
    idx = findall(c -> c isa GeometricMeanTwoCoin, cfmms)
    if !isempty(idx)
        R = Float64[c.R[j] for j in 1:2, c in cfmms[idx]]
        w = Float64[c.w[j] for j in 1:2, c in cfmms[idx]]
        γ = Float64[c.γ for c in cfmms[idx]]
        Ai = Int32[c.Ai[j] - 1 for j in 1:2, c in cfmms[idx]]
        GC.@preserve R w γ Ai check(ctx, ccall((:cfmm_pools_add_geomean, LIB), Cint,
            (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}), ctx, length(idx), R, w, γ, Ai))
        append!(order, idx)
    end
    # --- UniV3 segment, ticks in CSR form (src/cfmms.jl:226-245) ---
    idx = findall(c -> c isa UniV3, cfmms)
    if !isempty(idx)
        cp = Float64[c.current_price for c in cfmms[idx]]
        γ = Float64[c.γ for c in cfmms[idx]]
        Ai = Int32[c.Ai[j] - 1 for j in 1:2, c in cfmms[idx]]
        off = Int64[0; cumsum(length(c.lower_ticks) for c in cfmms[idx])]
        ticks = reduce(vcat, (Float64.(c.lower_ticks) for c in cfmms[idx]))
        liq = reduce(vcat, (Float64.(c.liquidity) for c in cfmms[idx]))
        GC.@preserve cp γ Ai off ticks liq check(ctx, ccall((:cfmm_pools_add_univ3, LIB), Cint,
            (Ptr{Cvoid}, Int64, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Ptr{Int64}, Ptr{Float64}, Ptr{Float64}),
            ctx, length(idx), cp, γ, Ai, off, ticks, liq))
        append!(order, idx)
    end
    # The reference's plugin seam (src/cfmms.jl:35,56; src/router.jl:40): a Router takes ANY CFMM{T} subtype that has a
    # find_arb!(Δ, Λ, cfmm, v) method.  Pools of such a type are not uploaded: every evaluation calls the user's own method
    # on the host and adds its (Λ − Δ) and dual term to what the device returns (host_part! below).
    host = setdiff(1:length(cfmms), order)
    for i in host
        hasmethod(CFMMRouter.find_arb!, Tuple{Vector{Float64},Vector{Float64},typeof(cfmms[i]),Vector{Float64}}) ||
            throw(ArgumentError("cfmms[$i]::$(typeof(cfmms[i])) has no device kernel and no find_arb!(Δ, Λ, cfmm, v) method"))
    end
    Δs = [CFMMRouter.zerotrade(c) for c in cfmms]; Λs = [CFMMRouter.zerotrade(c) for c in cfmms]    # src/router.jl:23-26
    r = AMDRouter{O,Float64}(objective, convert(Vector{CFMM{Float64}}, cfmms), Δs, Λs, zeros(n_tokens), ctx, order,
                             zeros(n_tokens), Ref(0.0), collect(host))
    finalizer(x -> ccall((:cfmm_ctx_destroy, LIB), Cvoid, (Ptr{Cvoid},), x.ctx), r)
    return r
end

# Pools without a device kernel: the user's find_arb! per pool (src/router.jl:40), their part of Ψ (src/router.jl:98-100)
# and of the dual scalar (:82) added to the device's
function host_part!(r::AMDRouter, v)
    for i in r.host
        c = r.cfmms[i]
        vl = v[c.Ai]
        CFMMRouter.find_arb!(r.Δs[i], r.Λs[i], c, vl)
        r.acc[] += sum(r.Λs[i] .* vl) - sum(r.Δs[i] .* vl)
        r.Ψ[c.Ai] .+= r.Λs[i] .- r.Δs[i]
    end
    return nothing
end

# find_arb!(r::Router, v) -- src/router.jl:38-42: materialising device sweep, then r.Δs/r.Λs are filled
function find_arb!(r::AMDRouter, v)
    vv = Vector{Float64}(v)
    GC.@preserve vv check(r.ctx, ccall((:cfmm_find_arb, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), r.ctx, vv))
    m = length(r.order)
    D = Matrix{Float64}(undef, 2, m); L = Matrix{Float64}(undef, 2, m)
    GC.@preserve D L check(r.ctx, ccall((:cfmm_get_trades, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), r.ctx, D, L))
    for (k, i) in enumerate(r.order)
        r.Δs[i] .= @view D[:, k]; r.Λs[i] .= @view L[:, k]
    end
    check(r.ctx, ccall((:cfmm_netflows, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}), r.ctx, r.Ψ))
    check(r.ctx, ccall((:cfmm_dual_value, LIB), Cint, (Ptr{Cvoid}, Ref{Float64}), r.ctx, r.acc))
    host_part!(r, vv)
    return nothing
end

# the fused evaluation used inside fn/g! (no O(m) trade write-back)
function eval_pools!(r::AMDRouter, v)
    vv = Vector{Float64}(v)
    GC.@preserve vv check(r.ctx, ccall((:cfmm_eval, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ref{Float64}),
                                       r.ctx, vv, r.Ψ, r.acc))
    host_part!(r, vv)
    return nothing
end

# route!(r; ...) -- src/router.jl:58-108, line for line, with the two O(m) loops replaced by Ψ/acc
function route!(r::AMDRouter; v=nothing, verbose=false, m=5, factr=1e1, pgtol=1e-5, maxfun=15_000, maxiter=15_000)
    optimizer = L_BFGS_B(length(r.v), 17)
    if isnothing(v)
        r.v .= ones(length(r.v)) / length(r.v)
    else
        r.v .= v
    end
    bounds = zeros(3, length(r.v))
    bounds[1, :] .= 2
    bounds[2, :] .= CFMMRouter.lower_limit(r.objective)
    bounds[3, :] .= CFMMRouter.upper_limit(r.objective)

    function fn(v)
        if !all(v .== r.v)
            eval_pools!(r, v)
            r.v .= v
        end
        return CFMMRouter.f(r.objective, v) + r.acc[]          # src/router.jl:79-85
    end
    function g!(G, v)
        G .= 0
        if !all(v .== r.v)
            eval_pools!(r, v)
            r.v .= v
        end
        CFMMRouter.grad!(G, r.objective, v)
        G .+= r.Ψ                                               # src/router.jl:98-100
    end

    eval_pools!(r, r.v)                                         # src/router.jl:104
    _, vopt = optimizer(fn, g!, r.v, bounds, m=m, factr=factr, pgtol=pgtol, iprint=verbose ? 1 : -1,
                        maxfun=maxfun, maxiter=maxiter)
    r.v .= vopt
    find_arb!(r, vopt)                                          # src/router.jl:107
end

# update_reserves!(r) -- src/router.jl:127-132 (upstream calls a per-pool method that exists nowhere).  Here:
# the pools move to R + γΔ − Λ (src/cfmms.jl:26-31) in place on the device, from the trades of the latest
# find_arb!/route!; UniV3 pools move to the price the arbitrage left them at.  The host-side pool objects
# are refreshed from the device (16 bytes per two-coin pool, 8 per UniV3 pool); `sync=false` skips that.
function update_reserves!(r::AMDRouter; sync::Bool=true)
    for i in r.host      # host-evaluated pool types: their own per-pool method, as src/router.jl:129 calls it
        CFMMRouter.update_reserves!(r.cfmms[i], r.Δs[i], r.Λs[i], r.v[r.cfmms[i].Ai])
    end
    check(r.ctx, ccall((:cfmm_update_reserves, LIB), Cint, (Ptr{Cvoid},), r.ctx))
    if sync
        seg, pos = Int32(0), 0
        for T in (ProductTwoCoin, GeometricMeanTwoCoin, UniV3)
            idx = [i for i in r.order[pos+1:end] if r.cfmms[i] isa T]   # r.order is grouped by family
            isempty(idx) && continue
            if T === UniV3
                p = Vector{Float64}(undef, length(idx))
                GC.@preserve p check(r.ctx, ccall((:cfmm_get_prices, LIB), Cint, (Ptr{Cvoid}, Int32, Ptr{Float64}), r.ctx, seg, p))
                for (k, i) in enumerate(idx)
                    c = r.cfmms[i]
                    r.cfmms[i] = UniV3(p[k], c.lower_ticks, c.liquidity, c.γ, c.Ai)   # re-derives current_tick (:235)
                end
            else
                R = Matrix{Float64}(undef, 2, length(idx))
                GC.@preserve R check(r.ctx, ccall((:cfmm_get_reserves, LIB), Cint, (Ptr{Cvoid}, Int32, Ptr{Float64}), r.ctx, seg, R))
                for (k, i) in enumerate(idx)
                    r.cfmms[i].R .= @view R[:, k]
                end
            end
            seg += Int32(1); pos += length(idx)
        end
    end
    foreach(d -> fill!(d, 0), r.Δs); foreach(l -> fill!(l, 0), r.Λs)
    return nothing
end

# Optional fast path: the whole of route! inside the library (cfmm_route: its own L-BFGS-B, the
# objective's f/grad!/bounds restated in C++) -- one ccall, no Julia between two device sweeps.
struct RouteInfo
    f::Float64
    proj_grad::Float64
    iterations::Int32
    evaluations::Int32
    sweeps::Int32
    status::Int32
    sweep_seconds::Float64
    total_seconds::Float64
end

function route_native!(r::AMDRouter; v=nothing, m=5, factr=1e1, pgtol=1e-5, maxfun=15_000, maxiter=15_000)
    isempty(r.host) || throw(ArgumentError("route_native! runs inside the library: routers with host-evaluated pool types use route!"))
    obj = r.objective
    kind, vec, idx = if obj isa CFMMRouter.LinearNonnegative
        (Int32(0), Vector{Float64}(obj.c), Int32(0))
    elseif obj isa CFMMRouter.BasketLiquidation
        (Int32(1), Vector{Float64}(obj.Δin), Int32(obj.i - 1))
    else
        throw(ArgumentError("route_native! knows LinearNonnegative and BasketLiquidation"))
    end
    n = length(r.v)
    vout = Vector{Float64}(undef, n)
    info = Ref(RouteInfo(0.0, 0.0, 0, 0, 0, 0, 0.0, 0.0))
    v0vec = isnothing(v) ? Float64[] : Vector{Float64}(v)
    GC.@preserve vec v0vec vout check(r.ctx, ccall((:cfmm_route, LIB), Cint,
        (Ptr{Cvoid}, Int32, Ptr{Float64}, Int32, Ptr{Float64}, Int32, Float64, Float64, Int32, Int32,
         Ptr{Float64}, Ptr{Float64}, Ref{RouteInfo}),
        r.ctx, kind, vec, idx, isnothing(v) ? C_NULL : pointer(v0vec), m, factr, pgtol, maxfun, maxiter,
        vout, r.Ψ, info))
    r.v .= vout
    # trades were materialised at v* by the same call
    mm = length(r.order)
    D = Matrix{Float64}(undef, 2, mm); L = Matrix{Float64}(undef, 2, mm)
    GC.@preserve D L check(r.ctx, ccall((:cfmm_get_trades, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), r.ctx, D, L))
    for (k, i) in enumerate(r.order)
        r.Δs[i] .= @view D[:, k]; r.Λs[i] .= @view L[:, k]
    end
    return info[]
end

# cfmm_polish (include/cfmm_amd.h): tighten a route!'s result with a gradient-only projected chord-Newton iteration on the
# dual's optimality conditions -- NOT part of CFMMRouter.jl (its route! ends where L-BFGS-B ends, with a stationarity
# residual of ~1e-6 max|Ψ| on interior optima); overwrites r.v, r.Δs, r.Λs like route! does.
struct PolishInfo
    residual0::Float64
    residual::Float64
    iterations::Int32
    sweeps::Int32
    total_seconds::Float64
end

function polish!(r::AMDRouter; max_iters=8, rel_step=1e-7)
    isempty(r.host) || throw(ArgumentError("polish! runs inside the library: not available with host-evaluated pool types"))
    obj = r.objective
    kind, vec, idx = if obj isa CFMMRouter.LinearNonnegative
        (Int32(0), Vector{Float64}(obj.c), Int32(0))
    elseif obj isa CFMMRouter.BasketLiquidation
        (Int32(1), Vector{Float64}(obj.Δin), Int32(obj.i - 1))
    else
        throw(ArgumentError("polish! knows LinearNonnegative and BasketLiquidation"))
    end
    vio = Vector{Float64}(r.v)
    info = Ref(PolishInfo(0.0, 0.0, 0, 0, 0.0))
    GC.@preserve vec vio check(r.ctx, ccall((:cfmm_polish, LIB), Cint,
        (Ptr{Cvoid}, Int32, Ptr{Float64}, Int32, Ptr{Float64}, Int32, Float64, Ptr{Float64}, Ref{PolishInfo}),
        r.ctx, kind, vec, idx, vio, max_iters, rel_step, r.Ψ, info))
    r.v .= vio
    mm = length(r.order)
    D = Matrix{Float64}(undef, 2, mm); L = Matrix{Float64}(undef, 2, mm)
    GC.@preserve D L check(r.ctx, ccall((:cfmm_get_trades, LIB), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}), r.ctx, D, L))
    for (k, i) in enumerate(r.order)
        r.Δs[i] .= @view D[:, k]; r.Λs[i] .= @view L[:, k]
    end
    return info[]
end

# Library options (include/cfmm_amd.h, cfmm_set_option): e.g. set_option!(r, "stop_in_noise", 1) for the noise-floor
# stop of route_native! (default 0 = the stopping rules of L-BFGS-B 3.0), set_option!(r, "fast_math", 0), ...
function set_option!(r::AMDRouter, key::AbstractString, value::Integer)
    check(r.ctx, ccall((:cfmm_set_option, LIB), Cint, (Ptr{Cvoid}, Cstring, Int64), r.ctx, key, Int64(value)))
    return nothing
end

# Sharding over the GPUs of a node, one Julia process per GPU (Distributed.jl workers, MPI.jl ranks ...): north_star's
# "RCCL all-reduce of Ψ and ∇g over xGMI per outer iteration" in three calls (include/cfmm_amd.h, sharded runs through RCCL).
# Every rank builds its AMDRouter over ITS contiguous block of the cfmms vector (the axis of src/router.jl:39) on its own
# device; rank 0 draws the id, the launcher's own channel carries the 128 bytes, every rank joins:
#     id = rank == 0 ? rccl_unique_id() : nothing;  id = MPI.bcast(id, 0, comm)      # or remotecall_fetch / a file
#     rccl_init_rank!(r, id, world, rank)            # collective
# From then on find_arb!(r, v), eval_pools!, route!, route_native! return the Ψ / dual value of the WHOLE market on every
# rank (bit-identical: the ranks' L-BFGS-B stays in lockstep); r.Δs / r.Λs are the local shard's trades.
function rccl_unique_id()
    id = Vector{UInt8}(undef, 128)
    GC.@preserve id check(C_NULL, ccall((:cfmm_rccl_unique_id, LIB), Cint, (Ptr{UInt8},), id))
    return id
end

function rccl_init_rank!(r::AMDRouter, id::Vector{UInt8}, world::Integer, rank::Integer)
    length(id) == 128 || throw(ArgumentError("the id of rccl_unique_id() has 128 bytes"))
    GC.@preserve id check(r.ctx, ccall((:cfmm_rccl_init_rank, LIB), Cint, (Ptr{Cvoid}, Ptr{UInt8}, Int32, Int32),
                                       r.ctx, id, Int32(world), Int32(rank)))
    return nothing
end

# a communicator the host created itself (RCCL.jl / a C launcher); C_NULL switches the exchange off
function set_rccl_comm!(r::AMDRouter, comm::Ptr{Cvoid})
    check(r.ctx, ccall((:cfmm_set_rccl_comm, LIB), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), r.ctx, comm))
    return nothing
end

# netflows!(ψ, r) / netflows(r) -- src/router.jl:111-125.
# exact = true (default): the reference's own loop over r.Δs / r.Λs / r.cfmms in router order (:113-116) -- the rows are on
# the host anyway after find_arb! / route! -- so `all_flows .== netflows(r)` of the reference's router test (test/arb.jl:16)
# holds unedited.  exact = false: the device's reduction of the same sweep (what route! itself consumed as the gradient;
# within 1e-12 max|Ψ| of the serial sum), no O(m) host loop.
function netflows!(ψ, r::AMDRouter; exact::Bool=true)
    if exact
        fill!(ψ, 0)
        for (Δ, Λ, c) in zip(r.Δs, r.Λs, r.cfmms)
            ψ[c.Ai] .+= Λ .- Δ
        end
    else
        ψ .= r.Ψ
    end
    return nothing
end
function netflows(r::AMDRouter; exact::Bool=true)
    ψ = zero(r.v)
    netflows!(ψ, r; exact=exact)
    return ψ
end

end # module
